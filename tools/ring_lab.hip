// ring_lab -- stand-alone bench of the round-3 forward-type ring kernel core (1x1 gather): fp32 A cut in the loop,
// B (weights) PRE-CUT into three bf16 planes by its producer, wave layout WGM x 1 (each wave 32 rows x all BN columns), so one
// A-fragment cut (44 VALU) feeds 24 MFMAs.  Variants are template parameters; the winner is what conv_gemm.hip ships
// (conv_fwd_pc_kernel).  Not part of the library.
//   C[m][n] = sum_k A[m][k] B[k][n], A row-major fp32, B handed over as
//   Bp[stage = k/16][tile_n][kq 2][plane 3][pos BN][8] bf16, pos = (n % NB) * 32 + n / NB inside a BN-column tile (NB = BN/32):
//   the lane that owns row-block position l31 of column block j holds column NB*l31 + j, i.e. NB adjacent columns over its NB
//   accumulators -> 16-byte epilogue stores.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ring_lab.hip -o tools/ring_lab
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct LabP {
  const float* A; const unsigned short* Bp; float* C;
  int M, N, K, lda, ldc;
  size_t a_bs, bp_bs, c_bs;       // per batch element (floats / bf16 elements / floats)
  int tiles_n, ntiles;
  unsigned a_bytes, bp_bytes;
};

__device__ __forceinline__ i32x4 make_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long a = (unsigned long long)ptr;
  i32x4 r;
  r[0] = (int)(unsigned)(a & 0xffffffffull); r[1] = (int)(unsigned)((a >> 32) & 0xffffull); r[2] = (int)bytes; r[3] = 0x00020000;
  return r;
}
// FAST = 0: the round-2 form (m0 saved / restored, s_nop 4).  FAST = 1: m0 declared clobbered, one wait state.
template <int FAST>
__device__ __forceinline__ void lds_dma16(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lds_dst) {
  if (FAST) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                 : : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory", "m0");
  } else {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
  }
}
__device__ __forceinline__ int xcd_swz(int bid, int n) {
  const int q = n >> 3, r = n & 7, xcd = bid & 7, i = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}
__device__ __forceinline__ void split8(const float* v, u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned x0 = __float_as_uint(v[2 * q]), x1 = __float_as_uint(v[2 * q + 1]);
    hi[q] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
    const float r0 = v[2 * q] - __uint_as_float(x0 & 0xffff0000u), r1 = v[2 * q + 1] - __uint_as_float(x1 & 0xffff0000u);
    const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1);
    mid[q] = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(y0 & 0xffff0000u), s1 = r1 - __uint_as_float(y1 & 0xffff0000u);
    lo[q] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
  }
}
__device__ __forceinline__ f32x16 mma(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// WGM waves (BM = 32 * WGM rows), BN = 32 * NB columns, NST LDS stages of 16 k, WPS = waves per SIMD the launch bound asks for
template <int WGM, int NB, int NST, int WPS, int FAST, int PRIO>
__global__ __launch_bounds__(64 * WGM, WPS)
void gemm_pc(LabP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 32 * WGM, BN = 32 * NB, BK = 16;
  constexpr int A_BYTES = BM * BK * 4;             // fp32 rows of 64 B, XOR-swizzled 16-B chunks
  constexpr int B_BYTES = 2 * 3 * BN * 16;         // [kq 2][plane 3][pos BN][16 B]
  constexpr int ST_BYTES = A_BYTES + B_BYTES;
  constexpr int APC = A_BYTES / 1024, BPC = B_BYTES / 1024;       // 1-KiB LDS-DMA pieces per stage
  constexpr int AI = APC / WGM, BI = BPC / WGM, BREM = BPC % WGM;  // per wave: AI of A, BI of B, + 1 of B for waves < BREM
  static_assert(APC % WGM == 0, "A pieces divide over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tile = xcd_swz(blockIdx.x, p.ntiles);
  const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
  const int m0 = tile_m * BM;
  const float* Ab = p.A + (size_t)blockIdx.z * p.a_bs;
  float* Cb = p.C + (size_t)blockIdx.z * p.c_bs;
  const i32x4 rsA = make_rsrc(Ab, p.a_bytes), rsB = make_rsrc(p.Bp + (size_t)blockIdx.z * p.bp_bs, p.bp_bytes);
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;

  // A piece x = wid * AI + r: rows 16x .. 16x+15 (lane: row 16x + lane/4, swizzled 16-B chunk); B piece: bytes 1024 y .. of the
  // stage's 12 KB block, y = wid * BI + r, and y = WGM * BI + wid for the waves that carry one more
  const bool extra = BREM > 0 && wid < BREM;
  unsigned a_voff[AI];
#pragma unroll
  for (int r = 0; r < AI; ++r) {
    const int row = 16 * (wid * AI + r) + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    const int gm = m0 + row;
    a_voff[r] = gm < p.M ? (unsigned)(gm * p.lda + 4 * c) * 4u : 0x80000000u;
  }
  const unsigned b_voff = (unsigned)lane * 16u;
  const unsigned b_stage = (unsigned)p.tiles_n * B_BYTES;         // bytes per 16-k stage of the pre-cut operand
  const unsigned b_tile = (unsigned)tile_n * B_BYTES;
  auto issue = [&](int st, int kb) {
    const unsigned S = lds0 + (unsigned)(st * ST_BYTES), SB = S + A_BYTES;
    const unsigned bsrc = (unsigned)kb * b_stage + b_tile;
#pragma unroll
    for (int r = 0; r < AI; ++r) lds_dma16<FAST>(a_voff[r], rsA, (unsigned)kb * (BK * 4), S + (unsigned)(wid * AI + r) * 1024u);
#pragma unroll
    for (int r = 0; r < BI; ++r) lds_dma16<FAST>(b_voff, rsB, bsrc + (unsigned)(wid * BI + r) * 1024u, SB + (unsigned)(wid * BI + r) * 1024u);
    if (extra) lds_dma16<FAST>(b_voff, rsB, bsrc + (unsigned)(WGM * BI + wid) * 1024u, SB + (unsigned)(WGM * BI + wid) * 1024u);
  };

  f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  const int h = lane >> 5, l31 = lane & 31;
  const int f = (l31 >> 2) & 3;
  const int a_rd = (wid * 32 + l31) * 64;                                   // byte offset of this lane's A row
  const int a_c0 = ((2 * h) ^ f) * 16, a_c1 = ((2 * h + 1) ^ f) * 16;
  const int b_rd = A_BYTES + (h * 3 * BN + l31) * 16;                       // + (plane * BN + 32 j) * 16
  auto compute = [&](int st) {
    const char* S = smem + st * ST_BYTES;
    float af[8];
    {
      const float4 v0 = *reinterpret_cast<const float4*>(S + a_rd + a_c0);
      const float4 v1 = *reinterpret_cast<const float4*>(S + a_rd + a_c1);
      af[0] = v0.x; af[1] = v0.y; af[2] = v0.z; af[3] = v0.w; af[4] = v1.x; af[5] = v1.y; af[6] = v1.z; af[7] = v1.w;
    }
    u32x4 bh[NB], bm[NB], bl[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      bh[j] = *reinterpret_cast<const u32x4*>(S + b_rd + (0 * BN + 32 * j) * 16);
      bm[j] = *reinterpret_cast<const u32x4*>(S + b_rd + (1 * BN + 32 * j) * 16);
      bl[j] = *reinterpret_cast<const u32x4*>(S + b_rd + (2 * BN + 32 * j) * 16);
    }
    u32x4 ah, am, al;
    split8(af, ah, am, al);
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      f32x16 c = acc[j];
      c = mma(al, bh[j], c); c = mma(ah, bl[j], c); c = mma(am, bm[j], c);
      c = mma(am, bh[j], c); c = mma(ah, bm[j], c); c = mma(ah, bh[j], c);
      acc[j] = c;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
  };

  const int nkb = p.K / BK;
  // NST - 1 stages in flight across every barrier
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nkb) issue(s, s);
  int st = 0;
  for (int kb = 0; kb < nkb; ++kb) {
    // this wave's share of stage kb has landed: at most the (NST - 2) younger stages may still be in flight
    const int younger = min(NST - 2, nkb - 1 - kb);
    if (extra) {
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * (AI + BI + 1)) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(AI + BI + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * (AI + BI)) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(AI + BI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int stn = st + NST - 1; if (stn >= NST) stn -= NST;
    if (kb + NST - 1 < nkb) issue(stn, kb + NST - 1);
    compute(st);
    st = st + 1 == NST ? 0 : st + 1;
  }

  // epilogue: lane holds rows wid*32 + (e&3) + 8*(e>>2) + 4*h, columns NB*l31 + j (j = 0..NB-1): one 16-byte (NB = 4) store
  const int colr = NB * l31;
  const int col = tile_n * BN + colr;
  if (col < p.N) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + wid * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (row >= p.M) continue;
      float* dst = Cb + (size_t)row * p.ldc + col;
      if (NB == 4) *reinterpret_cast<float4*>(dst) = make_float4(acc[0][e], acc[1][e], acc[2][e], acc[3][e]);
      else if (NB == 2) *reinterpret_cast<float2*>(dst) = make_float2(acc[0][e], acc[1][e]);
      else
        for (int j = 0; j < NB; ++j) dst[j] = acc[j][e];
    }
  }
#endif
}

// the round-2 kernel (both operands cut in the loop, 2 x 2 waves of 64 x 64), for the A/B on the same box
struct LabO { const float* A; const float* B; float* C; int M, N, K, lda, ldb, ldc; size_t a_bs, b_bs, c_bs; int tiles_n, ntiles; unsigned a_bytes, b_bytes; };
template <int FAST>
__global__ __launch_bounds__(256, 3)
void gemm_r2(LabO p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 128, BN = 128, BK = 16, NST = 3;
  constexpr int A_FL = BM * BK, B_FL = BK * BN, ST_FL = A_FL + B_FL;
  constexpr int AI = 2, BI = 2;
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  float* smem = reinterpret_cast<float*>(smem_c);
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int tile = xcd_swz(blockIdx.x, p.ntiles);
  const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const float* Ab = p.A + (size_t)blockIdx.z * p.a_bs;
  const float* Bb = p.B + (size_t)blockIdx.z * p.b_bs;
  float* Cb = p.C + (size_t)blockIdx.z * p.c_bs;
  const i32x4 rsA = make_rsrc(Ab, p.a_bytes), rsB = make_rsrc(Bb, p.b_bytes);
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
  unsigned a_voff[AI], b_voff[BI];
#pragma unroll
  for (int r = 0; r < AI; ++r) {
    const int row = 16 * (wid * AI + r) + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    const int gm = m0 + row;
    a_voff[r] = gm < p.M ? (unsigned)(gm * p.lda + 4 * c) * 4u : 0x80000000u;
  }
#pragma unroll
  for (int r = 0; r < BI; ++r) {
    const int krow = 2 * (wid * BI + r) + lane / 32;
    const int nn = n0 + 4 * (lane % 32);
    b_voff[r] = nn < p.N ? (unsigned)(krow * p.ldb + nn) * 4u : 0x80000000u;
  }
  auto issue = [&](int st, int kb) {
    const unsigned As = lds0 + (unsigned)(st * ST_FL) * 4u, Bs = As + A_FL * 4u;
#pragma unroll
    for (int r = 0; r < AI; ++r) lds_dma16<FAST>(a_voff[r], rsA, (unsigned)kb * (BK * 4), As + (unsigned)(wid * AI + r) * 1024u);
#pragma unroll
    for (int r = 0; r < BI; ++r) lds_dma16<FAST>(b_voff[r], rsB, (unsigned)kb * (unsigned)(BK * 4) * (unsigned)p.ldb, Bs + (unsigned)(wid * BI + r) * 1024u);
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int h = lane >> 5, l31 = lane & 31;
  const int f = (l31 >> 2) & 3;
  const int a_rd = (wm * 64 + l31) * BK;
  const int a_c0 = ((2 * h) ^ f) * 4, a_c1 = ((2 * h + 1) ^ f) * 4;
  const int b_rd = A_FL + (8 * h) * BN + wn * 64 + 2 * l31;
  auto compute = [&](int st) {
    const float* S = smem + st * ST_FL;
    float af[2][8], bf[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 v0 = *reinterpret_cast<const float4*>(S + a_rd + i * 32 * BK + a_c0);
      const float4 v1 = *reinterpret_cast<const float4*>(S + a_rd + i * 32 * BK + a_c1);
      af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w;
      af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
    }
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const float2 b = *reinterpret_cast<const float2*>(S + b_rd + s8 * BN);
      bf[0][s8] = b.x; bf[1][s8] = b.y;
    }
    u32x4 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { split8(af[i], ah[i], am[i], al[i]); split8(bf[i], bh[i], bm[i], bl[i]); }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];
        c = mma(al[i], bh[j], c); c = mma(ah[i], bl[j], c); c = mma(am[i], bm[j], c);
        c = mma(am[i], bh[j], c); c = mma(ah[i], bm[j], c); c = mma(ah[i], bh[j], c);
        acc[i][j] = c;
      }
  };
  const int nkb = p.K / BK;
  issue(0, 0);
  if (nkb > 1) issue(1, 1);
  int st = 0;
  for (int kb = 0; kb < nkb; ++kb) {
    if (kb + 1 < nkb) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(AI + BI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int st2 = st + 2; if (st2 >= NST) st2 -= NST;
    if (kb + 2 < nkb) issue(st2, kb + 2);
    compute(st);
    st = st + 1 == NST ? 0 : st + 1;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      const int col = n0 + wn * 64 + 2 * l31;
      if (row < p.M && col < p.N)
        *reinterpret_cast<float2*>(Cb + (size_t)row * p.ldc + col) = make_float2(acc[i][0][e], acc[i][1][e]);
    }
#endif
}

__global__ void gemm_ref(LabO p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)p.M * p.N) return;
  const int m = (int)(i / p.N), n = (int)(i % p.N);
  const float* A = p.A + (size_t)blockIdx.z * p.a_bs + (size_t)m * p.lda;
  const float* B = p.B + (size_t)blockIdx.z * p.b_bs + n;
  double acc = 0;
  for (int k = 0; k < p.K; ++k) acc += (double)A[k] * B[(size_t)k * p.ldb];
  p.C[(size_t)blockIdx.z * p.c_bs + (size_t)m * p.ldc + n] = (float)acc;
}

// device-side producer of the pre-cut operand (what the re-pack kernels of the library will do): one thread per (k / 8, n)
__global__ void precut_kernel(const float* B, unsigned short* Bp, int K, int N, int BN, size_t b_bs, size_t bp_bs) {
  const int NBc = BN / 32;
  const int tiles_n = (N + BN - 1) / BN;
  const size_t total = (size_t)(K / 8) * tiles_n * BN;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int pos = (int)(i % BN); size_t q = i / BN;
  const int tn = (int)(q % tiles_n); const int kq = (int)(q / tiles_n);
  const int nl = (pos % 32) * NBc + pos / 32;       // pos = (nl % NB) * 32 + nl / NB
  const int n = tn * BN + nl;
  B += (size_t)blockIdx.y * b_bs; Bp += (size_t)blockIdx.y * bp_bs;
  unsigned short hi[8], mid[8], lo[8];
  for (int j = 0; j < 8; ++j) {
    const float x = n < N ? B[(size_t)(kq * 8 + j) * N + n] : 0.f;
    const unsigned u = __float_as_uint(x);
    const float r = x - __uint_as_float(u & 0xffff0000u);
    const unsigned ur = __float_as_uint(r);
    const float s = r - __uint_as_float(ur & 0xffff0000u);
    hi[j] = (unsigned short)(u >> 16); mid[j] = (unsigned short)(ur >> 16); lo[j] = (unsigned short)(__float_as_uint(s) >> 16);
  }
  // [stage = kq / 2][tile_n][kq & 1][plane][pos][8]
  const size_t base = ((((size_t)(kq >> 1) * tiles_n + tn) * 2 + (kq & 1)) * 3) * BN;
  for (int j = 0; j < 8; ++j) {
    Bp[((base + 0 * (size_t)BN + pos) * 8) + j] = hi[j];
    Bp[((base + 1 * (size_t)BN + pos) * 8) + j] = mid[j];
    Bp[((base + 2 * (size_t)BN + pos) * 8) + j] = lo[j];
  }
}

struct Shape { const char* name; int M, N, K, batch; };
static void fill(std::vector<float>& v, unsigned seed) {
  unsigned s = seed * 2654435761u + 12345u;
  for (auto& x : v) {
    s = s * 1664525u + 1013904223u; const float m = ((s >> 8) & 0xFFFFFF) / 8388608.0f - 1.0f;
    s = s * 1664525u + 1013904223u; x = std::ldexp(m, -(int)((s >> 24) & 7));
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  const int only = argc > 2 ? atoi(argv[2]) : -1;          // run a single variant index (for PMC passes)
  std::vector<Shape> shapes = {
      {"wino_resblock (36 planes 512x1024x1024)", 512, 1024, 1024, 36},
      {"down4 (8192x512x4096)", 8192, 512, 4096, 1},
      {"down2 (131072x128x1024)", 131072, 128, 1024, 1},
      {"dual_up3 phase (131072x64x1536)", 131072, 64, 1536, 4},
  };
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (const Shape& s : shapes) {
    const size_t na = (size_t)s.M * s.K * s.batch, nb = (size_t)s.K * s.N * s.batch, nc = (size_t)s.M * s.N * s.batch;
    std::vector<float> ha(na), hb(nb);
    fill(ha, 1); fill(hb, 2);
    float *dA, *dB, *dC, *dR;
    CK(hipMalloc((void**)&dA, na * 4)); CK(hipMalloc((void**)&dB, nb * 4)); CK(hipMalloc((void**)&dC, nc * 4)); CK(hipMalloc((void**)&dR, nc * 4));
    CK(hipMemcpy(dA, ha.data(), na * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hb.data(), nb * 4, hipMemcpyHostToDevice));
    LabO o{};
    o.A = dA; o.B = dB; o.C = dR; o.M = s.M; o.N = s.N; o.K = s.K; o.lda = s.K; o.ldb = s.N; o.ldc = s.N;
    o.a_bs = (size_t)s.M * s.K; o.b_bs = (size_t)s.K * s.N; o.c_bs = (size_t)s.M * s.N;
    o.a_bytes = (unsigned)((size_t)s.M * s.K * 4); o.b_bytes = (unsigned)((size_t)s.K * s.N * 4);
    hipLaunchKernelGGL(gemm_ref, dim3((unsigned)(((size_t)s.M * s.N + 255) / 256), 1, s.batch), dim3(256), 0, st, o);
    CK(hipStreamSynchronize(st));
    o.C = dC;
    const double flops = 2.0 * s.M * s.N * s.K * s.batch;
    printf("== %s\n", s.name);
    std::vector<float> r(nc); CK(hipMemcpy(r.data(), dR, nc * 4, hipMemcpyDeviceToHost));
    int vidx = 0;
    auto run = [&](const char* what, auto&& fn) {
      const int my = vidx++;
      if (only >= 0 && my != only) return;
      CK(hipMemsetAsync(dC, 0, nc * 4, st));
      fn(); CK(hipStreamSynchronize(st));
      std::vector<float> c(nc); CK(hipMemcpy(c.data(), dC, nc * 4, hipMemcpyDeviceToHost));
      double num = 0, den = 0;
      for (size_t i = 0; i < nc; ++i) { const double d = (double)c[i] - r[i]; num += d * d; den += (double)r[i] * r[i]; }
      float best = 1e30f;
      for (int rnd = 0; rnd < 3; ++rnd) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); best = std::min(best, t / reps);
      }
      printf("   [%2d] %-26s rel-L2 vs fp64 %.3e   %8.3f ms  %7.1f fp32-equivalent TFLOP/s\n", my, what, std::sqrt(num / den), best, flops / best * 1e-9);
      fflush(stdout);
    };
    // pre-cut operands for 128- and 64-column tiles (device producer)
    unsigned short* dP[2]; size_t bp_bs[2]; const int BNs[2] = {128, 64};
    for (int v = 0; v < 2; ++v) {
      const int BN = BNs[v], tiles_n = (s.N + BN - 1) / BN;
      bp_bs[v] = (size_t)(s.K / 16) * tiles_n * 6 * BN * 8;
      CK(hipMalloc((void**)&dP[v], bp_bs[v] * s.batch * 2));
      const size_t total = (size_t)(s.K / 8) * tiles_n * BN;
      hipLaunchKernelGGL(precut_kernel, dim3((unsigned)((total + 255) / 256), s.batch), dim3(256), 0, st, dB, dP[v], s.K, s.N, BN, (size_t)s.K * s.N, bp_bs[v]);
    }
    CK(hipStreamSynchronize(st));
    auto launch_pc = [&](auto kern, int wgm, int nbk, int nst) {
      const int BN = 32 * nbk, BM = 32 * wgm;
      const int smem = nst * (BM * 64 + 6 * BN * 16);
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
      LabP p{};
      const int v = BN == 128 ? 0 : 1;
      p.A = dA; p.Bp = dP[v]; p.C = dC; p.M = s.M; p.N = s.N; p.K = s.K; p.lda = s.K; p.ldc = s.N;
      p.a_bs = (size_t)s.M * s.K; p.bp_bs = bp_bs[v]; p.c_bs = (size_t)s.M * s.N;
      p.a_bytes = (unsigned)((size_t)s.M * s.K * 4); p.bp_bytes = (unsigned)(bp_bs[v] * 2);
      p.tiles_n = (s.N + BN - 1) / BN; p.ntiles = ((s.M + BM - 1) / BM) * p.tiles_n;
      hipLaunchKernelGGL(kern, dim3(p.ntiles, 1, s.batch), dim3(64 * wgm), smem, st, p);
    };
    auto launch_r2 = [&](auto kern) {
      const int smem = 3 * (128 * 16 + 16 * 128) * 4;
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
      LabO w = o; w.tiles_n = (s.N + 127) / 128; w.ntiles = ((s.M + 127) / 128) * w.tiles_n;
      hipLaunchKernelGGL(kern, dim3(w.ntiles, 1, s.batch), dim3(256), smem, st, w);
    };
    for (int rnd = 0; rnd < 2; ++rnd) {
      vidx = 0;
      run("r2 2x2 both cut", [&] { launch_r2(gemm_r2<0>); });
      run("r2 2x2 both cut, fast dma", [&] { launch_r2(gemm_r2<1>); });
      //                 WGM NB NST WPS FAST PRIO
      run("pc 128x128 3st 2w", [&] { launch_pc(gemm_pc<4, 4, 3, 2, 1, 0>, 4, 4, 3); });
      run("pc 128x128 2st 3w", [&] { launch_pc(gemm_pc<4, 4, 2, 3, 1, 0>, 4, 4, 2); });
      run("pc 128x128 2st 3w prio", [&] { launch_pc(gemm_pc<4, 4, 2, 3, 1, 2>, 4, 4, 2); });
      run("pc 128x128 2st 4w", [&] { launch_pc(gemm_pc<4, 4, 2, 4, 1, 0>, 4, 4, 2); });
      run("pc 128x128 4st 2w", [&] { launch_pc(gemm_pc<4, 4, 4, 2, 1, 0>, 4, 4, 4); });
      run("pc 128x128 4st 2w prio", [&] { launch_pc(gemm_pc<4, 4, 4, 2, 1, 2>, 4, 4, 4); });
      run("pc 128x128 3st 2w slowdma", [&] { launch_pc(gemm_pc<4, 4, 3, 2, 0, 0>, 4, 4, 3); });
      run("pc 256x128 2st 4w", [&] { launch_pc(gemm_pc<8, 4, 2, 4, 1, 0>, 8, 4, 2); });
      run("pc 256x128 3st 2w", [&] { launch_pc(gemm_pc<8, 4, 3, 2, 1, 0>, 8, 4, 3); });
      run("pc 256x128 4st 2w", [&] { launch_pc(gemm_pc<8, 4, 4, 2, 1, 0>, 8, 4, 4); });
      run("pc 256x128 4st 2w prio", [&] { launch_pc(gemm_pc<8, 4, 4, 2, 1, 2>, 8, 4, 4); });
      if (s.N <= 64 || s.N % 64 == 0) {
        run("pc 256x64 3st 2w", [&] { launch_pc(gemm_pc<8, 2, 3, 2, 1, 0>, 8, 2, 3); });
        run("pc 256x64 4st 4w", [&] { launch_pc(gemm_pc<8, 2, 4, 4, 1, 0>, 8, 2, 4); });
        run("pc 128x64 4st 3w", [&] { launch_pc(gemm_pc<4, 2, 4, 3, 1, 0>, 4, 2, 4); });
      }
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dR)); CK(hipFree(dP[0])); CK(hipFree(dP[1]));
  }
  return 0;
}
