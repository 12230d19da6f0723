// gemm_lab_split -- stand-alone bench: the LDS-DMA ring GEMM with the fp32 products formed on the bf16 matrix cores from
// an exact 3-way bf16 split of both operands (x = hi + mid + lo, 8 mantissa bits each, by truncation: no rounding
// error in the split), 6 of the 9 partial products (all with weight >= 2^-16; the dropped ones are below 2^-24),
// fp32 accumulation.  MODE 0 = v_mfma_f32_32x32x2_f32 (reference), 1 = 6-term split, 2 = 3-term split (hi*hi + hi*mid
// + mid*hi: ~2^-16, tf32-like, for comparison only).  Not part of the library.
// (derived from gemm_lab.hip)
// C[m][n] = sum_k A[m][k] B[k][n], A row-major (k contiguous: an NHWC activation seen through a 1x1 tap),
// B row-major [K][N] (the packed weight panel), batched over blockIdx.z (Winograd planes).
//
//   v1<WGM>: LDS-DMA staging (buffer_load_dwordx4 ... lds: no staging VGPRs, no ds_write), BK = 16, 3-stage LDS
//            ring with counted vmcnt (two stages in flight across the barrier), XOR-swizzled A rows
//            (conflict-free ds_read_b128), B fragments as ds_read_b64 of adjacent columns (float2 epilogue
//            stores), <= 128 VGPRs so that two 8-wave workgroups (or three 4-wave ones) share a CU.
//   prod   : the library's conv_fwd on the same shape (1x1 gather), linked from libswapnet_hip.so.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.hip -Iswapnet_amd/csrc -Lswapnet_amd/csrc
//               -lswapnet_hip -Wl,-rpath,'$ORIGIN/../swapnet_amd/csrc' -o tools/gemm_lab
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

struct LabP {
  const float* A; const float* B; float* C;
  int M, N, K, lda, ldb, ldc;
  size_t a_bs, b_bs, c_bs;
  int tiles_n, ntiles;
  unsigned a_bytes, b_bytes;
  unsigned long long* trace;
  int padsim;        // rows m with m % 7 == 3 are "padding taps": fetched out of range (must read as 0)
};

typedef int i32x4 __attribute__((ext_vector_type(4)));
// raw buffer descriptor (stride 0, num_records = bytes; out-of-range offsets read 0)
__device__ __forceinline__ i32x4 make_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long a = (unsigned long long)ptr;
  i32x4 r;
  r[0] = (int)(unsigned)(a & 0xffffffffull); r[1] = (int)(unsigned)((a >> 32) & 0xffffull); r[2] = (int)bytes; r[3] = 0x00020000;
  return r;
}
// One LDS-DMA instruction: every lane fetches 16 bytes at rsrc.base + voff + soff; the wave's 1 KiB lands at LDS byte
// address lds_dst + 16 * lane.  Issued from an asm statement so that hipcc does NOT know LDS is written: its waitcnt
// pass would otherwise put s_waitcnt vmcnt(0) in front of every following ds_read (one pending LDS-DMA = "may alias"),
// which serialises the ring.  Completion is counted by hand (s_waitcnt vmcnt(N) + s_barrier before the reads).
__device__ __forceinline__ void lds_dma16(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
}

__device__ __forceinline__ int xcd_swz(int bid, int n) {
  const int q = n >> 3, r = n & 7, xcd = bid & 7, i = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 8 floats -> three registers-quadruples of packed bf16 (element j of the MFMA operand = value j)
__device__ __forceinline__ void split8(const float* v, u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned x0 = __float_as_uint(v[2 * q]), x1 = __float_as_uint(v[2 * q + 1]);
    hi[q] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);                 // {x1[31:16], x0[31:16]}
    const float r0 = v[2 * q] - __uint_as_float(x0 & 0xffff0000u), r1 = v[2 * q + 1] - __uint_as_float(x1 & 0xffff0000u);
    const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1);
    mid[q] = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(y0 & 0xffff0000u), s1 = r1 - __uint_as_float(y1 & 0xffff0000u);
    lo[q] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
  }
}
// variant: the two residual subtractions of a pair as one packed v_pk_add_f32
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8k(const float* v, u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2 x = {v[2 * q], v[2 * q + 1]};
    const unsigned x0 = __float_as_uint(x[0]), x1 = __float_as_uint(x[1]);
    hi[q] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
    const f32x2 hf = {__uint_as_float(x0 & 0xffff0000u), __uint_as_float(x1 & 0xffff0000u)};
    const f32x2 r = x - hf;
    const unsigned y0 = __float_as_uint(r[0]), y1 = __float_as_uint(r[1]);
    mid[q] = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
    const f32x2 mf = {__uint_as_float(y0 & 0xffff0000u), __uint_as_float(y1 & 0xffff0000u)};
    const f32x2 t = r - mf;
    lo[q] = __builtin_amdgcn_perm(__float_as_uint(t[1]), __float_as_uint(t[0]), 0x07060302u);
  }
}
// variant: both residual levels cut from x itself (two parallel and/sub pairs instead of a 4-deep chain)
__device__ __forceinline__ void split8p(const float* v, u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned x0 = __float_as_uint(v[2 * q]), x1 = __float_as_uint(v[2 * q + 1]);
    hi[q] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
    const float a0 = __uint_as_float(x0 & 0xffff0000u), a1 = __uint_as_float(x1 & 0xffff0000u);
    const float b0 = __uint_as_float(x0 & 0xffffff00u), b1 = __uint_as_float(x1 & 0xffffff00u);
    mid[q] = __builtin_amdgcn_perm(__float_as_uint(b1 - a1), __float_as_uint(b0 - a0), 0x07060302u);
    lo[q] = __builtin_amdgcn_perm(__float_as_uint(v[2 * q + 1] - b1), __float_as_uint(v[2 * q] - b0), 0x07060302u);
  }
}
__device__ __forceinline__ f32x16 mma(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(256, 3)
void gemm_v1(LabP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int WGM = 2;
  constexpr int NW = 2 * WGM, BM = 64 * WGM, BN = 128, BK = 16, NST = 3;
  constexpr int A_FL = BM * BK, B_FL = BK * BN, ST_FL = A_FL + B_FL;
  constexpr int AI = (BM / 16) / NW;
  constexpr int BI = 8 / NW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
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
    const int krow = 2 * (wid * BI + r) + (lane >> 5);
    const int nn = n0 + 4 * (lane & 31);
    b_voff[r] = nn < p.N ? (unsigned)(krow * p.ldb + nn) * 4u : 0x80000000u;
  }
  auto issue = [&](int st, int kb) {
    const unsigned As = lds0 + (unsigned)(st * ST_FL) * 4u, Bs = As + A_FL * 4u;
#pragma unroll
    for (int r = 0; r < AI; ++r) lds_dma16(a_voff[r], rsA, (unsigned)kb * (BK * 4), As + (unsigned)(wid * AI + r) * 1024u);
#pragma unroll
    for (int r = 0; r < BI; ++r)
      lds_dma16(b_voff[r], rsB, (unsigned)kb * (BK * 4) * (unsigned)p.ldb, Bs + (unsigned)(wid * BI + r) * 1024u);
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
    for (int s = 0; s < 8; ++s) {
      const float2 b = *reinterpret_cast<const float2*>(S + b_rd + s * BN);
      bf[0][s] = b.x; bf[1][s] = b.y;
    }
    if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    } else if (MODE >= 3) {
      // term-outer: consecutive MFMAs write different accumulators (no back-to-back dependent chain)
      u32x4 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { split8k(af[i], ah[i], am[i], al[i]); split8k(bf[i], bh[i], bm[i], bl[i]); }
#define TERM(X, Y) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = mma(X[i], Y[j], acc[i][j]);
      TERM(al, bh) TERM(ah, bl) TERM(am, bm) TERM(am, bh) TERM(ah, bm) TERM(ah, bh)
#undef TERM
    } else {
      // lane (l31, h) holds k = 8h .. 8h+7 of its A row / B column: exactly the 32x32x16 operand layout
      u32x4 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { split8(af[i], ah[i], am[i], al[i]); split8(bf[i], bh[i], bm[i], bl[i]); }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 c = acc[i][j];
          if (MODE == 1) { c = mma(al[i], bh[j], c); c = mma(ah[i], bl[j], c); c = mma(am[i], bm[j], c); }   // smallest first
          c = mma(am[i], bh[j], c); c = mma(ah[i], bm[j], c); c = mma(ah[i], bh[j], c);
          acc[i][j] = c;
        }
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

// 256 x 128 workgroup tile, 8 waves (wave tile 64 x 64), 1-2 workgroups / CU: 43.7 FLOP per L2 byte instead of 32
template <int MODE>
__global__ __launch_bounds__(512, 2)
void gemm_v3(LabP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int WGM = 4;
  constexpr int NW = 2 * WGM, BM = 64 * WGM, BN = 128, BK = 16, NST = 3;
  constexpr int A_FL = BM * BK, B_FL = BK * BN, ST_FL = A_FL + B_FL;
  constexpr int AI = (BM / 16) / NW;
  constexpr int BI = 8 / NW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
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
    const int krow = 2 * (wid * BI + r) + (lane >> 5);
    const int nn = n0 + 4 * (lane & 31);
    b_voff[r] = nn < p.N ? (unsigned)(krow * p.ldb + nn) * 4u : 0x80000000u;
  }
  auto issue = [&](int st, int kb) {
    const unsigned As = lds0 + (unsigned)(st * ST_FL) * 4u, Bs = As + A_FL * 4u;
#pragma unroll
    for (int r = 0; r < AI; ++r) lds_dma16(a_voff[r], rsA, (unsigned)kb * (BK * 4), As + (unsigned)(wid * AI + r) * 1024u);
#pragma unroll
    for (int r = 0; r < BI; ++r)
      lds_dma16(b_voff[r], rsB, (unsigned)kb * (BK * 4) * (unsigned)p.ldb, Bs + (unsigned)(wid * BI + r) * 1024u);
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
    for (int s = 0; s < 8; ++s) {
      const float2 b = *reinterpret_cast<const float2*>(S + b_rd + s * BN);
      bf[0][s] = b.x; bf[1][s] = b.y;
    }
    if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    } else if (MODE >= 3) {
      // term-outer: consecutive MFMAs write different accumulators (no back-to-back dependent chain)
      u32x4 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { split8k(af[i], ah[i], am[i], al[i]); split8k(bf[i], bh[i], bm[i], bl[i]); }
#define TERM(X, Y) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = mma(X[i], Y[j], acc[i][j]);
      TERM(al, bh) TERM(ah, bl) TERM(am, bm) TERM(am, bh) TERM(ah, bm) TERM(ah, bh)
#undef TERM
    } else {
      // lane (l31, h) holds k = 8h .. 8h+7 of its A row / B column: exactly the 32x32x16 operand layout
      u32x4 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { split8(af[i], ah[i], am[i], al[i]); split8(bf[i], bh[i], bm[i], bl[i]); }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 c = acc[i][j];
          if (MODE == 1) { c = mma(al[i], bh[j], c); c = mma(ah[i], bl[j], c); c = mma(am[i], bm[j], c); }   // smallest first
          c = mma(am[i], bh[j], c); c = mma(ah[i], bm[j], c); c = mma(ah[i], bh[j], c);
          acc[i][j] = c;
        }
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

template <int MODE>
__global__ __launch_bounds__(256, 2)
void gemm_v2(LabP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int WGM = 2;
  constexpr int NW = 2 * WGM, BM = 64 * WGM, BN = 128, BK = 16, NST = 4;
  constexpr int A_FL = BM * BK, B_FL = BK * BN, ST_FL = A_FL + B_FL;
  constexpr int AI = (BM / 16) / NW;
  constexpr int BI = 8 / NW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
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
    const int krow = 2 * (wid * BI + r) + (lane >> 5);
    const int nn = n0 + 4 * (lane & 31);
    b_voff[r] = nn < p.N ? (unsigned)(krow * p.ldb + nn) * 4u : 0x80000000u;
  }
  auto issue = [&](int st, int kb) {
    const unsigned As = lds0 + (unsigned)(st * ST_FL) * 4u, Bs = As + A_FL * 4u;
#pragma unroll
    for (int r = 0; r < AI; ++r) lds_dma16(a_voff[r], rsA, (unsigned)kb * (BK * 4), As + (unsigned)(wid * AI + r) * 1024u);
#pragma unroll
    for (int r = 0; r < BI; ++r)
      lds_dma16(b_voff[r], rsB, (unsigned)kb * (BK * 4) * (unsigned)p.ldb, Bs + (unsigned)(wid * BI + r) * 1024u);
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
    for (int s = 0; s < 8; ++s) {
      const float2 b = *reinterpret_cast<const float2*>(S + b_rd + s * BN);
      bf[0][s] = b.x; bf[1][s] = b.y;
    }
    if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    } else if (MODE >= 3) {
      // term-outer: consecutive MFMAs write different accumulators (no back-to-back dependent chain)
      u32x4 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { split8k(af[i], ah[i], am[i], al[i]); split8k(bf[i], bh[i], bm[i], bl[i]); }
#define TERM(X, Y) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = mma(X[i], Y[j], acc[i][j]);
      TERM(al, bh) TERM(ah, bl) TERM(am, bm) TERM(am, bh) TERM(ah, bm) TERM(ah, bh)
#undef TERM
    } else {
      // lane (l31, h) holds k = 8h .. 8h+7 of its A row / B column: exactly the 32x32x16 operand layout
      u32x4 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { split8(af[i], ah[i], am[i], al[i]); split8(bf[i], bh[i], bm[i], bl[i]); }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 c = acc[i][j];
          if (MODE == 1) { c = mma(al[i], bh[j], c); c = mma(ah[i], bl[j], c); c = mma(am[i], bm[j], c); }   // smallest first
          c = mma(am[i], bh[j], c); c = mma(ah[i], bm[j], c); c = mma(ah[i], bh[j], c);
          acc[i][j] = c;
        }
    }
  };
  // super-stages of 2 x 16 k: one barrier per 48 MFMAs; slots (2p, 2p+1) hold super-stage parity p
  const int nk2 = p.K / (2 * BK);
  issue(0, 0); issue(1, 1);
  for (int k2 = 0; k2 < nk2; ++k2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int pr = k2 & 1;
    if (k2 + 1 < nk2) { issue(2 * (pr ^ 1), 2 * k2 + 2); issue(2 * (pr ^ 1) + 1, 2 * k2 + 3); }
    compute(2 * pr);
    compute(2 * pr + 1);
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

// ---- variant: B (the weight panel) arrives PRE-CUT: three bf16 planes in k-inner layout [plane][K/8][N][8], produced once
// per step by whoever re-packs the weights.  The B fragment is then three ds_read_b128 per column and costs no VALU; only
// the A fragments are cut in the loop.  LDS per stage: A 8 KB (fp32) + B 12 KB (3 planes x 2 k-groups x 128 cols x 16 B).
struct LabQ { LabP p; const unsigned short* Bs; size_t bs_plane, bs_batch; unsigned bs_bytes; };
template <int NSTQ>
__global__ __launch_bounds__(256, 2)
void gemm_bpre(LabQ q) {
#if defined(__HIP_DEVICE_COMPILE__)
  const LabP& p = q.p;
  constexpr int BM = 128, BN = 128, BK = 16, NST = NSTQ;
  constexpr int A_FL = BM * BK;                 // floats
  constexpr int B_BYTES = 3 * 2 * BN * 16;      // 12 KB
  constexpr int ST_BYTES = A_FL * 4 + B_BYTES;
  constexpr int AI = 2, BI = 3;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int tile = xcd_swz(blockIdx.x, p.ntiles);
  const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const float* Ab = p.A + (size_t)blockIdx.z * p.a_bs;
  float* Cb = p.C + (size_t)blockIdx.z * p.c_bs;
  const i32x4 rsA = make_rsrc(Ab, p.a_bytes), rsB = make_rsrc(q.Bs + (size_t)blockIdx.z * q.bs_batch, q.bs_bytes);
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
  unsigned a_voff[AI], b_voff[BI];
#pragma unroll
  for (int r = 0; r < AI; ++r) {
    const int row = 16 * (wid * AI + r) + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    const int gm = m0 + row;
    a_voff[r] = gm < p.M ? (unsigned)(gm * p.lda + 4 * c) * 4u : 0x80000000u;
  }
  // B instruction x = wid * 3 + r in [0, 12): plane = x / 4, k-group g = (x / 2) & 1, column half = x & 1; lane -> column
#pragma unroll
  for (int r = 0; r < BI; ++r) {
    const int x = wid * BI + r;
    const int plane = x >> 2, g = (x >> 1) & 1, half = x & 1;
    const int nn = n0 + half * 64 + lane;
    b_voff[r] = nn < p.N ? (unsigned)((size_t)plane * q.bs_plane * 2 + ((size_t)g * p.N + nn) * 16) : 0x80000000u;
  }
  auto issue = [&](int st, int kb) {
    const unsigned As = lds0 + (unsigned)(st * ST_BYTES), Bs = As + A_FL * 4u;
#pragma unroll
    for (int r = 0; r < AI; ++r) lds_dma16(a_voff[r], rsA, (unsigned)kb * (BK * 4), As + (unsigned)(wid * AI + r) * 1024u);
#pragma unroll
    for (int r = 0; r < BI; ++r)        // one stage = 2 k-groups: advance by 2 * N * 16 bytes per stage
      lds_dma16(b_voff[r], rsB, (unsigned)kb * 2u * (unsigned)p.N * 16u, Bs + (unsigned)(wid * BI + r) * 1024u);
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
  auto compute = [&](int st) {
    const char* S = reinterpret_cast<const char*>(smem) + st * ST_BYTES;
    const float* SA = reinterpret_cast<const float*>(S);
    const char* SB = S + A_FL * 4;
    float af[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 v0 = *reinterpret_cast<const float4*>(SA + a_rd + i * 32 * BK + a_c0);
      const float4 v1 = *reinterpret_cast<const float4*>(SA + a_rd + i * 32 * BK + a_c1);
      af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w;
      af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
    }
    u32x4 bh[2], bm[2], bl[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {       // lane's columns wn*64 + 2*l31 + j, k-group h
      const int col = wn * 64 + 2 * l31 + j;
      bh[j] = *reinterpret_cast<const u32x4*>(SB + ((0 * 2 + h) * 128 + col) * 16);
      bm[j] = *reinterpret_cast<const u32x4*>(SB + ((1 * 2 + h) * 128 + col) * 16);
      bl[j] = *reinterpret_cast<const u32x4*>(SB + ((2 * 2 + h) * 128 + col) * 16);
    }
    u32x4 ah[2], am[2], al[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) split8(af[i], ah[i], am[i], al[i]);
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
  if (NST == 3 && nkb > 1) issue(1, 1);
  int st = 0;
  for (int kb = 0; kb < nkb; ++kb) {
    if (NST == 3) {
      if (kb + 1 < nkb) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(AI + BI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int st2 = st + 2; if (st2 >= NST) st2 -= NST;
      if (kb + 2 < nkb) issue(st2, kb + 2);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kb + 1 < nkb) issue(st ^ 1, kb + 1);
    }
    compute(st);
    if (NST == 3) st = st + 1 == NST ? 0 : st + 1; else st ^= 1;
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


// ---- variant: 64 x 128 wave tile (workgroup 128 x 256, 4 waves, 2 workgroups / CU): the cut of an A fragment is amortised over
// 4 column blocks, of a B fragment over 2 row blocks: (16 + 32) x 5.5 VALU per 48 MFMAs = 5.5 per MFMA instead of 7.3.
template <int MODE>
__global__ __launch_bounds__(256, 2)
void gemm_w128(LabP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 128, BN = 256, BK = 16, NST = 3;
  constexpr int A_FL = BM * BK, B_FL = BK * BN, ST_FL = A_FL + B_FL;
  constexpr int AI = 2, BI = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
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
    const int krow = wid * BI + r;
    const int nn = n0 + 4 * lane;
    b_voff[r] = nn < p.N ? (unsigned)(krow * p.ldb + nn) * 4u : 0x80000000u;
  }
  auto issue = [&](int st, int kb) {
    const unsigned As = lds0 + (unsigned)(st * ST_FL) * 4u, Bs = As + A_FL * 4u;
#pragma unroll
    for (int r = 0; r < AI; ++r) lds_dma16(a_voff[r], rsA, (unsigned)kb * (BK * 4), As + (unsigned)(wid * AI + r) * 1024u);
#pragma unroll
    for (int r = 0; r < BI; ++r)
      lds_dma16(b_voff[r], rsB, (unsigned)kb * (BK * 4) * (unsigned)p.ldb, Bs + (unsigned)(wid * BI + r) * 1024u);
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int h = lane >> 5, l31 = lane & 31;
  const int f = (l31 >> 2) & 3;
  const int a_rd = (wm * 64 + l31) * BK;
  const int a_c0 = ((2 * h) ^ f) * 4, a_c1 = ((2 * h + 1) ^ f) * 4;
  const int b_rd = A_FL + (8 * h) * BN + wn * 128 + 2 * l31;
  auto compute = [&](int st) {
    const float* S = smem + st * ST_FL;
    float af[2][8], bf[4][8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 v0 = *reinterpret_cast<const float4*>(S + a_rd + i * 32 * BK + a_c0);
      const float4 v1 = *reinterpret_cast<const float4*>(S + a_rd + i * 32 * BK + a_c1);
      af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w;
      af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
    }
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const float2 b = *reinterpret_cast<const float2*>(S + b_rd + s8 * BN + hf * 64);
        bf[2 * hf][s8] = b.x; bf[2 * hf + 1][s8] = b.y;
      }
    if (MODE == 0) {
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s8], bf[j][s8], acc[i][j], 0, 0, 0);
    } else {
      u32x4 ah[2], am[2], al[2], bh[4], bm[4], bl[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) split8(af[i], ah[i], am[i], al[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) split8(bf[j], bh[j], bm[j], bl[j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x16 c = acc[i][j];
          c = mma(al[i], bh[j], c); c = mma(ah[i], bl[j], c); c = mma(am[i], bm[j], c);
          c = mma(am[i], bh[j], c); c = mma(ah[i], bm[j], c); c = mma(ah[i], bh[j], c);
          acc[i][j] = c;
        }
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
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int col = n0 + wn * 128 + hf * 64 + 2 * l31;
        if (row < p.M && col < p.N)
          *reinterpret_cast<float2*>(Cb + (size_t)row * p.ldc + col) = make_float2(acc[i][2 * hf][e], acc[i][2 * hf + 1][e]);
      }
#endif
}


// ---- variant: BOTH operands arrive pre-cut (A: three bf16 planes [plane][M][K]; B as in gemm_bpre): no VALU in the loop at all.
// Upper bound of "producers write activations and weights pre-cut".  LDS per stage: A 12 KB + B 12 KB.
struct LabR { LabQ q; const unsigned short* As; size_t as_plane, as_batch; unsigned as_bytes; };
template <int NSTQ>
__global__ __launch_bounds__(256, 2)
void gemm_allpre(LabR rr) {
#if defined(__HIP_DEVICE_COMPILE__)
  const LabQ& q = rr.q; const LabP& p = q.p;
  constexpr int BM = 128, BN = 128, NST = NSTQ;
  constexpr int A_BYTES = 3 * BM * 32, B_BYTES = 3 * 2 * BN * 16, ST_BYTES = A_BYTES + B_BYTES;
  constexpr int AI = 3, BI = 3;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int tile = xcd_swz(blockIdx.x, p.ntiles);
  const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  float* Cb = p.C + (size_t)blockIdx.z * p.c_bs;
  const i32x4 rsA = make_rsrc(rr.As + (size_t)blockIdx.z * rr.as_batch, rr.as_bytes);
  const i32x4 rsB = make_rsrc(q.Bs + (size_t)blockIdx.z * q.bs_batch, q.bs_bytes);
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
  unsigned a_voff[AI], b_voff[BI];
#pragma unroll
  for (int r = 0; r < AI; ++r) {       // instruction x in [0,12): plane = x/4, rows (x%4)*32 + lane/2, k-half lane%2
    const int x = wid * AI + r;
    const int plane = x >> 2, row = (x & 3) * 32 + (lane >> 1), half = lane & 1;
    const int gm = m0 + row;
    a_voff[r] = gm < p.M ? (unsigned)((size_t)plane * rr.as_plane * 2 + (size_t)gm * p.K * 2 + half * 16) : 0x80000000u;
  }
#pragma unroll
  for (int r = 0; r < BI; ++r) {
    const int x = wid * BI + r;
    const int plane = x >> 2, g = (x >> 1) & 1, half = x & 1;
    const int nn = n0 + half * 64 + lane;
    b_voff[r] = nn < p.N ? (unsigned)((size_t)plane * q.bs_plane * 2 + ((size_t)g * p.N + nn) * 16) : 0x80000000u;
  }
  auto issue = [&](int st, int kb) {
    const unsigned As = lds0 + (unsigned)(st * ST_BYTES), Bs = As + A_BYTES;
#pragma unroll
    for (int r = 0; r < AI; ++r) lds_dma16(a_voff[r], rsA, (unsigned)kb * 32u, As + (unsigned)(wid * AI + r) * 1024u);
#pragma unroll
    for (int r = 0; r < BI; ++r)
      lds_dma16(b_voff[r], rsB, (unsigned)kb * 2u * (unsigned)p.N * 16u, Bs + (unsigned)(wid * BI + r) * 1024u);
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int h = lane >> 5, l31 = lane & 31;
  auto compute = [&](int st) {
    const char* SA = reinterpret_cast<const char*>(smem) + st * ST_BYTES;
    const char* SB = SA + A_BYTES;
    u32x4 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wm * 64 + i * 32 + l31;
      ah[i] = *reinterpret_cast<const u32x4*>(SA + ((0 * 128 + row) * 2 + h) * 16);
      am[i] = *reinterpret_cast<const u32x4*>(SA + ((1 * 128 + row) * 2 + h) * 16);
      al[i] = *reinterpret_cast<const u32x4*>(SA + ((2 * 128 + row) * 2 + h) * 16);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wn * 64 + 2 * l31 + j;
      bh[j] = *reinterpret_cast<const u32x4*>(SB + ((0 * 2 + h) * 128 + col) * 16);
      bm[j] = *reinterpret_cast<const u32x4*>(SB + ((1 * 2 + h) * 128 + col) * 16);
      bl[j] = *reinterpret_cast<const u32x4*>(SB + ((2 * 2 + h) * 128 + col) * 16);
    }
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
  const int nkb = p.K / 16;
  issue(0, 0);
  if (NST == 3 && nkb > 1) issue(1, 1);
  int st = 0;
  for (int kb = 0; kb < nkb; ++kb) {
    if (NST == 3) {
      if (kb + 1 < nkb) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(AI + BI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int st2 = st + 2; if (st2 >= NST) st2 -= NST;
      if (kb + 2 < nkb) issue(st2, kb + 2);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kb + 1 < nkb) issue(st ^ 1, kb + 1);
    }
    compute(st);
    if (NST == 3) st = st + 1 == NST ? 0 : st + 1; else st ^= 1;
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


// ---- variant: software-pipelined split loop.  4-slot ring (64 KB: 2 workgroups / CU = 2 waves per SIMD, where the bf16 MFMA
// issues at full rate).  Iteration k: wait for stage k+1, barrier, issue stage k+3, read stage k+1's fragments, then issue the
// 24 MFMAs of stage k (operands cut in the previous iteration) INTERLEAVED with the cut of stage k+1 -- the VALU work rides
// in the shadow of the matrix pipe inside the wave instead of in front of it.
struct Cut { u32x4 ah[2], am[2], al[2], bh[2], bm[2], bl[2]; };
template <int SCHED>
__global__ __launch_bounds__(256, 2)
void gemm_pipe(LabP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int WGM = 2;
  constexpr int NW = 2 * WGM, BM = 64 * WGM, BN = 128, BK = 16, NST = 4;
  constexpr int A_FL = BM * BK, B_FL = BK * BN, ST_FL = A_FL + B_FL;
  constexpr int AI = (BM / 16) / NW;
  constexpr int BI = 8 / NW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
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
    const int krow = 2 * (wid * BI + r) + (lane >> 5);
    const int nn = n0 + 4 * (lane & 31);
    b_voff[r] = nn < p.N ? (unsigned)(krow * p.ldb + nn) * 4u : 0x80000000u;
  }
  auto issue = [&](int st, int kb) {
    const unsigned As = lds0 + (unsigned)(st * ST_FL) * 4u, Bs = As + A_FL * 4u;
#pragma unroll
    for (int r = 0; r < AI; ++r) lds_dma16(a_voff[r], rsA, (unsigned)kb * (BK * 4), As + (unsigned)(wid * AI + r) * 1024u);
#pragma unroll
    for (int r = 0; r < BI; ++r)
      lds_dma16(b_voff[r], rsB, (unsigned)kb * (BK * 4) * (unsigned)p.ldb, Bs + (unsigned)(wid * BI + r) * 1024u);
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
  auto read_frags = [&](int st, float (&af)[2][8], float (&bf)[2][8]) {
    const float* S = smem + st * ST_FL;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 v0 = *reinterpret_cast<const float4*>(S + a_rd + i * 32 * BK + a_c0);
      const float4 v1 = *reinterpret_cast<const float4*>(S + a_rd + i * 32 * BK + a_c1);
      af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w;
      af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float2 b = *reinterpret_cast<const float2*>(S + b_rd + s * BN);
      bf[0][s] = b.x; bf[1][s] = b.y;
    }
  };
  auto cut = [&](const float (&af)[2][8], const float (&bf)[2][8], Cut& c) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { split8(af[i], c.ah[i], c.am[i], c.al[i]); split8(bf[i], c.bh[i], c.bm[i], c.bl[i]); }
  };
  auto mfmas = [&](const Cut& c) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 x = acc[i][j];
        x = mma(c.al[i], c.bh[j], x); x = mma(c.ah[i], c.bl[j], x); x = mma(c.am[i], c.bm[j], x);
        x = mma(c.am[i], c.bh[j], x); x = mma(c.ah[i], c.bm[j], x); x = mma(c.ah[i], c.bh[j], x);
        acc[i][j] = x;
      }
  };
  const int nkb = p.K / BK;
  // prologue: stages 0, 1, 2 in flight; stage 0's fragments cut
  issue(0, 0);
  if (nkb > 1) issue(1, 1);
  if (nkb > 2) issue(2, 2);
  if (nkb > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * (AI + BI)) : "memory");
  else if (nkb > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(AI + BI) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  Cut cur, nxt;
  {
    float af[2][8], bf[2][8];
    read_frags(0, af, bf);
    cut(af, bf, cur);
  }
  for (int kb = 0; kb < nkb; ++kb) {
    const bool more = kb + 1 < nkb;
    if (more) {
      // stage kb+1 has landed for this wave when at most the loads of stage kb+2 are outstanding
      if (kb + 2 < nkb) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(AI + BI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();            // ... and for every wave; everybody has read stage kb (previous iteration)
      asm volatile("" ::: "memory");
      if (kb + 3 < nkb) issue((kb + 3) & 3, kb + 3);       // slot of stage kb-1: last read two iterations ago
      float af[2][8], bf[2][8];
      read_frags((kb + 1) & 3, af, bf);
      mfmas(cur);
      cut(af, bf, nxt);
      if (SCHED) {
#pragma unroll
        for (int g = 0; g < 24; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 7, 0); }
      }
      cur = nxt;
    } else {
      mfmas(cur);
    }
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


// ---- variant: v_mfma_f32_16x16x32_bf16 (no issue-rate cliff at 3 waves per SIMD, tools/mfma_dep_probe).  Workgroup tile
// 128 x 64, 4 waves (2 x 2), wave tile 64 x 32 = 4 x 2 blocks of 16 x 16 (8 accumulators of 4 registers); 32-k stages, double
// buffered: (128 + 64) x 32 x 4 B = 24 KB per stage, 48 KB per workgroup -> 3 workgroups / CU.
//   A in LDS: rows of 32 floats (8 chunks of 16 B), physical chunk = logical ^ ((row >> 1) & 7): the 16 lanes of a
//   ds_read_b128 group (16 consecutive rows, same logical chunk) hit 16 distinct 4-bank groups.
//   B in LDS: [32 k][64 n] row-major; a lane reads float2 = columns (2c, 2c+1) of its wave's 32 for each of its 8 k.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mma16(u32x4 a, u32x4 b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int MODE>
__global__ __launch_bounds__(256, 3)
void gemm_m16(LabP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 128, BN = 64, BK = 32, NST = 2;
  constexpr int A_FL = BM * BK, B_FL = BK * BN, ST_FL = A_FL + B_FL;      // 4096 + 2048 floats
  constexpr int AI = 4, BI = 2;                                          // 16 + 8 KB per stage / 4 waves / 1 KB
  extern __shared__ __attribute__((aligned(16))) float smem[];
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
  for (int r = 0; r < AI; ++r) {            // instruction x = wid*4 + r covers rows 8x .. 8x+7; lane -> (row, physical chunk)
    const int row = 8 * (wid * AI + r) + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);          // logical chunk stored at physical slot lane & 7
    const int gm = m0 + row;
    a_voff[r] = gm < p.M ? (unsigned)(gm * p.lda + 4 * c) * 4u : 0x80000000u;
  }
#pragma unroll
  for (int r = 0; r < BI; ++r) {            // instruction x = wid*2 + r covers k rows 4x .. 4x+3 (64 floats = 256 B each)
    const int krow = 4 * (wid * BI + r) + (lane >> 4);
    const int nn = n0 + 4 * (lane & 15);
    b_voff[r] = nn < p.N ? (unsigned)(krow * p.ldb + nn) * 4u : 0x80000000u;
  }
  auto issue = [&](int st, int kb) {
    const unsigned As = lds0 + (unsigned)(st * ST_FL) * 4u, Bs = As + A_FL * 4u;
#pragma unroll
    for (int r = 0; r < AI; ++r) lds_dma16(a_voff[r], rsA, (unsigned)kb * (BK * 4), As + (unsigned)(wid * AI + r) * 1024u);
#pragma unroll
    for (int r = 0; r < BI; ++r)
      lds_dma16(b_voff[r], rsB, (unsigned)kb * (BK * 4) * (unsigned)p.ldb, Bs + (unsigned)(wid * BI + r) * 1024u);
  };
  f32x4v acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
  const int r16 = lane & 15, g = lane >> 4;          // row / column inside a block, k-group (k = 8g .. 8g+7)
  auto compute = [&](int st) {
    const float* S = smem + st * ST_FL;
    float af[4][8], bf[2][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wm * 64 + i * 16 + r16;
      const int sw = (row >> 1) & 7;
      const float4 v0 = *reinterpret_cast<const float4*>(S + row * BK + (((2 * g) ^ sw) * 4));
      const float4 v1 = *reinterpret_cast<const float4*>(S + row * BK + (((2 * g + 1) ^ sw) * 4));
      af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w;
      af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
    }
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const float2 b = *reinterpret_cast<const float2*>(S + A_FL + (8 * g + s8) * BN + wn * 32 + 2 * r16);
      bf[0][s8] = b.x; bf[1][s8] = b.y;
    }
    u32x4 ah[4], am[4], al[4], bh[2], bm[2], bl[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) split8(af[i], ah[i], am[i], al[i]);
#pragma unroll
    for (int j = 0; j < 2; ++j) split8(bf[j], bh[j], bm[j], bl[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4v c = acc[i][j];
        if (MODE == 1) { c = mma16(al[i], bh[j], c); c = mma16(ah[i], bl[j], c); c = mma16(am[i], bm[j], c); }
        c = mma16(am[i], bh[j], c); c = mma16(ah[i], bm[j], c); c = mma16(ah[i], bh[j], c);
        acc[i][j] = c;
      }
  };
  const int nkb = p.K / BK;
  issue(0, 0);
  int st = 0;
  for (int kb = 0; kb < nkb; ++kb) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kb + 1 < nkb) issue(st ^ 1, kb + 1);
    compute(st);
    st ^= 1;
  }
  // C layout of the 16x16 block: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = m0 + wm * 64 + i * 16 + g * 4 + e;
      const int col = n0 + wn * 32 + 2 * r16;
      if (row < p.M && col < p.N)
        *reinterpret_cast<float2*>(Cb + (size_t)row * p.ldc + col) = make_float2(acc[i][0][e], acc[i][1][e]);
    }
#endif
}

__global__ void gemm_ref(LabP p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)p.M * p.N) return;
  const int m = (int)(i / p.N), n = (int)(i % p.N);
  const float* A = p.A + (size_t)blockIdx.z * p.a_bs + (size_t)m * p.lda;
  const float* B = p.B + (size_t)blockIdx.z * p.b_bs + n;
  double acc = 0;
  for (int k = 0; k < p.K; ++k) acc += (double)A[k] * B[(size_t)k * p.ldb];
  p.C[(size_t)blockIdx.z * p.c_bs + (size_t)m * p.ldc + n] = (float)acc;
}


struct Shape { const char* name; int M, N, K, batch; };
// full 24-bit mantissas, magnitudes spread over ~2^8 (a 16-bit-quantised fill would hide the split error)
static void fill(std::vector<float>& v, unsigned seed) {
  unsigned s = seed * 2654435761u + 12345u;
  for (auto& x : v) {
    s = s * 1664525u + 1013904223u; const float m = ((s >> 8) & 0xFFFFFF) / 8388608.0f - 1.0f;
    s = s * 1664525u + 1013904223u; x = std::ldexp(m, -(int)((s >> 24) & 7));
  }
}
template <int MODE>
static void launch(hipStream_t st, LabP p, int batch) {
  constexpr int smem = 3 * (128 * 16 + 16 * 128) * 4;
  static bool once = false;
  if (!once) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_v1<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, smem)); once = true; }
  p.tiles_n = (p.N + 127) / 128;
  p.ntiles = ((p.M + 127) / 128) * p.tiles_n;
  hipLaunchKernelGGL((gemm_v1<MODE>), dim3(p.ntiles, 1, batch), dim3(256), smem, st, p);
}
int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  std::vector<Shape> shapes = {
      {"wino_resblock  (36 planes 512x1024x1024)", 512, 1024, 1024, 36},
      {"down4 (8192x512x4096)", 8192, 512, 4096, 1},
      {"down2 (131072x128x1024)", 131072, 128, 1024, 1},
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
    LabP p{};
    p.A = dA; p.B = dB; p.C = dR; p.M = s.M; p.N = s.N; p.K = s.K; p.lda = s.K; p.ldb = s.N; p.ldc = s.N;
    p.a_bs = (size_t)s.M * s.K; p.b_bs = (size_t)s.K * s.N; p.c_bs = (size_t)s.M * s.N;
    p.a_bytes = (unsigned)((size_t)s.M * s.K * 4); p.b_bytes = (unsigned)((size_t)s.K * s.N * 4);
    hipLaunchKernelGGL(gemm_ref, dim3((unsigned)(((size_t)s.M * s.N + 255) / 256), 1, s.batch), dim3(256), 0, st, p);
    CK(hipStreamSynchronize(st));
    p.C = dC;
    const double flops = 2.0 * s.M * s.N * s.K * s.batch;
    printf("== %s\n", s.name);
    std::vector<float> r(nc); CK(hipMemcpy(r.data(), dR, nc * 4, hipMemcpyDeviceToHost));
    auto run = [&](const char* what, auto&& fn) {
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
      printf("   %-12s rel-L2 vs fp64-accumulated reference %.3e   %8.3f ms  %7.1f fp32-equivalent TFLOP/s\n", what, std::sqrt(num / den), best, flops / best * 1e-9);
    };
    // pre-cut B: [batch][plane][K/8][N][8] bf16 (truncation split, exact)
    std::vector<unsigned short> hs((size_t)3 * nb);
    const size_t plane = (size_t)s.K * s.N;
    for (int z = 0; z < s.batch; ++z)
      for (int k = 0; k < s.K; ++k)
        for (int n = 0; n < s.N; ++n) {
          const float x = hb[(size_t)z * plane + (size_t)k * s.N + n];
          unsigned u; memcpy(&u, &x, 4);
          unsigned uh = u & 0xffff0000u; float fh; memcpy(&fh, &uh, 4);
          const float r = x - fh; unsigned ur; memcpy(&ur, &r, 4);
          unsigned um = ur & 0xffff0000u; float fm; memcpy(&fm, &um, 4);
          const float q2 = r - fm; unsigned ul; memcpy(&ul, &q2, 4);
          const size_t o = ((size_t)(k / 8) * s.N + n) * 8 + (k % 8);
          hs[((size_t)z * 3 + 0) * plane + o] = (unsigned short)(u >> 16);
          hs[((size_t)z * 3 + 1) * plane + o] = (unsigned short)(ur >> 16);
          hs[((size_t)z * 3 + 2) * plane + o] = (unsigned short)(ul >> 16);
        }
    unsigned short* dS; CK(hipMalloc((void**)&dS, hs.size() * 2));
    CK(hipMemcpy(dS, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    LabQ q{}; q.Bs = dS; q.bs_plane = plane; q.bs_batch = 3 * plane; q.bs_bytes = (unsigned)(3 * plane * 2);
    auto launch_bpre = [&](auto kern, int nst) {
      const int smem = nst * (128 * 16 * 4 + 3 * 2 * 128 * 16);
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
      q.p = p; q.p.tiles_n = (p.N + 127) / 128; q.p.ntiles = ((p.M + 127) / 128) * q.p.tiles_n;
      hipLaunchKernelGGL(kern, dim3(q.p.ntiles, 1, s.batch), dim3(256), smem, st, q);
    };
    // pre-cut A: [batch][plane][M][K] bf16
    std::vector<unsigned short> hsa((size_t)3 * na);
    const size_t aplane = (size_t)s.M * s.K;
    for (int z = 0; z < s.batch; ++z)
      for (size_t e = 0; e < aplane; ++e) {
        const float x = ha[(size_t)z * aplane + e];
        unsigned u; memcpy(&u, &x, 4);
        unsigned uh = u & 0xffff0000u; float fh; memcpy(&fh, &uh, 4);
        const float r = x - fh; unsigned ur; memcpy(&ur, &r, 4);
        unsigned um = ur & 0xffff0000u; float fm; memcpy(&fm, &um, 4);
        const float q2 = r - fm; unsigned ul; memcpy(&ul, &q2, 4);
        hsa[((size_t)z * 3 + 0) * aplane + e] = (unsigned short)(u >> 16);
        hsa[((size_t)z * 3 + 1) * aplane + e] = (unsigned short)(ur >> 16);
        hsa[((size_t)z * 3 + 2) * aplane + e] = (unsigned short)(ul >> 16);
      }
    unsigned short* dSA; CK(hipMalloc((void**)&dSA, hsa.size() * 2));
    CK(hipMemcpy(dSA, hsa.data(), hsa.size() * 2, hipMemcpyHostToDevice));
    auto launch_all = [&](auto kern, int nst) {
      const int smem = nst * (3 * 128 * 32 + 3 * 2 * 128 * 16);
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
      LabR rr{}; rr.q = q; rr.q.p = p; rr.q.p.tiles_n = (p.N + 127) / 128; rr.q.p.ntiles = ((p.M + 127) / 128) * rr.q.p.tiles_n;
      rr.As = dSA; rr.as_plane = aplane; rr.as_batch = 3 * aplane; rr.as_bytes = (unsigned)(3 * aplane * 2);
      hipLaunchKernelGGL(kern, dim3(rr.q.p.ntiles, 1, s.batch), dim3(256), smem, st, rr);
    };
    auto launch_w = [&](auto kern) {
      const int smem = 3 * (128 * 16 + 16 * 256) * 4;
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
      LabP w = p; w.tiles_n = (p.N + 255) / 256; w.ntiles = ((p.M + 127) / 128) * w.tiles_n;
      hipLaunchKernelGGL(kern, dim3(w.ntiles, 1, s.batch), dim3(256), smem, st, w);
    };
    for (int rnd = 0; rnd < 2; ++rnd) {

      run("fp32 mfma", [&] { launch<0>(st, p, s.batch); });
      run("bf16 x6", [&] { launch<1>(st, p, s.batch); });
      run("x6 16x16x32", [&] {
        const int smem = 2 * (128 * 32 + 32 * 64) * 4;
        static bool once = false;
        if (!once) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_m16<1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem)); once = true; }
        LabP w = p; w.tiles_n = (p.N + 63) / 64; w.ntiles = ((p.M + 127) / 128) * w.tiles_n;
        hipLaunchKernelGGL((gemm_m16<1>), dim3(w.ntiles, 1, s.batch), dim3(256), smem, st, w);
      });



    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dR)); CK(hipFree(dS)); CK(hipFree(dSA));
  }
  return 0;
}
