// tile_lab (round 5) -- what bounds the two-plane ring loop, and the CU-level tile that answers it.
// C[z][m][n] = sum_k A[z][m][k] B[z][k][n] with both operands pre-cut into two fp16 planes (x = h + l; 3 MFMAs per product), the
// arithmetic conv_fwd_pc_kernel<.., PL = 2, APAIR> ships.  One kernel template, waves arranged WGM x WGN, each wave MR x NB
// fragments of 32 x 32; 16-k LDS stages in a ring of NSS, filled by buffer_load ... lds in 1-KiB pieces.
//   MODE 0 the loop        MODE 1 no MFMAs (fills + fragment reads)      MODE 2 no fills after the prologue (reads + MFMAs)
//   MODE 3 fills only (no fragment reads, no MFMAs)    MODE 4 MFMAs only (fragments read once, registers thereafter; no barrier)
// Layouts: Ap[stage = k/16][tile_m][kq 2][plane 2][row BM][8] f16, Bp[stage][tile_n][kq 2][plane 2][pos BN][8] f16 with
// pos = (n % NBc) * 32 + n / NBc, NBc = BN / 32 (the lane at position l of column block j owns column NBc l + j).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tile_lab.hip -o tools/tile_lab
#include <hip/hip_runtime.h>

#include <algorithm>
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
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct LabW {
  const unsigned short* Ap; const unsigned short* Bp; float* C;
  int M, N, K, ldc;
  size_t ap_bs, bp_bs, c_bs;      // per batch element (f16 elements / f16 elements / floats)
  int tiles_m, tiles_n, ntiles;
  unsigned ap_bytes, bp_bytes;
  int zswz;
  unsigned long long* clk;      // {s_memtime delta, s_memrealtime delta} of up to 16 workgroups -> effective shader clock
};

__device__ __forceinline__ i32x4 make_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long a = (unsigned long long)ptr;
  i32x4 r;
  r[0] = (int)(unsigned)(a & 0xffffffffull); r[1] = (int)(unsigned)((a >> 32) & 0xffffull); r[2] = (int)bytes; r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void lds_dma16(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" : : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ int xcd_swz(int bid, int n) {
  const int q = n >> 3, r = n & 7, xcd = bid & 7, i = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}
template <int MODE>
__device__ __forceinline__ f32x16 mma_h(u32x4 a, u32x4 b, f32x16 c) {
  if constexpr (MODE == 1) { c[0] += __uint_as_float((a[0] ^ b[1]) & 0x3fffffffu); c[1] += __uint_as_float((a[2] ^ b[3]) & 0x3fffffffu); return c; }
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

template <int WGM, int WGN, int MR, int NB, int NSS, int WPS, int MODE, int PIPE = 0>
__global__ __launch_bounds__(64 * WGM * WGN, WPS)
void gemm_w(LabW p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = WGM * WGN, BM = 32 * WGM * MR, BN = 32 * NB * WGN;
  constexpr int A_BYTES = 4 * BM * 16, B_BYTES = 4 * BN * 16, ST_BYTES = A_BYTES + B_BYTES;
  constexpr int APC = A_BYTES / 1024, PC = ST_BYTES / 1024, PPW = PC / NW;
  static_assert(PC % NW == 0, "pieces divide over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long ck0 = 0, ck1 = 0;
  const bool ckme = p.clk && (blockIdx.x % 13) == 0 && blockIdx.x / 13 < 16 && threadIdx.x == 0;
  if (ckme) { ck0 = __builtin_readcyclecounter(); ck1 = __builtin_amdgcn_s_memrealtime(); }
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wid / WGN, wn = wid - wm * WGN;
  // XCD-aware order over ALL tiles of the launch (as the product does): an XCD owns a contiguous run of (plane, tile) pairs
  const int gtile = p.zswz ? xcd_swz(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  int z = gtile / p.ntiles, tile = gtile - z * p.ntiles;
  if (!p.zswz) tile = xcd_swz(tile, p.ntiles);                       // round-5 first lab run: swizzle inside a plane only
  const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
  float* Cb = p.C + (size_t)z * p.c_bs;
  const i32x4 rsA = make_rsrc(p.Ap + (size_t)z * p.ap_bs, p.ap_bytes);
  const i32x4 rsB = make_rsrc(p.Bp + (size_t)z * p.bp_bs, p.bp_bytes);
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
  const unsigned voff = (unsigned)lane * 16u;
  const unsigned a_stage = (unsigned)p.tiles_m * A_BYTES, a_tile = (unsigned)tile_m * A_BYTES;
  const unsigned b_stage = (unsigned)p.tiles_n * B_BYTES, b_tile = (unsigned)tile_n * B_BYTES;
  auto issue = [&](int ss, int kb) {
    const unsigned S = lds0 + (unsigned)(ss * ST_BYTES);
#pragma unroll
    for (int r = 0; r < PPW; ++r) {
      const int q = wid * PPW + r;                       // piece of the stage image: [A pieces | B pieces]
      if (q < APC) lds_dma16(voff, rsA, (unsigned)kb * a_stage + a_tile + (unsigned)q * 1024u, S + (unsigned)q * 1024u);
      else lds_dma16(voff, rsB, (unsigned)kb * b_stage + b_tile + (unsigned)(q - APC) * 1024u, S + (unsigned)q * 1024u);
    }
  };
  f32x16 acc[MR][NB];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int h = lane >> 5, l31 = lane & 31;
  const int a_rd = (h * 2 * BM + wm * MR * 32 + l31) * 16;                   // + (plane * BM + 32 i) * 16
  const int b_rd = A_BYTES + (h * 2 * BN + wn * NB * 32 + l31) * 16;         // + (plane * BN + 32 j) * 16
  auto compute = [&](int ss) {
    const char* S = smem + ss * ST_BYTES;
    u32x4 ah[MR], al[MR], bh[NB], bl[NB];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      ah[i] = *reinterpret_cast<const u32x4*>(S + a_rd + (0 * BM + 32 * i) * 16);
      al[i] = *reinterpret_cast<const u32x4*>(S + a_rd + (1 * BM + 32 * i) * 16);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      bh[j] = *reinterpret_cast<const u32x4*>(S + b_rd + (0 * BN + 32 * j) * 16);
      bl[j] = *reinterpret_cast<const u32x4*>(S + b_rd + (1 * BN + 32 * j) * 16);
    }
    // three terms, each over all MR x NB accumulators: dependent MFMAs on one accumulator are MR * NB issues apart
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = mma_h<MODE>(al[i], bh[j], acc[i][j]);
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = mma_h<MODE>(ah[i], bl[j], acc[i][j]);
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = mma_h<MODE>(ah[i], bh[j], acc[i][j]);
  };
  const int nss = p.K / 16;
#pragma unroll
  for (int s = 0; s < NSS - 1; ++s)
    if (s < nss) issue(s, s);
  if constexpr (MODE == 4) {
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* S = smem;
    u32x4 ah[MR], al[MR], bh[NB], bl[NB];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      ah[i] = *reinterpret_cast<const u32x4*>(S + a_rd + (0 * BM + 32 * i) * 16);
      al[i] = *reinterpret_cast<const u32x4*>(S + a_rd + (1 * BM + 32 * i) * 16);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      bh[j] = *reinterpret_cast<const u32x4*>(S + b_rd + (0 * BN + 32 * j) * 16);
      bl[j] = *reinterpret_cast<const u32x4*>(S + b_rd + (1 * BN + 32 * j) * 16);
    }
    for (int kb = 0; kb < nss; ++kb) {
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = mma_h<0>(al[i], bh[j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = mma_h<0>(ah[i], bl[j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = mma_h<0>(ah[i], bh[j], acc[i][j]);
      asm volatile("" ::: "memory");
    }
  } else if constexpr (PIPE == 0) {
    int ss = 0;
    for (int kb = 0; kb < nss; ++kb) {
      if (MODE != 2 || kb < NSS - 1) {
        const int younger = MODE == 2 ? 0 : min(NSS - 2, nss - 1 - kb);
        if (younger >= 3) wait_vm<3 * PPW>();
        else if (younger == 2) wait_vm<2 * PPW>();
        else if (younger == 1) wait_vm<PPW>();
        else wait_vm<0>();
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int sn = ss + NSS - 1; if (sn >= NSS) sn -= NSS;
      if (MODE != 2 && kb + NSS - 1 < nss) issue(sn, kb + NSS - 1);
      if (MODE != 3) compute(ss);
      ss = ss + 1 == NSS ? 0 : ss + 1;
    }
  } else {
    // software-pipelined form: the fragments of stage kb + 1 are read (into the other register set) right after the barrier that
    // publishes that stage, and the MFMAs of stage kb -- whose fragments were read one iteration earlier -- run under those reads.
    static_assert(NSS >= 3, "one landed stage + one in registers + at least one in flight");
    struct Frag { u32x4 ah[MR], al[MR], bh[NB], bl[NB]; };
    auto rd = [&](Frag& f, int ss) {
      const char* S = smem + ss * ST_BYTES;
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        f.ah[i] = *reinterpret_cast<const u32x4*>(S + a_rd + (0 * BM + 32 * i) * 16);
        f.al[i] = *reinterpret_cast<const u32x4*>(S + a_rd + (1 * BM + 32 * i) * 16);
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        f.bh[j] = *reinterpret_cast<const u32x4*>(S + b_rd + (0 * BN + 32 * j) * 16);
        f.bl[j] = *reinterpret_cast<const u32x4*>(S + b_rd + (1 * BN + 32 * j) * 16);
      }
    };
    auto mm = [&](const Frag& f) {
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = mma_h<MODE>(f.al[i], f.bh[j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = mma_h<MODE>(f.ah[i], f.bl[j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = mma_h<MODE>(f.ah[i], f.bh[j], acc[i][j]);
    };
    Frag F, G;
    {
      const int younger = min(NSS - 2, nss - 1);
      if (younger >= 2) wait_vm<2 * PPW>(); else if (younger == 1) wait_vm<PPW>(); else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      rd(F, 0);
    }
    int ss = 0;                                   // slot of stage kb
    auto step = [&](Frag& cur, Frag& nxt, int kb) {
      int s1 = ss + 1; if (s1 >= NSS) s1 -= NSS;
      if (kb + 1 < nss) {
        const int younger = min(NSS - 3, nss - 2 - kb);     // stages younger than kb + 1 that may still be in flight
        if (younger >= 2) wait_vm<2 * PPW>(); else if (younger == 1) wait_vm<PPW>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int sp = ss - 1; if (sp < 0) sp += NSS;             // slot of stage kb - 1: every wave has consumed it
        if (kb + NSS - 1 < nss && kb >= 1) issue(sp, kb + NSS - 1);
        rd(nxt, s1);
      }
      mm(cur);
      ss = s1;
    };
    // (the first iteration has no free slot yet: stages 0 .. NSS - 2 fill all but one slot, stage NSS - 1 goes into the last one)
    if (NSS - 1 < nss) { int sl = NSS - 1; issue(sl, NSS - 1); }
    for (int kb = 0; kb < nss; kb += 2) {
      step(F, G, kb);
      if (kb + 1 < nss) step(G, F, kb + 1);
    }
  }
  constexpr int NBc = BN / 32;
  const int col = tile_n * BN + NBc * l31 + wn * NB;
  if (col < p.N) {
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = tile_m * BM + (wm * MR + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row >= p.M) continue;
        float* dst = Cb + (size_t)row * p.ldc + col;
        if constexpr (NB == 4) *reinterpret_cast<float4*>(dst) = make_float4(acc[i][0][e], acc[i][1][e], acc[i][2][e], acc[i][3][e]);
        else if constexpr (NB == 2) *reinterpret_cast<float2*>(dst) = make_float2(acc[i][0][e], acc[i][1][e]);
        else {
#pragma unroll
          for (int j = 0; j < NB; ++j) dst[j] = acc[i][j][e];
        }
      }
  }
  if (ckme) { p.clk[2 * (blockIdx.x / 13)] = __builtin_readcyclecounter() - ck0; p.clk[2 * (blockIdx.x / 13) + 1] = __builtin_amdgcn_s_memrealtime() - ck1; }
#endif
}


// ---- round 6: the ragged last round.  36 planes = 1152 tiles on 1024 slots: the 128 tiles of the second round run ONE workgroup per
// CU, and a lone two-stage ring waits a whole fill latency per 16-k step (fills only: 52.8 us at 1024 tiles, 125 us at 1152).  Here the
// launch is `slots` workgroups; workgroup u runs whole tile u and then, if u < rem * tail_s, one K-slice (1 / tail_s) of a remainder
// tile -- all four workgroups of a CU stay busy to the end.  A slice's partial tile goes to a slab; the LAST slice of a tile to arrive
// (agent-scope counter) sums the tail_s slabs in slice order (fixed: bit-reproducible) and stores the tile.  XLOC: the slices of one
// remainder tile sit on one XCD (u % 8).
struct FuseP { int slots, rem, tail_s, per_split, xloc; float* slab; int* counters; int fence; };
template <int MODE>
__global__ __launch_bounds__(256, 4)
void gemm_fused(LabW p, FuseP f) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NB = 4, BM = 128, BN = 128;
  constexpr int A_BYTES = 4 * BM * 16, B_BYTES = 4 * BN * 16, ST_BYTES = A_BYTES + B_BYTES;
  constexpr int APC = A_BYTES / 1024, PC = ST_BYTES / 1024, PPW = PC / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_last;
  unsigned long long ck0 = 0, ck1 = 0;
  const bool ckme = p.clk && (blockIdx.x % 13) == 0 && blockIdx.x / 13 < 16 && threadIdx.x == 0;
  if (ckme) { ck0 = __builtin_readcyclecounter(); ck1 = __builtin_amdgcn_s_memrealtime(); }
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
  const unsigned voff = (unsigned)lane * 16u;
  const int h = lane >> 5, l31 = lane & 31;
  const int a_rd = (h * 2 * BM + wid * 32 + l31) * 16;
  const int b_rd = A_BYTES + (h * 2 * BN + l31) * 16;
  const int nss_all = p.K / 16;
  const int u = blockIdx.x;
  for (int item = 0; item < 2; ++item) {
    int gtile, kb0 = 0, kb1 = nss_all, tt = 0, piece = 0;
    if (item == 0) gtile = xcd_swz(u, f.slots);
    else {
      if (u >= f.rem * f.tail_s) break;
      if (f.xloc) { tt = (u & 7) + 8 * (u / (8 * f.tail_s)); piece = (u >> 3) % f.tail_s; }
      else { tt = u / f.tail_s; piece = u - tt * f.tail_s; }
      if (tt >= f.rem) break;
      gtile = f.slots + tt;
      kb0 = piece * f.per_split; kb1 = min(nss_all, kb0 + f.per_split);
      __syncthreads();                 // everybody is done with the ring (and the epilogue) of the first item
    }
    const int z = gtile / p.ntiles, tile = gtile - z * p.ntiles;
    const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
    float* Cb = p.C + (size_t)z * p.c_bs;
    const i32x4 rsA = make_rsrc(p.Ap + (size_t)z * p.ap_bs, p.ap_bytes);
    const i32x4 rsB = make_rsrc(p.Bp + (size_t)z * p.bp_bs, p.bp_bytes);
    const unsigned a_stage = (unsigned)p.tiles_m * A_BYTES, a_tile = (unsigned)tile_m * A_BYTES;
    const unsigned b_stage = (unsigned)p.tiles_n * B_BYTES, b_tile = (unsigned)tile_n * B_BYTES;
    auto issue = [&](int ss, int kb) {
      const unsigned S = lds0 + (unsigned)(ss * ST_BYTES);
#pragma unroll
      for (int r = 0; r < PPW; ++r) {
        const int q = wid * PPW + r;
        if (q < APC) lds_dma16(voff, rsA, (unsigned)kb * a_stage + a_tile + (unsigned)q * 1024u, S + (unsigned)q * 1024u);
        else lds_dma16(voff, rsB, (unsigned)kb * b_stage + b_tile + (unsigned)(q - APC) * 1024u, S + (unsigned)q * 1024u);
      }
    };
    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    if (kb0 < kb1) {
      issue(0, kb0);
      int ss = 0;
      for (int kb = kb0; kb < kb1; ++kb) {
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kb + 1 < kb1) issue(ss ^ 1, kb + 1);
        if (MODE != 3) {
          const char* S = smem + ss * ST_BYTES;
          u32x4 ah, al, bh[NB], bl[NB];
          ah = *reinterpret_cast<const u32x4*>(S + a_rd + (0 * BM) * 16);
          al = *reinterpret_cast<const u32x4*>(S + a_rd + (1 * BM) * 16);
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            bh[j] = *reinterpret_cast<const u32x4*>(S + b_rd + (0 * BN + 32 * j) * 16);
            bl[j] = *reinterpret_cast<const u32x4*>(S + b_rd + (1 * BN + 32 * j) * 16);
          }
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[j] = mma_h<MODE>(al, bh[j], acc[j]);
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[j] = mma_h<MODE>(ah, bl[j], acc[j]);
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[j] = mma_h<MODE>(ah, bh[j], acc[j]);
        }
        ss ^= 1;
      }
    }
    const int colr = NB * l31;
    if (item == 1) {
      float* slab = f.slab + ((size_t)(tt * f.tail_s + piece) * BM) * BN;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wid * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        *reinterpret_cast<float4*>(slab + (size_t)row * BN + colr) = make_float4(acc[0][e], acc[1][e], acc[2][e], acc[3][e]);
      }
      // release.  FENCE 1: agent-scope fence (buffer_wbl2 + inv of the whole L2: measured ruinous, 422 us).  FENCE 0: the slab lives in
      // UNCACHED device memory (hipDeviceMallocUncached): its stores bypass the L2, so "my stores have completed" (vmcnt 0) is all the
      // release there is to do.
      if (f.fence) __threadfence(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) {
        const int old = __hip_atomic_fetch_add(f.counters + tt, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == f.tail_s - 1;
        if (old == f.tail_s - 1) __hip_atomic_store(f.counters + tt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
      }
      __syncthreads();
      if (!s_last) break;
      if (f.fence) __threadfence();    // acquire: the other slices' slabs (uncached memory: nothing to invalidate)
      const float* s0 = f.slab + ((size_t)(tt * f.tail_s) * BM) * BN;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wid * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        float4 a = *reinterpret_cast<const float4*>(s0 + (size_t)row * BN + colr);
        for (int sp = 1; sp < f.tail_s; ++sp) {
          const float4 b = *reinterpret_cast<const float4*>(s0 + ((size_t)sp * BM + row) * BN + colr);
          a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        acc[0][e] = a.x; acc[1][e] = a.y; acc[2][e] = a.z; acc[3][e] = a.w;
      }
    }
    const int col = tile_n * BN + colr;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = tile_m * BM + wid * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      *reinterpret_cast<float4*>(Cb + (size_t)row * p.ldc + col) = make_float4(acc[0][e], acc[1][e], acc[2][e], acc[3][e]);
    }
  }
  if (ckme) { p.clk[2 * (blockIdx.x / 13)] = __builtin_readcyclecounter() - ck0; p.clk[2 * (blockIdx.x / 13) + 1] = __builtin_amdgcn_s_memrealtime() - ck1; }
#endif
}

__global__ void precut_b_kernel(const float* B, unsigned short* Bp, int K, int N, int BN, size_t b_bs, size_t bp_bs) {
  const int NBc = BN / 32, tiles_n = (N + BN - 1) / BN;
  const size_t total = (size_t)(K / 8) * tiles_n * BN;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int pos = (int)(i % BN); size_t q = i / BN;
  const int tn = (int)(q % tiles_n), kq = (int)(q / tiles_n);
  const int n = tn * BN + (pos % 32) * NBc + pos / 32;
  B += (size_t)blockIdx.y * b_bs; Bp += (size_t)blockIdx.y * bp_bs;
  const size_t base = ((((size_t)(kq >> 1) * tiles_n + tn) * 2 + (kq & 1)) * 2) * BN;
  for (int j = 0; j < 8; ++j) {
    const float x = n < N ? B[(size_t)(kq * 8 + j) * N + n] : 0.f;
    const _Float16 hh = (_Float16)x; const _Float16 ll = (_Float16)(x - (float)hh);
    Bp[(base + 0 * (size_t)BN + pos) * 8 + j] = __builtin_bit_cast(unsigned short, hh);
    Bp[(base + 1 * (size_t)BN + pos) * 8 + j] = __builtin_bit_cast(unsigned short, ll);
  }
}
__global__ void precut_a_kernel(const float* A, unsigned short* Ap, int M, int K, int BM, size_t a_bs, size_t ap_bs) {
  const size_t total = (size_t)(K / 8) * M;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int m = (int)(i % M), kq = (int)(i / M);
  const int tiles_m = M / BM, tm = m / BM, r = m % BM;
  A += (size_t)blockIdx.y * a_bs; Ap += (size_t)blockIdx.y * ap_bs;
  const size_t base = ((((size_t)(kq >> 1) * tiles_m + tm) * 2 + (kq & 1)) * 2) * BM;
  for (int j = 0; j < 8; ++j) {
    const float x = A[(size_t)m * K + kq * 8 + j];
    const _Float16 hh = (_Float16)x; const _Float16 ll = (_Float16)(x - (float)hh);
    Ap[(base + 0 * (size_t)BM + r) * 8 + j] = __builtin_bit_cast(unsigned short, hh);
    Ap[(base + 1 * (size_t)BM + r) * 8 + j] = __builtin_bit_cast(unsigned short, ll);
  }
}
__global__ void gemm_ref(const float* A, const float* B, float* C, int M, int N, int K) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  A += (size_t)blockIdx.z * M * K; B += (size_t)blockIdx.z * K * N; C += (size_t)blockIdx.z * M * N;
  double s = 0;
  for (int k = 0; k < K; ++k) s += (double)A[(size_t)m * K + k] * (double)B[(size_t)k * N + n];
  C[i] = (float)s;
}

struct Shape { const char* name; int M, N, K, batch; };
static void fill(std::vector<float>& v, unsigned seed) {
  unsigned s = seed * 2654435761u + 12345u;
  for (auto& x : v) {
    s = s * 1664525u + 1013904223u; const float m = ((s >> 8) & 0xFFFFFF) / 8388608.0f - 1.0f;
    s = s * 1664525u + 1013904223u; x = std::ldexp(m, -(int)(((s >> 20) & 0xFFF) % 8u));
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  std::vector<int> only_set;
  if (const char* e = getenv("LAB_ONLY")) { std::string z(e); size_t q0 = 0; while (q0 < z.size()) { size_t c = z.find(',', q0); if (c == std::string::npos) c = z.size(); only_set.push_back(atoi(z.substr(q0, c - q0).c_str())); q0 = c + 1; } }
  const int zswz = getenv("LAB_ZSWZ") ? atoi(getenv("LAB_ZSWZ")) : 1;
  const int only_shape = getenv("LAB_SHAPE") ? atoi(getenv("LAB_SHAPE")) : -1;
  std::vector<Shape> shapes = {
      {"exact fit: 32 planes 512x1024x1024", 512, 1024, 1024, 32},
      {"wino_resblock: 36 planes 512x1024x1024", 512, 1024, 1024, 36},
      {"down4 (8192x512x4096)", 8192, 512, 4096, 1},
      {"partial occupancy: 8 planes (256 tiles of 128x128)", 512, 1024, 1024, 8},
      {"partial occupancy: 16 planes (512 tiles)", 512, 1024, 1024, 16},
      {"partial occupancy: 24 planes (768 tiles)", 512, 1024, 1024, 24},
      {"partial occupancy: 4 planes (128 tiles)", 512, 1024, 1024, 4},
      {"down2 (131072x128x1024)", 131072, 128, 1024, 1},
  };
  hipStream_t st; CK(hipStreamCreate(&st));
  unsigned long long* dClk; CK(hipMalloc((void**)&dClk, 32 * sizeof(unsigned long long)));
  int wall_khz = 0; CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (size_t si = 0; si < shapes.size(); ++si) {
    if (only_shape >= 0 && (int)si != only_shape) continue;
    const Shape& s = shapes[si];
    const size_t na = (size_t)s.M * s.K * s.batch, nb = (size_t)s.K * s.N * s.batch, nc = (size_t)s.M * s.N * s.batch;
    std::vector<float> ha(na), hb(nb);
    fill(ha, 1); fill(hb, 2);
    if (getenv("LAB_ZERO") && atoi(getenv("LAB_ZERO"))) { std::fill(ha.begin(), ha.end(), 0.f); std::fill(hb.begin(), hb.end(), 0.f); }
    float *dA, *dB, *dC, *dR;
    CK(hipMalloc((void**)&dA, na * 4)); CK(hipMalloc((void**)&dB, nb * 4)); CK(hipMalloc((void**)&dC, nc * 4)); CK(hipMalloc((void**)&dR, nc * 4));
    CK(hipMemcpy(dA, ha.data(), na * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hb.data(), nb * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(gemm_ref, dim3((unsigned)(((size_t)s.M * s.N + 255) / 256), 1, s.batch), dim3(256), 0, st, dA, dB, dR, s.M, s.N, s.K);
    CK(hipStreamSynchronize(st));
    std::vector<float> r(nc); CK(hipMemcpy(r.data(), dR, nc * 4, hipMemcpyDeviceToHost));
    const double flops = 2.0 * s.M * s.N * s.K * s.batch;
    printf("== %s\n", s.name);
    // operands for 128- / 256-wide tiles on either side
    unsigned short *dPA[2], *dPB[2]; const int Ts[2] = {128, 256};
    const size_t ap_bs = (size_t)s.M * s.K * 2;
    size_t bp_bs[2];
    for (int v = 0; v < 2; ++v) {
      CK(hipMalloc((void**)&dPA[v], ap_bs * s.batch * 2));
      const size_t ta = (size_t)(s.K / 8) * s.M;
      hipLaunchKernelGGL(precut_a_kernel, dim3((unsigned)((ta + 255) / 256), s.batch), dim3(256), 0, st, dA, dPA[v], s.M, s.K, Ts[v], (size_t)s.M * s.K, ap_bs);
      const int BN = Ts[v], tiles_n = (s.N + BN - 1) / BN;
      bp_bs[v] = (size_t)(s.K / 16) * tiles_n * 4 * BN * 8;
      CK(hipMalloc((void**)&dPB[v], bp_bs[v] * s.batch * 2));
      const size_t tb = (size_t)(s.K / 8) * tiles_n * BN;
      hipLaunchKernelGGL(precut_b_kernel, dim3((unsigned)((tb + 255) / 256), s.batch), dim3(256), 0, st, dB, dPB[v], s.K, s.N, BN, (size_t)s.K * s.N, bp_bs[v]);
    }
    CK(hipStreamSynchronize(st));
    int vidx = 0;
    auto run = [&](const char* what, int BM, int BN, int nss, bool check, auto kern, int nthreads) {
      const int my = vidx++;
      if (!only_set.empty() && std::find(only_set.begin(), only_set.end(), my) == only_set.end()) return;
      if (s.M % BM || s.N % BN) { printf("   [%2d] %-44s (shape does not divide)\n", my, what); return; }
      const int smem = nss * (4 * BM * 16 + 4 * BN * 16);
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
      LabW q{};
      const int va = BM == 128 ? 0 : 1, vb = BN == 128 ? 0 : 1;
      q.Ap = dPA[va]; q.Bp = dPB[vb]; q.C = dC; q.M = s.M; q.N = s.N; q.K = s.K; q.ldc = s.N;
      q.ap_bs = ap_bs; q.bp_bs = bp_bs[vb]; q.c_bs = (size_t)s.M * s.N;
      q.ap_bytes = (unsigned)(ap_bs * 2); q.bp_bytes = (unsigned)(bp_bs[vb] * 2);
      q.tiles_n = s.N / BN; q.tiles_m = s.M / BM; q.ntiles = q.tiles_m * q.tiles_n; q.zswz = zswz; q.clk = dClk;
      // LAB_ROTATE=n (round 6): n copies of both operands and of the output, one per launch in turn -- 300 MB per copy for the 36-plane
      // shape, so with n >= 2 no launch finds its operands in the 256 MB Infinity Cache the way back-to-back launches on ONE set do
      static const int rot = getenv("LAB_ROTATE") ? std::max(1, atoi(getenv("LAB_ROTATE"))) : 1;
      std::vector<LabW> qs(rot, q); std::vector<void*> extra;
      for (int r = 1; r < rot; ++r) {
        unsigned short *a2, *b2; float* c2;
        CK(hipMalloc((void**)&a2, ap_bs * s.batch * 2)); CK(hipMalloc((void**)&b2, bp_bs[vb] * s.batch * 2)); CK(hipMalloc((void**)&c2, nc * 4));
        CK(hipMemcpyAsync(a2, dPA[va], ap_bs * s.batch * 2, hipMemcpyDeviceToDevice, st)); CK(hipMemcpyAsync(b2, dPB[vb], bp_bs[vb] * s.batch * 2, hipMemcpyDeviceToDevice, st));
        qs[r].Ap = a2; qs[r].Bp = b2; qs[r].C = c2; extra.push_back(a2); extra.push_back(b2); extra.push_back(c2);
      }
      CK(hipStreamSynchronize(st));
      int turn = 0;
      auto fn = [&] { hipLaunchKernelGGL(kern, dim3(q.ntiles * s.batch), dim3(nthreads), smem, st, qs[turn]); turn = (turn + 1) % rot; };
      struct FreeExtra { std::vector<void*>& v; ~FreeExtra() { for (void* p : v) (void)hipFree(p); } } free_extra{extra};
      CK(hipMemsetAsync(dC, 0, nc * 4, st));
      fn(); CK(hipStreamSynchronize(st));
      double err = -1;
      if (check) {
        std::vector<float> c(nc); CK(hipMemcpy(c.data(), dC, nc * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0;
        for (size_t i = 0; i < nc; ++i) { const double d = (double)c[i] - r[i]; num += d * d; den += (double)r[i] * r[i]; }
        err = std::sqrt(num / den);
      }
      CK(hipMemsetAsync(dClk, 0, 32 * sizeof(unsigned long long), st));
      float best = 1e30f;
      for (int rnd = 0; rnd < 3; ++rnd) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); best = std::min(best, t / reps);
      }
      const double fill_bytes = (double)q.ntiles * s.batch * (s.K / 16) * (4.0 * BM * 16 + 4.0 * BN * 16);
      unsigned long long hclk[32]; CK(hipMemcpy(hclk, dClk, sizeof(hclk), hipMemcpyDeviceToHost));
      double cs = 0, rs = 0; for (int i = 0; i < 16; ++i) { cs += (double)hclk[2 * i]; rs += (double)hclk[2 * i + 1]; }
      const double ghz = rs > 0 ? cs / rs * wall_khz * 1e-6 : 0.0;
      const double mfma_s = flops * 3 / 2 / (32.0 * 32 * 16) * 32 / 1024;     // matrix-pipe cycles per SIMD, 1024 SIMDs
      printf("   [%2d] %-46s rel-L2 %9.3e  %7.1f us  %6.1f fp32-eq TF  fills %5.2f TB/s  clk %.2f GHz  pipe %4.1f %%\n", my, what, err,
             best * 1e3, flops / best * 1e-9, fill_bytes / best * 1e-9, ghz, ghz > 0 ? 100.0 * mfma_s / (best * 1e-3 * ghz * 1e9) : 0.0);
      fflush(stdout);
    };
    //                                                         WGM WGN MR NB NSS WPS MODE PIPE
    run("128x128 4wv 2st 4/CU (ships)         the loop", 128, 128, 2, true,  gemm_w<4, 1, 1, 4, 2, 4, 0>, 256);
    run("128x128 4wv 2st 4/CU                 no MFMA", 128, 128, 2, false, gemm_w<4, 1, 1, 4, 2, 4, 1>, 256);
    run("128x128 4wv 2st 4/CU                 no fills", 128, 128, 2, false, gemm_w<4, 1, 1, 4, 2, 4, 2>, 256);
    run("128x128 4wv 2st 4/CU                 fills only", 128, 128, 2, false, gemm_w<4, 1, 1, 4, 2, 4, 3>, 256);
    run("128x128 4wv 2st 5/CU (<= 96 VGPRs)    the loop", 128, 128, 2, true,  gemm_w<4, 1, 1, 4, 2, 5, 0>, 256);
    run("128x128 4wv 2st 4/CU                 MFMAs only", 128, 128, 2, false, gemm_w<4, 1, 1, 4, 2, 4, 4>, 256);
    run("256x128 4wv x64r 2st 2/CU            MFMAs only", 256, 128, 2, false, gemm_w<4, 1, 2, 4, 2, 2, 4>, 256);
    run("128x128 4wv 3st 3/CU pipelined       the loop", 128, 128, 3, true,  gemm_w<4, 1, 1, 4, 3, 3, 0, 1>, 256);
    run("256x128 4wv x64r 2st 2/CU            the loop", 256, 128, 2, true,  gemm_w<4, 1, 2, 4, 2, 2, 0>, 256);
    run("256x128 4wv x64r 2st 2/CU            no fills", 256, 128, 2, false, gemm_w<4, 1, 2, 4, 2, 2, 2>, 256);
    run("256x128 4wv x64r 2st 2/CU            fills only", 256, 128, 2, false, gemm_w<4, 1, 2, 4, 2, 2, 3>, 256);
    run("256x128 4wv x64r 3st 2/CU pipelined  the loop", 256, 128, 3, true,  gemm_w<4, 1, 2, 4, 3, 2, 0, 1>, 256);
    run("256x128 4wv x64r 3st 2/CU pipelined  no fills", 256, 128, 3, false, gemm_w<4, 1, 2, 4, 3, 2, 2, 1>, 256);
    run("256x256 8wv (4x2) x64x128 3st 1/CU   the loop", 256, 256, 3, true,  gemm_w<4, 2, 2, 4, 3, 2, 0>, 512);
    run("256x256 8wv (4x2) 4st                no MFMA", 256, 256, 4, false, gemm_w<4, 2, 2, 4, 4, 2, 1>, 512);
    run("256x256 8wv (4x2) 4st                no fills", 256, 256, 4, false, gemm_w<4, 2, 2, 4, 4, 2, 2>, 512);
    run("256x256 8wv (4x2) 4st                fills only", 256, 256, 4, false, gemm_w<4, 2, 2, 4, 4, 2, 3>, 512);
    run("256x256 8wv (4x2) 4st pipelined      the loop", 256, 256, 4, true,  gemm_w<4, 2, 2, 4, 4, 2, 0, 1>, 512);
    run("256x256 8wv (4x2) 5st pipelined      the loop", 256, 256, 5, true,  gemm_w<4, 2, 2, 4, 5, 2, 0, 1>, 512);
    run("256x256 8wv (4x2) 4st pipelined      no fills", 256, 256, 4, false, gemm_w<4, 2, 2, 4, 4, 2, 2, 1>, 512);
    run("256x256 16wv (4x4) x64x64 4st 1/CU   the loop", 256, 256, 4, true,  gemm_w<4, 4, 2, 2, 4, 4, 0>, 1024);
    run("256x256 16wv (4x4) 4st pipelined     the loop", 256, 256, 4, true,  gemm_w<4, 4, 2, 2, 4, 4, 0, 1>, 1024);
    run("256x128 8wv (4x2) x64x64 3st 2/CU    the loop", 256, 128, 3, true,  gemm_w<4, 2, 2, 2, 3, 4, 0>, 512);
    run("256x128 8wv (4x2) x64x64 3st 2/CU pipelined", 256, 128, 3, true,  gemm_w<4, 2, 2, 2, 3, 4, 0, 1>, 512);
    {   // fused remainder (round 6): `slots` workgroups, whole tile + one K-slice of a remainder tile each
      const int total = (s.M / 128) * (s.N / 128) * s.batch, slots = 1024;
      if (s.M % 128 == 0 && s.N % 128 == 0 && total > slots && total < 2 * slots) {
        const int rem = total - slots, nss = s.K / 16;
        float* dSlab; int* dCnt;
        float* dSlabU;
        CK(hipMalloc((void**)&dSlab, (size_t)rem * 8 * 128 * 128 * 4)); CK(hipMalloc((void**)&dCnt, rem * sizeof(int)));
        CK(hipExtMallocWithFlags((void**)&dSlabU, (size_t)rem * 8 * 128 * 128 * 4, hipDeviceMallocUncached));
        CK(hipMemset(dCnt, 0, rem * sizeof(int)));
        auto runf = [&](const char* what, int tail_s, int xloc, auto kern, bool check, int fence = 0) {
          const int my = vidx++;
          if (!only_set.empty() && std::find(only_set.begin(), only_set.end(), my) == only_set.end()) return;
          if (rem * tail_s > slots) { printf("   [%2d] %-46s (more slices than slots)\n", my, what); return; }
          const int smem = 2 * (4 * 128 * 16 + 4 * 128 * 16);
          CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
          LabW q{};
          q.Ap = dPA[0]; q.Bp = dPB[0]; q.C = dC; q.M = s.M; q.N = s.N; q.K = s.K; q.ldc = s.N;
          q.ap_bs = ap_bs; q.bp_bs = bp_bs[0]; q.c_bs = (size_t)s.M * s.N;
          q.ap_bytes = (unsigned)(ap_bs * 2); q.bp_bytes = (unsigned)(bp_bs[0] * 2);
          q.tiles_n = s.N / 128; q.tiles_m = s.M / 128; q.ntiles = q.tiles_m * q.tiles_n; q.zswz = 1; q.clk = dClk;
          FuseP f{slots, rem, tail_s, (nss + tail_s - 1) / tail_s, xloc, fence ? dSlab : dSlabU, dCnt, fence};
          auto fn = [&] { hipLaunchKernelGGL(kern, dim3(slots), dim3(256), smem, st, q, f); };
          CK(hipMemsetAsync(dC, 0, nc * 4, st));
          fn(); CK(hipStreamSynchronize(st));
          double err = -1;
          if (check) {
            std::vector<float> c(nc); CK(hipMemcpy(c.data(), dC, nc * 4, hipMemcpyDeviceToHost));
            double num = 0, den = 0;
            for (size_t i = 0; i < nc; ++i) { const double d = (double)c[i] - r[i]; num += d * d; den += (double)r[i] * r[i]; }
            err = std::sqrt(num / den);
            // bit-reproducible: a second launch gives the same bits
            fn(); CK(hipStreamSynchronize(st));
            std::vector<float> c2(nc); CK(hipMemcpy(c2.data(), dC, nc * 4, hipMemcpyDeviceToHost));
            if (memcmp(c.data(), c2.data(), nc * 4) != 0) printf("        !! two launches differ\n");
          }
          CK(hipMemsetAsync(dClk, 0, 32 * sizeof(unsigned long long), st));
          float best = 1e30f;
          for (int rnd = 0; rnd < 3; ++rnd) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) fn();
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); best = std::min(best, t / reps);
          }
          unsigned long long hclk[32]; CK(hipMemcpy(hclk, dClk, sizeof(hclk), hipMemcpyDeviceToHost));
          double cs = 0, rs = 0; for (int i = 0; i < 16; ++i) { cs += (double)hclk[2 * i]; rs += (double)hclk[2 * i + 1]; }
          const double ghz = rs > 0 ? cs / rs * wall_khz * 1e-6 : 0.0;
          printf("   [%2d] %-46s rel-L2 %9.3e  %7.1f us  %6.1f fp32-eq TF  slabs %5.1f MB  clk %.2f GHz\n", my, what, err, best * 1e3,
                 flops / best * 1e-9, tail_s > 1 ? (double)rem * tail_s * 65536 / 1e6 : 0.0, ghz);
          fflush(stdout);
        };
        runf("fused remainder: K/8, agent fences, cached slab", 8, 0, gemm_fused<0>, true, 1);
        runf("fused remainder: 1024 WGs, tile + K/8 slice", 8, 0, gemm_fused<0>, true);
        runf("fused remainder: K/8 slices, one XCD per tile", 8, 1, gemm_fused<0>, true);
        runf("fused remainder: K/4 slices (512 WGs take one)", 4, 0, gemm_fused<0>, true);
        runf("fused remainder: K/2 slices (256 WGs take one)", 2, 0, gemm_fused<0>, true);
        runf("fused remainder: K/8 slices      fills only", 8, 0, gemm_fused<3>, false);
        runf("fused remainder: K/8 slices      no MFMA", 8, 0, gemm_fused<1>, false);
        CK(hipFree(dSlab)); CK(hipFree(dSlabU)); CK(hipFree(dCnt));
      }
    }
    for (int v = 0; v < 2; ++v) { CK(hipFree(dPA[v])); CK(hipFree(dPB[v])); }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dR));
  }
  return 0;
}
