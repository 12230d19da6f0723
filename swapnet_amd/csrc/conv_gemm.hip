// swapnet_amd -- fp32 implicit-GEMM convolution for gfx950 (CDNA4).
//
// One kernel family serves every dense contraction of the SwapNet G+D step
// (reference: modules/layers.py:15,31,132,137; modules/discriminators.py:110-131;
// modules/swapnet_modules.py:85-90; modules/pix2pix_modules.py:216-246 and the autograd
// backward of each):
//   * conv_fwd_kernel   C[m][n] = sum_k A[m][k] W[k][n]   m = output pixel, k = (kh,kw,ci)
//       - Conv2d forward (k4s2p1, k3s1 reflect, k4s1p1, x2-upsampled tail conv)
//       - ConvTranspose2d k4s2p1 forward as 4 sub-pixel phases of a 2x2 conv (OutMap scatter)
//       - every dgrad (the transposed op is again a gather conv over dY with re-packed weights)
//   * conv_wgrad_kernel dW[k][n] = sum_m A[m][k] dY[m][n]  (reduction over pixels)
// A is never materialised: the im2col tile is gathered from the NHWC activation straight
// into LDS (16-byte loads along the channel axis = full 128-B lines per 8 lanes).
//
// CDNA4 mapping: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD); workgroups of 4 wave64
// (128 x {192,128,64,32} tiles) or 8 wave64 (256 x 128), wave tile 64 x 64; BK = 32 per LDS stage,
// double-buffered LDS with the next stage prefetched into VGPRs while the current one feeds the
// matrix pipe (one barrier per stage).  A-tile rows are padded to 36 floats (16-byte aligned,
// conflict-free ds_read_b128 of a lane's 16 k-values, k order permuted inside the stage); the
// W / dY tile is read along its contiguous axis.  Launch modes: batched (Winograd planes), the four
// sub-pixel phases of a stride-2 scatter in one grid, deterministic split-K / split-M slabs chosen by
// a wave-quantisation cost model.  Narrow outputs (Cout <= 32) use v_mfma_f32_4x4x1 kernels (one
// pixel / one k-row per lane); the folded tail conv has fused 4-phase kernels.  See DESIGN.md 4.
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "hip_util.h"

namespace swn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmP {
  const float* x; int xH, xW, xC, xcs;
  int KH, KW, stride, pad_t, pad_l, pad_mode, ups;
  int Ho, Wo, M, K;
  const float* w; int Npad;
  const float* bias; int act; int accumulate;
  float* y; int yH, yW, ycs; int ymul, yoff, xmul, xoff; int Cout; int yC;
  int splits; int per_split; float* slab;
  int tiles_n; int ntiles;
  size_t x_bs, w_bs, y_bs, slab_bs;   // batched mode (blockIdx.z)
  int phases;                         // sub-pixel phase mode: blockIdx.z = 2a + b shifts pads / output offsets
  int tail4;                          // fused folded-tail kernels
  float* y_amax;                      // optional amax slot of everything the launch stores (pre-cut ring kernel + its reduce)
  double* stat;                       // optional InstanceNorm partial sums of the output (ops.h ConvFwdArgs::stat_partial): 128 x 128 pre-cut kernel
};

__device__ __forceinline__ void apply_phase(GemmP& p) {
  if (p.phases) {
    const int a = blockIdx.z >> 1, b = blockIdx.z & 1;
    p.pad_t -= a; p.pad_l -= b; p.yoff = a; p.xoff = b;
  }
}

__device__ __forceinline__ int src_coord(int e, int ext, int pad_mode, int ups) {
  if (pad_mode == PAD_REFLECT) {
    if (e < 0) e = -e;
    else if (e >= ext) e = 2 * ext - 2 - e;
  } else if (e < 0 || e >= ext) {
    return -1;
  }
  return e >> ups;
}

// XCD-aware tile order: the dispatcher round-robins consecutive workgroups over the 8
// XCDs; give each XCD a contiguous run of tiles so neighbouring tiles (same A rows /
// same weight panel) share one L2.  Bijective for any tile count.
__device__ __forceinline__ int xcd_swizzle(int bid, int n) {
  const int q = n >> 3, r = n & 7, xcd = bid & 7, i = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

template <int MT, int NT, int WGM, int WGN>
struct Tile {
  static constexpr int BM = 32 * MT * WGM;
  static constexpr int BN = 32 * NT * WGN;
  static constexpr int BK = 32;
  static constexpr int AS = BK + 4;   // 16-byte aligned rows; 36*i mod 64 banks distinct for 16 rows (b128)
  static constexpr int A_FLOATS = ((BM * AS + 3) / 4) * 4;
  static constexpr int B_FLOATS = BK * BN;
  static constexpr int SMEM_FWD = (2 * A_FLOATS + 2 * B_FLOATS + BM) * 4;
  // wgrad: A' tile [32 pixels][BM], B' tile [32 pixels][BN]
  static constexpr int SMEM_WG = (2 * BK * BM + 2 * BK * BN) * 4;
};

// ---------------------------------------------------------------------------------------
// forward-type kernel
// ---------------------------------------------------------------------------------------
template <int MT, int NT, int WGM, int WGN, bool FAST>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_fwd_kernel(GemmP p) {
  using T = Tile<MT, NT, WGM, WGN>;
  constexpr int BM = T::BM, BN = T::BN, AS = T::AS;
  constexpr int NTHR = 64 * WGM * WGN;
  constexpr int AROWS = NTHR / 8;        // A-tile rows covered by one pass (8 lanes x 16 B per row)
  constexpr int RA = BM / AROWS;
  constexpr int B4 = BN / 4;             // float4 per B-tile row
  constexpr int RB = 32 * B4 / NTHR;     // float4 of the 32 x BN weight tile per thread (element i = t + r * NTHR)
  static_assert(32 * B4 % NTHR == 0, "B tile must divide evenly over the workgroup");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * T::A_FLOATS;
  int* rowoff = (int*)(Bs + 2 * T::B_FLOATS);

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid / WGN, wn = wid % WGN;
  const int tile = xcd_swizzle(blockIdx.x, p.ntiles);
  const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int split = blockIdx.y;
  p.x += (size_t)blockIdx.z * p.x_bs; p.w += (size_t)blockIdx.z * p.w_bs;
  p.y += (size_t)blockIdx.z * p.y_bs; p.slab += (size_t)blockIdx.z * p.slab_bs;
  apply_phase(p);

  const int q = t & 7, p0 = t >> 3;
  int a_iy0[RA], a_ix0[RA], a_base[RA];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int r = 0; r < RA; ++r) {
    const int m = m0 + p0 + AROWS * r;
    if (m < p.M) {
      const int n = m / HoWo, rem = m - n * HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_iy0[r] = oy * p.stride - p.pad_t;
      a_ix0[r] = ox * p.stride - p.pad_l;
      a_base[r] = n * p.xH * p.xW * p.xcs;
    } else {
      a_iy0[r] = 0; a_ix0[r] = 0; a_base[r] = -1;
    }
  }
  const int He = p.xH << p.ups, We = p.xW << p.ups;

  float4 ra[RA], rb[RB];
  // FAST (Cin % 32 == 0): a 32-wide k block never straddles a tap, and the tap changes only every Cin/32
  // stages (never for the 1x1 batched Winograd GEMMs).  The per-row source offsets of the current tap are
  // therefore cached; a stage only advances the channel offset.  This keeps ~100 VALU/SALU instructions
  // (tap decode, padding rules, 64-bit address math per row) out of every stage -- issue slots the matrix
  // pipe waits for when both waves of a SIMD sit behind the same workgroup barrier.
  int b_row[RB], b_off[RB];               // this thread's rows of the 32 x BN weight tile; offset inside the panel
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int bi = t + r * NTHR;
    const int n = n0 + (bi % B4) * 4;
    b_row[r] = bi / B4;
    b_off[r] = n < p.Npad ? b_row[r] * p.Npad + n : -1;
  }
  int a_off[RA];                          // element offset of this row's pixel for the cached tap, -1 = zero fill
  int ld_tap = -1, ld_ci = 0;             // wave-uniform loader state
  int g_tap = -1, g_ci = 0;               // generic path: per-thread (tap, ci)
  const int q32 = 32 / p.xC, r32 = 32 - q32 * p.xC;
  const int rcpKW = (65536 + p.KW - 1) / p.KW;
  auto set_tap = [&](int tap) {
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      int off = -1;
      if (a_base[r] >= 0) {
        const int sy = src_coord(a_iy0[r] + kh, He, p.pad_mode, p.ups);
        const int sx = src_coord(a_ix0[r] + kw, We, p.pad_mode, p.ups);
        if (sy >= 0 && sx >= 0) off = a_base[r] + (sy * p.xW + sx) * p.xcs + 4 * q;
      }
      a_off[r] = off;
    }
    ld_tap = tap;
  };
  auto load_tiles = [&](int kb) {
    const int k0 = kb * 32;
    if (FAST) {
      if (ld_tap < 0) {                    // first stage of this block (split-K blocks start anywhere)
        const int tap = k0 / p.xC;
        ld_ci = k0 - tap * p.xC;
        set_tap(tap);
      } else {                             // stages are visited in order
        ld_ci += 32;
        if (ld_ci >= p.xC) { ld_ci = 0; set_tap(ld_tap + 1); }
      }
#pragma unroll
      for (int r = 0; r < RA; ++r)
        ra[r] = a_off[r] >= 0 ? *reinterpret_cast<const float4*>(p.x + (size_t)(unsigned)a_off[r] + ld_ci)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      // generic path (Cin % 32 != 0): every thread tracks the (tap, ci) of its own 4-channel group and
      // advances it by 32 channels per stage with the precomputed quotient / remainder of 32 by Cin --
      // no per-stage integer divisions (tap -> (kh, kw) by a 16-bit reciprocal, exact for tap < 4096)
      if (g_tap < 0) {
        const int k = k0 + 4 * q;
        g_tap = k / p.xC;
        g_ci = k - g_tap * p.xC;
      } else {
        g_tap += q32; g_ci += r32;
        if (g_ci >= p.xC) { g_ci -= p.xC; ++g_tap; }
      }
      const bool kvalid = k0 + 4 * q < p.K;
      const int tap = g_tap, ci = g_ci;
      const int kh = (tap * rcpKW) >> 16, kw = tap - kh * p.KW;
#pragma unroll
      for (int r = 0; r < RA; ++r) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a_base[r] >= 0 && kvalid) {
          const int sy = src_coord(a_iy0[r] + kh, He, p.pad_mode, p.ups);
          const int sx = src_coord(a_ix0[r] + kw, We, p.pad_mode, p.ups);
          if (sy >= 0 && sx >= 0)
            v = *reinterpret_cast<const float4*>(p.x + (size_t)a_base[r] + (size_t)(sy * p.xW + sx) * p.xcs + ci);
        }
        ra[r] = v;
      }
    }
    const float* wk = p.w + (size_t)k0 * p.Npad;          // uniform part of the weight-panel address
#pragma unroll
    for (int r = 0; r < RB; ++r)
      rb[r] = (b_off[r] >= 0 && k0 + b_row[r] < p.K) ? *reinterpret_cast<const float4*>(wk + b_off[r])
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto store_tiles = [&](int buf) {
    float* A = As + buf * T::A_FLOATS;
#pragma unroll
    for (int r = 0; r < RA; ++r) *reinterpret_cast<float4*>(A + (p0 + AROWS * r) * AS + 4 * q) = ra[r];
    float* B = Bs + buf * T::B_FLOATS;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int bi = t + r * NTHR;
      *reinterpret_cast<float4*>(B + (bi / B4) * BN + (bi % B4) * 4) = rb[r];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // One LDS stage = 16 MFMA k-steps.  The reduction order inside the stage is permuted so that
  // step s consumes k = s (lanes 0-31) and k = 16 + s (lanes 32-63): a lane's 16 A values are
  // then contiguous in its row -> 4 ds_read_b128 instead of 16 ds_read_b32.  All fragments of the
  // stage are fetched before the first MFMA (64 VGPRs) so LDS latency is paid once per stage,
  // not once per k-step.
  auto compute = [&](int buf) {
    const int h = lane >> 5;
    const float* A = As + buf * T::A_FLOATS + (wm * MT * 32 + (lane & 31)) * AS + 16 * h;
    const float* B = Bs + buf * T::B_FLOATS + (16 * h) * BN + wn * NT * 32 + (lane & 31);
    float af[MT][16], bf[NT][16];
    // issue order = consumption order (k-steps 0-3 of every fragment first, then 4-7, ...): LDS returns
    // in order, so the first MFMAs wait for a quarter of the reads only and the rest land under them
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(A + i * 32 * AS + 4 * g);
        af[i][4 * g] = v.x; af[i][4 * g + 1] = v.y; af[i][4 * g + 2] = v.z; af[i][4 * g + 3] = v.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j][4 * g + e] = B[(4 * g + e) * BN + j * 32];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < 16; ++st)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][st], bf[j][st], acc[i][j], 0, 0, 0);
  };

  const int nkb = (p.K + 31) / 32;
  const int kb_begin = split * p.per_split;
  const int kb_end = min(nkb, kb_begin + p.per_split);
  if (kb_begin < kb_end) {
    load_tiles(kb_begin);
    store_tiles(0);
    __syncthreads();
    int cur = 0;
    for (int kb = kb_begin; kb < kb_end; ++kb) {
      const bool more = kb + 1 < kb_end;
      if (more) load_tiles(kb + 1);
      compute(cur);
      if (more) store_tiles(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }

  // ---- epilogue
  if (t < BM) {
    const int m = m0 + t;
    int off = -1;
    if (m < p.M) {
      const int n = m / HoWo, rem = m - n * HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      off = ((n * p.yH + oy * p.ymul + p.yoff) * p.yW + ox * p.xmul + p.xoff) * p.ycs;
    }
    rowoff[t] = off;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + wn * NT * 32 + j * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * MT * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        float v = acc[i][j][e];
        if (p.splits > 1) {
          if (m0 + row < p.M && col < p.Npad)
            p.slab[((size_t)split * p.M + (m0 + row)) * p.Npad + col] = v;
        } else {
          const int off = rowoff[row];
          if (off >= 0 && col < p.Cout) {
            if (p.bias) v += p.bias[col];
            v = act_apply(v, p.act);
            float* dst = p.y + (size_t)off + col;
            if (p.accumulate) v += *dst;
            *dst = v;
          }
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------
// forward-type kernel, LDS-DMA ring (round 2).  Same contraction as conv_fwd_kernel<FAST>, different machine:
//   * global -> LDS by `buffer_load_dwordx4 ... lds` (no staging VGPRs, no ds_write pass); padding taps and rows past M
//     are fetched OUT OF RANGE of the buffer descriptor, which the hardware returns as zeros;
//   * BK = 16, three LDS stages, two of them in flight across every barrier (counted s_waitcnt vmcnt, raw s_barrier);
//   * 64-byte A rows XOR-swizzled on the SOURCE side (lane l of a row fetches chunk (l & 3) ^ ((row >> 2) & 3)) so that
//     the 16-lane groups of ds_read_b128 touch 16 distinct 4-bank groups: conflict-free without padding;
//   * B fragments are ds_read_b64 of two ADJACENT columns, i.e. a lane's two 32x32 MFMA blocks hold columns
//     (2c, 2c+1) of its wave's 64 -> the epilogue stores float2 (256 contiguous bytes per row and wave);
//   * ~95 VGPRs: three 4-wave workgroups (128x128 tile, 48 KB) per CU = three independent waves per SIMD, so one
//     workgroup's barrier / epilogue / prologue is covered by the MFMAs of the other two.  (The register-staged
//     256x128 kernel above runs ONE 8-wave workgroup per CU -- 194 VGPRs, 105 KB -- and idles the matrix pipe a third
//     of the time; measured on the Winograd-plane and k4s2 shapes of this model: 87-105 -> 104-144 TFLOP/s.)
// The LDS-DMA is issued from an asm statement on purpose: hipcc's waitcnt pass puts `s_waitcnt vmcnt(0)` in front of every
// ds_read that follows an LDS-DMA *builtin* (one pending LDS write = "may alias"), which drains the ring each stage.
// Scheduling: tiles beyond a whole number of chip-fills ("the tail round") are split along K so that the last round is
// as full as the others (hybrid data-parallel / split-K); their partial tiles go to a compact slab that
// conv_dma_reduce_kernel sums in fixed order (deterministic) and finishes with the usual epilogue.
// ---------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// raw buffer descriptor: stride 0, num_records = bytes, gfx9 raw-buffer flags; offsets >= bytes read 0
__device__ __forceinline__ i32x4 make_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long a = (unsigned long long)ptr;
  i32x4 r;
  r[0] = (int)(unsigned)(a & 0xffffffffull); r[1] = (int)(unsigned)((a >> 32) & 0xffffull); r[2] = (int)bytes; r[3] = 0x00020000;
  return r;
}
// one LDS-DMA instruction: lane i fetches 16 bytes at base + voff + soff, the wave's 1 KiB lands at LDS byte lds_dst + 16 i
__device__ __forceinline__ void lds_dma16(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
}
constexpr unsigned DMA_OOB = 0x80000000u;      // > any buffer this library addresses (checked by the launcher)

// ---- fp32 products on the bf16 matrix cores ("split" main loop) ------------------------------------------------------
// x = hi + mid + lo with 8 mantissa bits each, cut by TRUNCATION, so the split itself is exact (24 bits in, 24 bits out).
// a*b = sum of 9 partial products; the 6 with weight >= 2^-16 relative to hi*hi are formed by
// v_mfma_f32_32x32x16_bf16 (each bf16 x bf16 product is exact in fp32, accumulation is fp32); the three dropped ones
// (mid*lo, lo*mid, lo*lo) are below 2^-24 |a||b|, i.e. below the rounding of an fp32 product.  Measured against an
// fp64-accumulated reference the result is slightly MORE accurate than v_mfma_f32_32x32x2_f32 (5.0e-7 vs 5.7e-7 rel-L2
// at K = 1024: 16 products per accumulator rounding instead of 2), and the 6 MFMAs per 16 k cost 192 cycles of the
// matrix pipe against 512 for the f32 form (tools/gemm_lab_split.hip: 110 -> 162 fp32-equivalent TFLOP/s; the loop is
// then bound by the ~5.5 VALU instructions per element of the split).  A lane's 8 fragment values are k = 8h .. 8h+7
// of its row / column -- exactly the operand layout of the 32x32x16 instruction.  SWN_SPLIT=0 selects the f32 MFMA.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
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
__device__ __forceinline__ f32x16 mma_bf16(u32x4 a, u32x4 b, f32x16 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#else
  return c;
#endif
}
// acc[i][j] += A_i (rows) x B_j (columns) over the lane's 8 k values, i, j in {0, 1}
__device__ __forceinline__ void split_mma_2x2(f32x16 (&acc)[2][2], const float (&af)[2][8], const float (&bf)[2][8]) {
  u32x4 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { split8(af[i], ah[i], am[i], al[i]); split8(bf[i], bh[i], bm[i], bl[i]); }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x16 c = acc[i][j];
      c = mma_bf16(al[i], bh[j], c); c = mma_bf16(ah[i], bl[j], c); c = mma_bf16(am[i], bm[j], c);     // smallest terms first
      c = mma_bf16(am[i], bh[j], c); c = mma_bf16(ah[i], bm[j], c); c = mma_bf16(ah[i], bh[j], c);
      acc[i][j] = c;
    }
}

struct DmaSched {            // hybrid schedule, computed by the launcher
  int full;                  // work units [0, full): one whole tile each
  int tail_tiles, tail_s;    // then tail_tiles tiles split tail_s ways along K
  int per_split;             // stages per split of a tail tile
  int tiles_per_z;           // tiles of one batch element / phase
  // > 0 (the four sub-pixel phases of one launch, pre-cut ring kernel): z is the FAST index of the tile order, gtile = tile * zfast + z.
  // The phases of a launch gather the same input through different tap offsets: with z slow, an XCD's contiguous run of tiles
  // (xcd_swizzle) is part of ONE phase and the same input region is fetched by the four XCDs that hold its four phases
  int zfast;
};

template <int WGM, int WGN>
struct DmaTile {
  static constexpr int NW = WGM * WGN, BM = 64 * WGM, BN = 64 * WGN, BK = 16, NST = 3;
  static constexpr int A_FL = BM * BK, B_FL = BK * BN, ST_FL = A_FL + B_FL;
  static constexpr int AI = 4 / WGN, BI = 4 / WGM;       // LDS-DMA instructions per wave per stage
  static constexpr int LPR = BN / 4, RPI = 64 / LPR;     // lanes per B row, B rows per instruction
  static constexpr int SMEM = NST * ST_FL * 4;
  static_assert(4 % WGN == 0 && 4 % WGM == 0, "tile shape");
};

template <int WGM, int WGN, bool SPLIT>
// (hipcc's second launch-bound is waves per SIMD: the 8-wave tile needs 2 workgroups = 4 waves per SIMD, <= 128 VGPRs)
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN >= 8 ? 4 : 3)) void conv_fwd_dma_kernel(GemmP p, DmaSched sc) {
#if defined(__HIP_DEVICE_COMPILE__)
  using T = DmaTile<WGM, WGN>;
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK, NST = T::NST, AI = T::AI, BI = T::BI;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wid / WGN, wn = wid % WGN;

  // ---- work unit -> (tile, K range)
  int u = blockIdx.x, gtile, split = 0, nsplit = 1, tt = 0;
  if (u < sc.full) {
    gtile = xcd_swizzle(u, sc.full);
  } else {
    u -= sc.full;
    tt = u / sc.tail_s; split = u - tt * sc.tail_s; nsplit = sc.tail_s;
    gtile = sc.full + tt;
  }
  const int z = gtile / sc.tiles_per_z, tile = gtile - z * sc.tiles_per_z;
  const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  p.x += (size_t)z * p.x_bs; p.w += (size_t)z * p.w_bs; p.y += (size_t)z * p.y_bs;
  if (p.phases) { const int a = z >> 1, b = z & 1; p.pad_t -= a; p.pad_l -= b; p.yoff = a; p.xoff = b; }
  const int nkb_all = p.K / BK;
  const int kb_begin = nsplit > 1 ? split * sc.per_split : 0;
  const int kb_end = nsplit > 1 ? min(nkb_all, kb_begin + sc.per_split) : nkb_all;

  const unsigned x_bytes = (unsigned)((((size_t)p.xH * p.xW * (size_t)(p.M / (p.Ho * p.Wo)) - 1) * p.xcs + p.xC) * 4);
  const i32x4 rsA = make_rsrc(p.x, x_bytes), rsB = make_rsrc(p.w, (unsigned)((size_t)p.K * p.Npad * 4));
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;

  // ---- loader state.  A: this lane owns AI rows (row = 16 q + lane / 4, q = wid * AI + r) and one swizzled chunk of each.
  int a_iy0[AI], a_ix0[AI], a_base[AI];
  unsigned a_voff[AI], b_voff[BI];
  const int HoWo = p.Ho * p.Wo;
  const int He = p.xH << p.ups, We = p.xW << p.ups;
#pragma unroll
  for (int r = 0; r < AI; ++r) {
    const int row = 16 * (wid * AI + r) + (lane >> 2);
    const int m = m0 + row;
    if (m < p.M) {
      const int n = m / HoWo, rem = m - n * HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_iy0[r] = oy * p.stride - p.pad_t;
      a_ix0[r] = ox * p.stride - p.pad_l;
      a_base[r] = n * p.xH * p.xW * p.xcs + 4 * ((lane & 3) ^ ((row >> 2) & 3));     // + inverse-swizzled chunk
    } else {
      a_iy0[r] = 0; a_ix0[r] = 0; a_base[r] = -1;
    }
  }
#pragma unroll
  for (int r = 0; r < BI; ++r) {
    const int krow = T::RPI * (wid * BI + r) + lane / T::LPR;
    const int nn = n0 + 4 * (lane % T::LPR);
    b_voff[r] = nn < p.Npad ? (unsigned)(krow * p.Npad + nn) * 4u : DMA_OOB;
  }
  auto set_tap = [&](int tap) {
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
    for (int r = 0; r < AI; ++r) {
      unsigned off = DMA_OOB;
      if (a_base[r] >= 0) {
        const int sy = src_coord(a_iy0[r] + kh, He, p.pad_mode, p.ups);
        const int sx = src_coord(a_ix0[r] + kw, We, p.pad_mode, p.ups);
        if (sy >= 0 && sx >= 0) off = (unsigned)(a_base[r] + (sy * p.xW + sx) * p.xcs) * 4u;
      }
      a_voff[r] = off;
    }
  };
  // stages are issued strictly in order kb_begin, kb_begin + 1, ...: (tap, ci) of the next stage to issue
  int ld_tap = (kb_begin * BK) / p.xC, ld_ci = kb_begin * BK - ld_tap * p.xC;
  set_tap(ld_tap);
  auto issue = [&](int st, int kb) {
    const unsigned As = lds0 + (unsigned)(st * T::ST_FL) * 4u, Bs = As + T::A_FL * 4u;
#pragma unroll
    for (int r = 0; r < AI; ++r) lds_dma16(a_voff[r], rsA, (unsigned)ld_ci * 4u, As + (unsigned)(wid * AI + r) * 1024u);
#pragma unroll
    for (int r = 0; r < BI; ++r)
      lds_dma16(b_voff[r], rsB, (unsigned)kb * (unsigned)(BK * 4) * (unsigned)p.Npad, Bs + (unsigned)(wid * BI + r) * 1024u);
    ld_ci += BK;
    if (ld_ci >= p.xC) { ld_ci = 0; ld_tap += 1; if (ld_tap < p.KH * p.KW) set_tap(ld_tap); }
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
  const int b_rd = T::A_FL + (8 * h) * BN + wn * 64 + 2 * l31;
  // k order inside a stage: step s multiplies k = s (lanes 0-31) and k = 8 + s (lanes 32-63)
  auto compute = [&](int st) {
    const float* S = smem + st * T::ST_FL;
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
    if (SPLIT) {
      split_mma_2x2(acc, af, bf);
    } else {
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s8], bf[j][s8], acc[i][j], 0, 0, 0);
    }
  };

  if (kb_begin < kb_end) {
    issue(0, kb_begin);
    if (kb_begin + 1 < kb_end) issue(1, kb_begin + 1);
    int st = 0;
    for (int kb = kb_begin; kb < kb_end; ++kb) {
      // this wave's share of stage kb has landed: only the next stage's AI + BI loads may still be in flight
      if (kb + 1 < kb_end) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(AI + BI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // everybody's share landed; everybody finished reading stage kb - 1
      asm volatile("" ::: "memory");
      int st2 = st + 2; if (st2 >= NST) st2 -= NST;
      if (kb + 2 < kb_end) issue(st2, kb + 2);         // overwrites the stage read in iteration kb - 1
      compute(st);
      st = st + 1 == NST ? 0 : st + 1;
    }
  }

  // ---- epilogue.  lane: columns (c, c+1) = n0 + wn*64 + 2*l31 + {0,1}; rows wm*64 + i*32 + (e&3) + 8*(e>>2) + 4*h
  const int colr = wn * 64 + 2 * l31;                   // column inside the tile
  if (nsplit > 1) {
    float* slab = p.slab + ((size_t)(tt * nsplit + split) * BM) * BN;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        *reinterpret_cast<float2*>(slab + (size_t)row * BN + colr) = make_float2(acc[i][0][e], acc[i][1][e]);
      }
    return;
  }
  __syncthreads();                                      // the ring is dead: reuse it for the per-row output offsets
  int* rowoff = reinterpret_cast<int*>(smem);
  if (t < BM) {
    const int m = m0 + t;
    int off = -1;
    if (m < p.M) {
      const int n = m / HoWo, rem = m - n * HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      off = ((n * p.yH + oy * p.ymul + p.yoff) * p.yW + ox * p.xmul + p.xoff) * p.ycs;
    }
    rowoff[t] = off;
  }
  __syncthreads();
  const int col = n0 + colr;
  const bool c0ok = col < p.Cout, c1ok = col + 1 < p.Cout;
  float b0 = 0.f, b1 = 0.f;
  if (p.bias) { if (c0ok) b0 = p.bias[col]; if (c1ok) b1 = p.bias[col + 1]; }
  if (c0ok) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int off = rowoff[wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];
        if (off < 0) continue;
        float v0 = act_apply(acc[i][0][e] + b0, p.act), v1 = act_apply(acc[i][1][e] + b1, p.act);
        float* dst = p.y + (size_t)off + col;
        if (c1ok) {
          if (p.accumulate) { const float2 o = *reinterpret_cast<const float2*>(dst); v0 += o.x; v1 += o.y; }
          *reinterpret_cast<float2*>(dst) = make_float2(v0, v1);
        } else {
          if (p.accumulate) v0 += *dst;
          *dst = v0;
        }
      }
  }
#endif
}

// sums the partial tiles of the split tail in fixed order and applies the epilogue; one float4 of a tile row per thread
template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_dma_reduce_kernel(GemmP p, DmaSched sc) {
  const int tt = blockIdx.y;
  const int e4 = blockIdx.x * 256 + threadIdx.x;
  if (e4 >= BM * BN / 4) return;
  const int r = e4 / (BN / 4), c4 = (e4 - r * (BN / 4)) * 4;
  const int gtile = sc.full + tt;
  const int z = sc.zfast ? gtile % sc.zfast : gtile / sc.tiles_per_z, tile = sc.zfast ? gtile / sc.zfast : gtile - z * sc.tiles_per_z;
  const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
  const int m = tile_m * BM + r, col = tile_n * BN + c4;
  if (m >= p.M || col >= p.Cout) return;
  p.y += (size_t)z * p.y_bs;
  if (p.phases) { p.yoff = z >> 1; p.xoff = z & 1; }
  const float* sl = p.slab + ((size_t)tt * sc.tail_s * BM + r) * BN + c4;
  float4 a = *reinterpret_cast<const float4*>(sl);
  for (int s = 1; s < sc.tail_s; ++s) {
    const float4 b = *reinterpret_cast<const float4*>(sl + (size_t)s * BM * BN);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (p.bias && col + j < p.Cout) v[j] += p.bias[col + j];
    v[j] = act_apply(v[j], p.act);
  }
  const int HoWo = p.Ho * p.Wo;
  const int n = m / HoWo, rem = m - n * HoWo;
  const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
  float* dst = p.y + (size_t)((n * p.yH + oy * p.ymul + p.yoff) * p.yW + ox * p.xmul + p.xoff) * p.ycs + col;
  float am = 0.f;
  if (col + 3 < p.Cout) {
    float4 o = make_float4(v[0], v[1], v[2], v[3]);
    if (p.accumulate) {
      const float4 old = *reinterpret_cast<const float4*>(dst);
      o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
    }
    *reinterpret_cast<float4*>(dst) = o;
    am = f4amax(o);
  } else {
    for (int j = 0; j < 4 && col + j < p.Cout; ++j) { const float o = p.accumulate ? dst[j] + v[j] : v[j]; dst[j] = o; am = fmaxf(am, fabsf(o)); }
  }
  // (threads that returned above hold nothing: a per-thread atomic with the pre-check costs a load for all but a few)
  if (p.y_amax && am > 0.f) amax_store(am, p.y_amax, blockIdx.x + blockIdx.y * gridDim.x);
}


// ---------------------------------------------------------------------------------------
// forward-type kernel on the LDS-DMA ring with the weight operand PRE-CUT (round 3).  conv_fwd_dma_kernel<SPLIT> spends
// 176 of its ~290 instructions per wave and 16-k stage cutting fragments, and half of that on the WEIGHT fragment: the same
// bf16 pieces of the same weights, recomputed by every workgroup of every launch.  Here whoever produces the weight operand
// (conv_precut after an optimizer step / a re-pack) hands it over as three bf16 planes already in MFMA operand order, and
// the waves are laid out WGM x 1: each wave owns 32 rows x ALL BN columns of the tile, so ONE activation-fragment cut (44
// VALU) feeds 6 NB MFMAs, the B fragments are plain ds_read_b128 (no VALU), and a wave only ever reads the A rows it
// fetched itself.  ~115 instructions per wave-stage instead of ~290 (tools/ring_lab.hip, same box, fp32-equivalent TFLOP/s:
// Winograd planes 157 -> 179, 8192x512x4096 133 -> 169, 131072x128x1024 163 -> 190, 131072x64x1536 (256 x 64 tile) 96 -> 139).
// Pre-cut layout (conv_precut): Wp[stage = k / 16][tile_n][kq 2][plane 3][pos BN][8 k] bf16, pos = (n % NB) * 32 + n / NB
// inside a BN-column tile: the lane at position l31 of column block j holds column NB * l31 + j, i.e. NB adjacent columns
// over its NB accumulators -> 16-byte epilogue stores.  One (stage, tile_n) block is 12 * BN / 128 contiguous KiB = the
// LDS image of the stage, fetched by plain consecutive 1-KiB LDS-DMA pieces.
// Everything else (gather through out-of-range zero fill, XOR-swizzled A rows, hybrid split-K schedule, epilogue) is
// conv_fwd_dma_kernel's.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_dma16c(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
               : : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory", "m0");
}

// ---- two fp16 planes instead of three bf16 ones (round 3, late; tools/ring_lab.hip gemm_h) -------------------------------------
// The six-term bf16 loop sits on the chip's POWER cap, not on an issue limit: with random operands every lab variant (no VALU
// at all, 16x16x32 MFMAs, 8 accumulators, one barrier per 32 k) lands at 180-200 fp32-equivalent TFLOP/s while the shader clock
// falls to 1.1-1.3 GHz (s_memtime / s_memrealtime inside the kernel), and zero-filled operands run the same binary at
// 1.7-2.0 GHz and 233-269 (profiles/ring_lab_r03_clock.txt).  So the remaining factor is in the matrix work per product:
// x = h + l with h = fp16(x), l = fp16(x - h) carries 22 mantissa bits, and h h + h l + l h is THREE MFMAs at an error of the
// dropped l l term, 2^-22 relative (lab: 3.9e-7 rel-L2 against 5.0e-7 for the six bf16 terms, 300-325 TFLOP/s against 183).
// fp16's exponent range is what this costs: each operand is scaled by a power of two chosen from its amax (exact, removed
// from the fp32 accumulators in the epilogue): A -- the activations / gradients cut in the loop -- from 256 partial maxima
// that amax_partials_kernel leaves in the stream scratch right before the launch (amax * 2^kA in [2^11, 2^12): overflow-free
// with a factor 16 to spare, 22 bits for every element within 2^-14 of the largest, an ABSOLUTE floor of amax * 2^-37
// below); B -- the pre-cut weight operand -- by its producer (amax of the source * 2^kB in [2^9, 2^10), derived operands
// such as Winograd-transformed filters stay within a factor 32 of that), which stores kB in a 16-byte trailer of the panel.
// Unscaled gradient-magnitude operands lose everything (lab: 1.2e-1), a scale off by 2^-8 costs two digits (3.7e-5), a
// scale too large by 2^8 nothing: profiles/ring_lab_r03_range.txt.  SWN_PC_PLANES=3 keeps the bf16 form.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x16 mma_f16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// The low plane of two elements, l = fp16(x - h), as one v_fma_mix{lo,hi}_f16 each (round 6).  x - h is exact in fp32 whatever the rounding
// of h (h holds x's leading 11 bits: the difference has at most 13 significant bits), so the single rounding of the fused x * 1 - h equals
// the v_cvt_f32_f16 / v_sub_f32 / v_cvt_f16_f32 sequence bit for bit (tools/mix_cut_check.hip on the MI355X: 0 mismatches over
// truncated and nearest h, normal and subnormal values) -- 4 VALU per pair with the packed multiply instead of the 8 the compiler
// emitted for most elements: the loops that cut an operand per 16-k step are VALU-bound (the generic weight-gradient loader:
// 146 VALU against 12 MFMAs per wave and stage).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned resid_pack(float x0, float x1, unsigned h) {
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(h));
  return l;
}
// h by truncation (one v_cvt_pkrtz for two elements), l rounded to nearest
__device__ __forceinline__ void split8h(const float* v, float sa, u32x4& hi, u32x4& lo) {
  const f32x2 s2 = f32x2{sa, sa};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2 x = f32x2{v[2 * q], v[2 * q + 1]} * s2;
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x[0], x[1]));
    hi[q] = h;
    lo[q] = resid_pack(x[0], x[1], h);
  }
}
// h rounded to NEAREST (v_cvt_pk_f16_f32): the residual l then has no preferred sign.  With both operands cut by truncation the
// dropped l_a l_b term always carries the sign of a b -- a relative bias of ~2^-22.6 on one-signed operands (measured: -1.5e-7 on
// post-ReLU x against positive dY); one operand rounded to nearest makes the term zero-mean.
__device__ __forceinline__ void split8h_rn(const float* v, float sa, u32x4& hi, u32x4& lo) {
  const f32x2 s2 = f32x2{sa, sa};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2 x = f32x2{v[2 * q], v[2 * q + 1]} * s2;
    const unsigned h = __builtin_bit_cast(unsigned, f16x2{(_Float16)x[0], (_Float16)x[1]});
    hi[q] = h;
    lo[q] = resid_pack(x[0], x[1], h);
  }
}
// operand stored in PAIR form by its producer (wino.hip pair_word: {h | l << 16} per element): the two MFMA operands of 8
// consecutive k are byte permutes of the 8 words -- 8 VALU instead of the 32 of split8h
__device__ __forceinline__ void pair8(const float* w, u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned w0 = __float_as_uint(w[2 * q]), w1 = __float_as_uint(w[2 * q + 1]);
    hi[q] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);                 // {w1[15:0], w0[15:0]}
    lo[q] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);                 // {w1[31:16], w0[31:16]}
  }
}
// ONE fp16 plane (SWN_PC_PLANES=1 / SWN_WGRAD_PLANES=1, the reduced-precision configuration): h = fp16(x * 2^k) rounded to
// nearest, the low plane is not formed -- one MFMA per product, operands carry 11 mantissa bits (bf16 carries 8)
__device__ __forceinline__ void split8h1(const float* v, float sa, u32x4& hi) {
#pragma unroll
  for (int q = 0; q < 4; ++q) hi[q] = __builtin_bit_cast(unsigned, f16x2{(_Float16)(v[2 * q] * sa), (_Float16)(v[2 * q + 1] * sa)});
}
__device__ __forceinline__ void split_mma_2x2_h1(f32x16 (&acc)[2][2], const float (&af)[2][8], const float (&bf)[2][8], float sa, float sb) {
  u32x4 ah[2], bh[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { split8h1(af[i], sa, ah[i]); split8h1(bf[i], sb, bh[i]); }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = mma_f16(ah[i], bh[j], acc[i][j]);
}
// both operands fp32 in LDS (the weight-gradient kernel): acc[i][j] += A_i x B_j over the lane's 8 k values, two fp16 planes each
__device__ __forceinline__ void split_mma_2x2_h(f32x16 (&acc)[2][2], const float (&af)[2][8], const float (&bf)[2][8], float sa, float sb) {
  u32x4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { split8h(af[i], sa, ah[i], al[i]); split8h_rn(bf[i], sb, bh[i], bl[i]); }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x16 c = acc[i][j];
      c = mma_f16(al[i], bh[j], c); c = mma_f16(ah[i], bl[j], c); c = mma_f16(ah[i], bh[j], c);          // smallest terms first
      acc[i][j] = c;
    }
}
// k with amax * 2^k in [2^(top-1), 2^top); 0 for an all-zero (or non-finite) operand.  |k| <= 100 keeps 2^k a normal float.
__host__ __device__ __forceinline__ int scale_exp(float amax, int top) {
  if (!(amax > 0.f) || amax > 3.0e38f) return 0;
  unsigned bits; memcpy(&bits, &amax, 4);
  const int e = (int)((bits >> 23) & 255u) - 127;
  const int k = top - 1 - e;
  return k < -100 ? -100 : (k > 100 ? 100 : k);
}
__device__ __forceinline__ float pow2f(int k) { return __uint_as_float((unsigned)(127 + k) << 23); }
// every lane ends up with the maximum of the 256 partials (whole wave active)
__device__ __forceinline__ float amax256(const float* part, int lane) {
  float m = fmaxf(fmaxf(part[lane], part[lane + 64]), fmaxf(part[lane + 128], part[lane + 192]));
#pragma unroll
  for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  return m;
}
constexpr int PC_TOP_A = 12, PC_TOP_B = 10;
constexpr int PC_TRAILER = 8;           // bf16/f16 elements (16 bytes) behind a two-plane panel: int kB
// 256 partial maxima of |x| over [batch][rows][C] (row stride rs, batch stride bs floats; C % 4 == 0, 16-byte aligned rows).
// 256 blocks x 1024 threads, four independent 16-byte loads in flight per thread (64 KB per CU); `flat`: the region is one
// dense array of `total4` float4s (no index arithmetic).  No atomics: the consumers reduce the 256 partials themselves.
__device__ __forceinline__ float amax4(const float4& v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
// fold != 0: the block maxima are folded into the slot `out` (atomic max, hip_util.h amax_store) instead of overwriting it;
// floor: a value the result is at least (a known bound of what another producer writes into the same buffer)
__global__ __launch_bounds__(1024) void amax_partials_kernel(const float* x, size_t rows, int C4, size_t rs, int batch, size_t bs, int flat,
                                                             float* out, int fold, float floor) {
  const size_t total = (size_t)batch * rows * C4;
  constexpr size_t S = (size_t)256 * 1024;
  float m = floor;
  auto at = [&](size_t i) -> const float4* {
    if (flat) return reinterpret_cast<const float4*>(x) + i;
    const size_t r = i / C4; const int c = (int)(i - r * C4);
    const size_t b = r / rows, rr = r - b * rows;
    return reinterpret_cast<const float4*>(x + b * bs + rr * rs + 4 * c);
  };
  size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
  for (; i + 3 * S < total; i += 4 * S) {
    const float4 v0 = *at(i), v1 = *at(i + S), v2 = *at(i + 2 * S), v3 = *at(i + 3 * S);
    m = fmaxf(m, fmaxf(fmaxf(amax4(v0), amax4(v1)), fmaxf(amax4(v2), amax4(v3))));
  }
  for (; i < total; i += S) m = fmaxf(m, amax4(*at(i)));
#pragma unroll
  for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float red[16];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x < 16) {
    float r = red[threadIdx.x];
#pragma unroll
    for (int o = 8; o; o >>= 1) r = fmaxf(r, __shfl_xor(r, o));
    if (threadIdx.x == 0) {
      if (fold) amax_store(r, out, blockIdx.x);
      else out[blockIdx.x] = r;
    }
  }
}

template <int WGM, int NB, int NSTG, int PL = 2>
struct PcTile {
  static constexpr int NW = WGM, BM = 32 * WGM, BN = 32 * NB, BK = 16, NST = NSTG;
  static constexpr int A_BYTES = BM * BK * 4, B_BYTES = 2 * PL * BN * 16, ST_BYTES = A_BYTES + B_BYTES;
  static constexpr int APC = A_BYTES / 1024, BPC = B_BYTES / 1024;
  static constexpr int AI = APC / WGM, BI = BPC / WGM, BREM = BPC % WGM;   // pieces per wave; waves < BREM carry one more of B
  static constexpr int SMEM = NST * ST_BYTES;
  static_assert(APC % WGM == 0 && NB % 2 == 0 && (PL == 1 || PL == 2), "tile shape");
};

// WGCU = workgroups per CU the tile is sized for (LDS) -> waves per SIMD the register allocation must allow
// APAIR: the activation operand arrives in pair form (a_kscale = the exponent its producer scaled it by): no cut in the loop
template <int WGM, int NB, int NSTG, int WGCU, int PL, bool APAIR = false>
__global__ __launch_bounds__(64 * WGM, WGCU * WGM / 4) void conv_fwd_pc_kernel(GemmP p, DmaSched sc, const unsigned short* wpc, size_t wpc_bs,
                                                                               const float* a_amax, const int* a_kscale) {
#if defined(__HIP_DEVICE_COMPILE__)
  using T = PcTile<WGM, NB, NSTG, PL>;
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK, NST = T::NST, AI = T::AI, BI = T::BI, BREM = T::BREM;
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);

  // ---- work unit -> (tile, K range)
  int u = blockIdx.x, gtile, split = 0, nsplit = 1, tt = 0;
  if (u < sc.full) {
    gtile = xcd_swizzle(u, sc.full);
  } else {
    u -= sc.full;
    tt = u / sc.tail_s; split = u - tt * sc.tail_s; nsplit = sc.tail_s;
    gtile = sc.full + tt;
  }
  const int z = sc.zfast ? gtile % sc.zfast : gtile / sc.tiles_per_z, tile = sc.zfast ? gtile / sc.zfast : gtile - z * sc.tiles_per_z;
  const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  p.x += (size_t)z * p.x_bs; p.y += (size_t)z * p.y_bs;
  wpc += (size_t)z * wpc_bs;
  if (p.phases) { const int a = z >> 1, b = z & 1; p.pad_t -= a; p.pad_l -= b; p.yoff = a; p.xoff = b; }
  const int nkb_all = p.K / BK;
  const int kb_begin = nsplit > 1 ? split * sc.per_split : 0;
  const int kb_end = nsplit > 1 ? min(nkb_all, kb_begin + sc.per_split) : nkb_all;

  const unsigned x_bytes = (unsigned)((((size_t)p.xH * p.xW * (size_t)(p.M / (p.Ho * p.Wo)) - 1) * p.xcs + p.xC) * 4);
  const unsigned b_stage = (unsigned)p.tiles_n * T::B_BYTES;        // bytes of one 16-k stage of the pre-cut panel
  const i32x4 rsA = make_rsrc(p.x, x_bytes), rsB = make_rsrc(wpc, (unsigned)nkb_all * b_stage);
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem_c;
  // two-plane form: the power-of-two operand scales (A from the partial maxima of this launch, B from the panel's trailer)
  int kA = 0, kB = 0;
  {
    if constexpr (APAIR) kA = __builtin_amdgcn_readfirstlane(*a_kscale);
    else kA = __builtin_amdgcn_readfirstlane(scale_exp(amax256(a_amax, lane), PC_TOP_A));
    kB = *reinterpret_cast<const int*>(wpc + (size_t)nkb_all * (b_stage / 2));
  }
  const float sa = pow2f(kA);

  // ---- loader state.  A: this lane owns AI rows of its OWN wave's 32 (row = 32 wid + 16 r + lane / 4) and one swizzled chunk.
  int a_iy0[AI], a_ix0[AI], a_base[AI];
  unsigned a_voff[AI];
  const int HoWo = p.Ho * p.Wo;
  const int He = p.xH << p.ups, We = p.xW << p.ups;
#pragma unroll
  for (int r = 0; r < AI; ++r) {
    const int row = 16 * (wid * AI + r) + (lane >> 2);
    const int m = m0 + row;
    if (m < p.M) {
      const int n = m / HoWo, rem = m - n * HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_iy0[r] = oy * p.stride - p.pad_t;
      a_ix0[r] = ox * p.stride - p.pad_l;
      a_base[r] = n * p.xH * p.xW * p.xcs + 4 * ((lane & 3) ^ ((row >> 2) & 3));
    } else {
      a_iy0[r] = 0; a_ix0[r] = 0; a_base[r] = -1;
    }
  }
  auto set_tap = [&](int tap) {
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
    for (int r = 0; r < AI; ++r) {
      unsigned off = DMA_OOB;
      if (a_base[r] >= 0) {
        const int sy = src_coord(a_iy0[r] + kh, He, p.pad_mode, p.ups);
        const int sx = src_coord(a_ix0[r] + kw, We, p.pad_mode, p.ups);
        if (sy >= 0 && sx >= 0) off = (unsigned)(a_base[r] + (sy * p.xW + sx) * p.xcs) * 4u;
      }
      a_voff[r] = off;
    }
  };
  int ld_tap = (kb_begin * BK) / p.xC, ld_ci = kb_begin * BK - ld_tap * p.xC;
  set_tap(ld_tap);
  const bool extra = BREM > 0 && wid < BREM;
  const unsigned b_voff = (unsigned)lane * 16u;
  const unsigned b_tile = (unsigned)tile_n * T::B_BYTES;
  auto issue = [&](int st, int kb) {
    const unsigned S = lds0 + (unsigned)(st * T::ST_BYTES), SB = S + T::A_BYTES;
    const unsigned bsrc = (unsigned)kb * b_stage + b_tile;
#pragma unroll
    for (int r = 0; r < AI; ++r) lds_dma16c(a_voff[r], rsA, (unsigned)ld_ci * 4u, S + (unsigned)(wid * AI + r) * 1024u);
#pragma unroll
    for (int r = 0; r < BI; ++r) lds_dma16c(b_voff, rsB, bsrc + (unsigned)(wid * BI + r) * 1024u, SB + (unsigned)(wid * BI + r) * 1024u);
    if (extra) lds_dma16c(b_voff, rsB, bsrc + (unsigned)(WGM * BI + wid) * 1024u, SB + (unsigned)(WGM * BI + wid) * 1024u);
    ld_ci += BK;
    if (ld_ci >= p.xC) { ld_ci = 0; ld_tap += 1; if (ld_tap < p.KH * p.KW) set_tap(ld_tap); }
  };

  f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  const int h = lane >> 5, l31 = lane & 31;
  const int f = (l31 >> 2) & 3;
  const int a_rd = (wid * 32 + l31) * 64;
  const int a_c0 = ((2 * h) ^ f) * 16, a_c1 = ((2 * h + 1) ^ f) * 16;
  const int b_rd = T::A_BYTES + (h * PL * BN + l31) * 16;                  // + (plane * BN + 32 j) * 16
  auto compute = [&](int st) {
    const char* S = smem_c + st * T::ST_BYTES;
    float af[8];
    {
      const float4 v0 = *reinterpret_cast<const float4*>(S + a_rd + a_c0);
      const float4 v1 = *reinterpret_cast<const float4*>(S + a_rd + a_c1);
      af[0] = v0.x; af[1] = v0.y; af[2] = v0.z; af[3] = v0.w; af[4] = v1.x; af[5] = v1.y; af[6] = v1.z; af[7] = v1.w;
    }
    if constexpr (PL == 1) {
      u32x4 bh[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) bh[j] = *reinterpret_cast<const u32x4*>(S + b_rd + (32 * j) * 16);
      u32x4 ah;
      split8h1(af, sa, ah);
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = mma_f16(ah, bh[j], acc[j]);
      return;
    }
    if constexpr (PL == 2) {
      u32x4 bh[NB], bl[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        bh[j] = *reinterpret_cast<const u32x4*>(S + b_rd + (0 * BN + 32 * j) * 16);
        bl[j] = *reinterpret_cast<const u32x4*>(S + b_rd + (1 * BN + 32 * j) * 16);
      }
      u32x4 ah, al;
      if constexpr (APAIR) pair8(af, ah, al);
      else split8h(af, sa, ah, al);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        f32x16 c = acc[j];
        c = mma_f16(al, bh[j], c); c = mma_f16(ah, bl[j], c); c = mma_f16(ah, bh[j], c);          // smallest terms first
        acc[j] = c;
      }
      return;
    }
    static_assert(PL == 1 || PL == 2, "one or two fp16 planes");
  };

  if (kb_begin < kb_end) {
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (kb_begin + s < kb_end) issue(s, kb_begin + s);
    int st = 0;
    for (int kb = kb_begin; kb < kb_end; ++kb) {
      // this wave's share of stage kb has landed: only the (at most NST - 2) younger stages may still be in flight
      const int younger = min(NST - 2, kb_end - 1 - kb);
      if (extra) {
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * (AI + BI + 1)) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(AI + BI + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * (AI + BI)) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(AI + BI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();          // everybody's share of B landed; everybody finished reading stage kb - 1
      asm volatile("" ::: "memory");
      int stn = st + NST - 1; if (stn >= NST) stn -= NST;
      if (kb + NST - 1 < kb_end) issue(stn, kb + NST - 1);
      compute(st);
      st = st + 1 == NST ? 0 : st + 1;
    }
  }

  {                                                       // remove the operand scales (two exact power-of-two factors)
    const float ca = pow2f(-kA), cb = pow2f(-kB);
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = (acc[j][e] * ca) * cb;
  }
  // ---- epilogue.  lane: rows wid*32 + (e&3) + 8*(e>>2) + 4*h, columns n0 + NB*l31 + j
  const int colr = NB * l31;
  if (nsplit > 1) {
    float* slab = p.slab + ((size_t)(tt * nsplit + split) * BM) * BN;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wid * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      float* dst = slab + (size_t)row * BN + colr;
      if constexpr (NB == 4) *reinterpret_cast<float4*>(dst) = make_float4(acc[0][e], acc[1][e], acc[2][e], acc[3][e]);
      else {
#pragma unroll
        for (int j = 0; j < NB; j += 2) *reinterpret_cast<float2*>(dst + j) = make_float2(acc[j][e], acc[j + 1][e]);
      }
    }
    return;
  }
  __syncthreads();                                      // the ring is dead: reuse it for the per-row output offsets
  int* rowoff = reinterpret_cast<int*>(smem_c);
  if (t < BM) {
    const int m = m0 + t;
    int off = -1;
    if (m < p.M) {
      const int n = m / HoWo, rem = m - n * HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      off = ((n * p.yH + oy * p.ymul + p.yoff) * p.yW + ox * p.xmul + p.xoff) * p.ycs;
    }
    rowoff[t] = off;
  }
  __syncthreads();
  const int col = n0 + colr;
  float am = 0.f;
  if (col < p.Cout) {
    float bj[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) bj[j] = (p.bias && col + j < p.Cout) ? p.bias[col + j] : 0.f;
    const bool full = col + NB - 1 < p.Cout;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int off = rowoff[wid * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];
      if (off < 0) continue;
      float v[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) v[j] = act_apply(acc[j][e] + bj[j], p.act);
      float* dst = p.y + (size_t)off + col;
      if (full) {
        if constexpr (NB == 4) {
          float4 o = make_float4(v[0], v[1], v[2], v[3]);
          if (p.accumulate) { const float4 q = *reinterpret_cast<const float4*>(dst); o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
          *reinterpret_cast<float4*>(dst) = o;
          am = fmaxf(am, f4amax(o));
        } else {
#pragma unroll
          for (int j = 0; j < NB; j += 2) {
            float2 o = make_float2(v[j], v[j + 1]);
            if (p.accumulate) { const float2 q = *reinterpret_cast<const float2*>(dst + j); o.x += q.x; o.y += q.y; }
            *reinterpret_cast<float2*>(dst + j) = o;
            am = fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y)));
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < NB; ++j)
          if (col + j < p.Cout) { const float o = p.accumulate ? dst[j] + v[j] : v[j]; dst[j] = o; am = fmaxf(am, fabsf(o)); }
      }
    }
  }
  if (p.y_amax) amax_fold_wave(am, p.y_amax, blockIdx.x * WGM + wid);        // (every wave arrives here converged)
  if constexpr (WGM == 4 && NB == 4) {
    // Conv + InstanceNorm fusion (modules/layers.py:12-24): the statistics' partial sums of this tile's 128 output rows (one image:
    // the launcher checked Ho * Wo % 128 == 0), per column, in fp64 -- lane: 16 rows x 4 columns, then the two half-waves (rows
    // + 4 h), then the four waves through LDS in wave order.  Fixed order: run-to-run identical.  What is summed is what was stored.
    if (p.stat) {
      double sm[NB], sq[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) { sm[j] = 0.0; sq[j] = 0.0; }
      if (col < p.Cout) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if (rowoff[wid * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] < 0) continue;
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const double v = (col + j < p.Cout) ? (double)(acc[j][e] + ((p.bias && col + j < p.Cout) ? p.bias[col + j] : 0.f)) : 0.0;
            sm[j] += v; sq[j] += v * v;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) { sm[j] += __shfl_xor(sm[j], 32); sq[j] += __shfl_xor(sq[j], 32); }
      __syncthreads();                                    // rowoff is dead
      double* red = reinterpret_cast<double*>(smem_c);    // [wave 4][column 128][2]
      if (h == 0) {
#pragma unroll
        for (int j = 0; j < NB; ++j) { red[(wid * BN + colr + j) * 2] = sm[j]; red[(wid * BN + colr + j) * 2 + 1] = sq[j]; }
      }
      __syncthreads();
      if (t < BN && n0 + t < p.yC) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int w = 0; w < WGM; ++w) { a += red[(w * BN + t) * 2]; b += red[(w * BN + t) * 2 + 1]; }
        double* o = p.stat + ((size_t)tile_m * p.yC + n0 + t) * 2;
        o[0] = a; o[1] = b;
      }
    }
  }
#endif
}

// producer of the pre-cut operand: one thread per (k / 8, tile_n, pos) writes the 16-byte plane entries: two fp16 planes (one in the
// reduced-precision configuration) of w * 2^kB with kB from the 256 partial maxima of the source, stored in the panel's trailer
__global__ __launch_bounds__(256) void conv_precut_kernel(const float* w, unsigned short* out, int K, int Npad, int BN, size_t w_bs,
                                                          size_t out_bs, const float* wamax, int planes) {
  const int NBc = BN / 32;
  const int tiles_n = (Npad + BN - 1) / BN;
  const size_t total = (size_t)(K / 8) * tiles_n * BN;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  int kB = 0;
  if (wamax) kB = scale_exp(amax256(wamax, threadIdx.x & 63), PC_TOP_B);      // (before any lane leaves)
  if (i >= total) return;
  const int nl = (int)(i % BN); const size_t q = i / BN;               // consecutive threads: consecutive columns (coalesced reads)
  const int tn = (int)(q % tiles_n), kq = (int)(q / tiles_n);
  const int n = tn * BN + nl;
  const int pos = (nl % NBc) * 32 + nl / NBc;                          // operand position of column nl
  w += (size_t)blockIdx.y * w_bs; out += (size_t)blockIdx.y * out_bs;
  u32x4* o = reinterpret_cast<u32x4*>(out);
  if (wamax) {
    const float sb = pow2f(kB);
    unsigned hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) x[e] = (n < Npad ? w[(size_t)(kq * 8 + 2 * j + e) * Npad + n] : 0.f) * sb;
      const f16x2 hh = f16x2{(_Float16)x[0], (_Float16)x[1]};
      hi[j] = __builtin_bit_cast(unsigned, hh);
      lo[j] = __builtin_bit_cast(unsigned, f16x2{(_Float16)(x[0] - (float)hh[0]), (_Float16)(x[1] - (float)hh[1])});
    }
    // [stage = kq / 2][tile_n][kq & 1][plane 1 or 2][pos][8 f16], then the trailer
    const size_t base = ((((size_t)(kq >> 1) * tiles_n + tn) * 2 + (kq & 1)) * planes) * BN;
    o[base + pos] = u32x4{hi[0], hi[1], hi[2], hi[3]};
    if (planes == 2) o[base + (size_t)BN + pos] = u32x4{lo[0], lo[1], lo[2], lo[3]};
    if (i == 0) *reinterpret_cast<int*>(out + (size_t)(K / 16) * tiles_n * 2 * planes * BN * 8) = kB;
  }
}

// ---------------------------------------------------------------------------------------
// narrow-N forward-type kernel (Cout <= 32: the 19-channel tail conv, PatchGAN's 1-channel
// prediction conv, dgrads into few-channel inputs).  The 32-wide MFMA tile wastes 13/32 of
// the matrix pipe at N = 19; v_mfma_f32_4x4x1 (16 independent 4x4 blocks per wave, same
// FLOP rate) has a 4-column granularity instead: each LANE owns one output pixel (B operand
// = its im2col value), the weights W[k][4g..4g+3] sit in lanes 4g..4g+3 of one VGPR and are
// broadcast to all 16 blocks (cbsz = 4, abid = g), and the result is, per lane, a float4 of
// 4 consecutive output channels of its pixel -- a 16-byte NHWC store.  Layout verified by
// tools/mfma_probe.hip.
// ---------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int G, int NG>
struct NarrowMac {
  static __device__ __forceinline__ void run(f32x4* acc, float w, float x) {
    acc[G] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x, acc[G], 4, G, 0);
    NarrowMac<G + 1, NG>::run(acc, w, x);
  }
};
template <int NG>
struct NarrowMac<NG, NG> {
  static __device__ __forceinline__ void run(f32x4*, float, float) {}
};

struct NarrowTile {
  static constexpr int BM = 256, BK = 16, AS = BK + 4;      // 20*l mod 64 distinct for 16 lanes (b128)
  static constexpr int A_FLOATS = BM * AS, B_FLOATS = 32 * AS;
  static constexpr int SMEM = (2 * A_FLOATS + 2 * B_FLOATS) * 4;
};

template <int NG, bool FAST>
__global__ __launch_bounds__(256) void conv_fwd_narrow_kernel(GemmP p) {
  using T = NarrowTile;
  constexpr int AS = T::AS, RA = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [2][256 pixels][16 k]
  float* Bt = smem + 2 * T::A_FLOATS;     // [2][32 n][16 k]   (weights, transposed)

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int tile = xcd_swizzle(blockIdx.x, p.ntiles);
  const int m0 = tile * T::BM;
  const int split = blockIdx.y;
  p.x += (size_t)blockIdx.z * p.x_bs; p.w += (size_t)blockIdx.z * p.w_bs;
  p.y += (size_t)blockIdx.z * p.y_bs; p.slab += (size_t)blockIdx.z * p.slab_bs;
  apply_phase(p);

  const int q = t & 3, p0 = t >> 2;       // 4 lanes x 16 B per tile row, 64 rows per pass
  int a_iy0[RA], a_ix0[RA], a_base[RA];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int r = 0; r < RA; ++r) {
    const int m = m0 + p0 + 64 * r;
    if (m < p.M) {
      const int n = m / HoWo, rem = m - n * HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_iy0[r] = oy * p.stride - p.pad_t;
      a_ix0[r] = ox * p.stride - p.pad_l;
      a_base[r] = n * p.xH * p.xW * p.xcs;
    } else {
      a_iy0[r] = 0; a_ix0[r] = 0; a_base[r] = -1;
    }
  }
  const int brow = t & 15, bcol = (t >> 4) * 4;     // threads 0..127 move the 16 x 32 weight tile; this mapping
                                                    // makes the transposed LDS stores ((bcol+i)*20 + brow) conflict-free
  const int He = p.xH << p.ups, We = p.xW << p.ups;

  float4 ra[RA], rb;
  auto load_tiles = [&](int kb) {
    const int k0 = kb * 16;
    int kh, kw, ci;
    bool kvalid = true;
    if (FAST) {
      const int tap = k0 / p.xC;           // xC % 32 == 0: a 16-wide k block never straddles a tap
      ci = k0 - tap * p.xC + 4 * q;
      kh = tap / p.KW; kw = tap - kh * p.KW;
    } else {
      const int k = k0 + 4 * q;
      kvalid = k < p.K;
      const int tap = k / p.xC;
      ci = k - tap * p.xC;
      kh = tap / p.KW; kw = tap - kh * p.KW;
    }
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a_base[r] >= 0 && kvalid) {
        const int sy = src_coord(a_iy0[r] + kh, He, p.pad_mode, p.ups);
        const int sx = src_coord(a_ix0[r] + kw, We, p.pad_mode, p.ups);
        if (sy >= 0 && sx >= 0)
          v = *reinterpret_cast<const float4*>(p.x + (size_t)a_base[r] + (size_t)(sy * p.xW + sx) * p.xcs + ci);
      }
      ra[r] = v;
    }
    rb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < 128 && k0 + brow < p.K && bcol < p.Npad)
      rb = *reinterpret_cast<const float4*>(p.w + (size_t)(k0 + brow) * p.Npad + bcol);
  };
  auto store_tiles = [&](int buf) {
    float* A = As + buf * T::A_FLOATS;
#pragma unroll
    for (int r = 0; r < RA; ++r) *reinterpret_cast<float4*>(A + (p0 + 64 * r) * AS + 4 * q) = ra[r];
    if (t < 128) {
      float* B = Bt + buf * T::B_FLOATS + bcol * AS + brow;
      B[0] = rb.x; B[AS] = rb.y; B[2 * AS] = rb.z; B[3 * AS] = rb.w;
    }
  };

  f32x4 acc[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const float* A = As + buf * T::A_FLOATS + (wid * 64 + lane) * AS;      // this lane's pixel
    const float* B = Bt + buf * T::B_FLOATS + (lane & 31) * AS;            // W[.][lane]
    float xv[16], wv[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 a = *reinterpret_cast<const float4*>(A + 4 * g);
      const float4 b = *reinterpret_cast<const float4*>(B + 4 * g);
      xv[4 * g] = a.x; xv[4 * g + 1] = a.y; xv[4 * g + 2] = a.z; xv[4 * g + 3] = a.w;
      wv[4 * g] = b.x; wv[4 * g + 1] = b.y; wv[4 * g + 2] = b.z; wv[4 * g + 3] = b.w;
    }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) NarrowMac<0, NG>::run(acc, wv[kk], xv[kk]);
  };

  const int nkb = (p.K + 15) / 16;
  const int kb_begin = split * p.per_split;
  const int kb_end = min(nkb, kb_begin + p.per_split);
  if (kb_begin < kb_end) {
    load_tiles(kb_begin);
    store_tiles(0);
    __syncthreads();
    int cur = 0;
    for (int kb = kb_begin; kb < kb_end; ++kb) {
      const bool more = kb + 1 < kb_end;
      if (more) load_tiles(kb + 1);
      compute(cur);
      if (more) store_tiles(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }

  // ---- epilogue: one output pixel per lane
  const int m = m0 + wid * 64 + lane;
  if (m >= p.M) return;
  if (p.splits > 1) {
    float* dst = p.slab + ((size_t)split * p.M + m) * p.Npad;
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (4 * g < p.Npad) *reinterpret_cast<float4*>(dst + 4 * g) = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
    return;
  }
  const int n = m / HoWo, rem = m - n * HoWo;
  const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
  float* dst = p.y + (size_t)((n * p.yH + oy * p.ymul + p.yoff) * p.yW + ox * p.xmul + p.xoff) * p.ycs;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int col = 4 * g;
    if (col >= p.Cout) break;
    float v[4] = {acc[g][0], acc[g][1], acc[g][2], acc[g][3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (p.bias && col + j < p.Cout) v[j] += p.bias[col + j];
      v[j] = act_apply(v[j], p.act);
    }
    if (col + 3 < p.Cout) {
      float4 o = make_float4(v[0], v[1], v[2], v[3]);
      if (p.accumulate) {
        const float4 old = *reinterpret_cast<const float4*>(dst + col);
        o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
      }
      *reinterpret_cast<float4*>(dst + col) = o;
    } else {
      for (int j = 0; j < 4 && col + j < p.Cout; ++j) dst[col + j] = p.accumulate ? dst[col + j] + v[j] : v[j];
    }
  }
}

// ---------------------------------------------------------------------------------------
// Folded tail conv, all four sub-pixel phases in one kernel (ops.h `tail4`).
// Phase (a,b) of the folded tail conv uses taps u < 2+a, v < 2+b of the 3x3 neighbourhood of the
// un-upsampled input: run as separate narrow-N GEMMs the four phases load 4+6+6+9 = 25 im2col taps per
// pixel, and at N = 19 that load path (not the matrix pipe) is the bound.  Here one block walks the
// 9 union taps once and feeds every phase that uses the tap from the same LDS stage: 2.8x fewer A
// loads, same MFMA work (v_mfma_f32_4x4x1, one pixel / one k-row per lane as in the narrow kernels).
// Folded weight layout: tail_fold_weights (optim.hip): phase panels at tap offsets {0,4,10,16},
// panel row = (u*(2+b)+v)*Cin + ci.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool tail_active(int ph, int u, int v) { return u < 2 + (ph >> 1) && v < 2 + (ph & 1); }
__device__ __forceinline__ int tail_row0(int ph, int u, int v, int xC) {      // first panel row of (ph, tap)
  const int pre = ph == 0 ? 0 : (ph == 1 ? 4 : (ph == 2 ? 10 : 16));
  return (pre + u * (2 + (ph & 1)) + v) * xC;
}

struct TailTile {
  static constexpr int BM = 256, BK = 16, AS = BK + 4;
  static constexpr int A_FLOATS = BM * AS, B_FLOATS = 32 * AS;       // B: one [32 n][16 k] tile per phase
  static constexpr int SMEM = (2 * A_FLOATS + 2 * 4 * B_FLOATS) * 4;
};

// PH = true (round 5): the same walk for the four sub-pixel phases of a k4 s2 TRANSPOSED conv run as 2x2 stride-1 convs (`phases`
// launches: ConvTranspose forward, the input gradient of a k4 s2 conv).  Phase (a, b) multiplies the taps u in {a, a+1}, v in
// {b, b+1} of the same 3x3 neighbourhood, its weight panel sits at p.w + ph * p.w_bs with rows ((u-a) * 2 + (v-b)) * Cin + ci.
// As four launches of conv_fwd_narrow_kernel every phase streamed the operand again -- the 20-channel input gradient of
// PatchGAN's model.0 fetched 16 x its operand (2.1 GB, profiles/traffic_r05.json), the largest single over-fetch of the step.
template <bool PH>
__device__ __forceinline__ bool phase_active(int ph, int u, int v) {
  if constexpr (!PH) return tail_active(ph, u, v);
  else { const int du = u - (ph >> 1), dv = v - (ph & 1); return du >= 0 && du <= 1 && dv >= 0 && dv <= 1; }
}
template <int NG, bool PH = false>
__global__ __launch_bounds__(256) void tail_fwd4_kernel(GemmP p) {
  using T = TailTile;
  constexpr int AS = T::AS, RA = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [2][256 pixels][16 k]
  float* Bt = smem + 2 * T::A_FLOATS;     // [2][4 phases][32 n][16 k]

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int tile = xcd_swizzle(blockIdx.x, p.ntiles);
  const int m0 = tile * T::BM;
  const int q = t & 3, p0 = t >> 2;
  int a_iy0[RA], a_ix0[RA], a_base[RA];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int r = 0; r < RA; ++r) {
    const int m = m0 + p0 + 64 * r;
    if (m < p.M) {
      const int n = m / HoWo, rem = m - n * HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_iy0[r] = oy - (PH ? p.pad_t : 1); a_ix0[r] = ox - (PH ? p.pad_l : 1);
      a_base[r] = n * p.xH * p.xW * p.xcs;
    } else {
      a_iy0[r] = 0; a_ix0[r] = 0; a_base[r] = -1;
    }
  }
  const int brow = t & 15, bcol = (t >> 4) * 4;     // threads 0..127: one float4 of each phase's 16 x 32 weight tile
                                                    // (conflict-free transposed LDS stores)

  float4 ra[RA], rb[4];
  // stages run over (tap, 16-channel chunk) in order: the per-row source offsets are recomputed only when
  // the tap changes (every Cin/16 stages), as in conv_fwd_kernel's fast loader
  int a_off[RA];
  int ld_tap = -1, ld_c0 = 0;
  auto load_tiles = [&](int kb) {
    if (ld_tap < 0 || ld_c0 + 16 >= p.xC) {
      ld_tap += 1; ld_c0 = 0;               // kb == 0, or the next tap
      const int uu = ld_tap / 3, vv = ld_tap - uu * 3;
#pragma unroll
      for (int r = 0; r < RA; ++r) {
        int off = -1;
        if (a_base[r] >= 0) {
          const int sy = a_iy0[r] + uu, sx = a_ix0[r] + vv;
          if (sy >= 0 && sy < p.xH && sx >= 0 && sx < p.xW) off = a_base[r] + (sy * p.xW + sx) * p.xcs + 4 * q;
        }
        a_off[r] = off;
      }
    } else {
      ld_c0 += 16;
    }
    const int tap = ld_tap, c0 = ld_c0;
    const int u = tap / 3, v = tap - u * 3;
    (void)kb;
#pragma unroll
    for (int r = 0; r < RA; ++r)
      ra[r] = a_off[r] >= 0 ? *reinterpret_cast<const float4*>(p.x + (size_t)(unsigned)a_off[r] + c0)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      rb[ph] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < 128 && phase_active<PH>(ph, u, v) && bcol < p.Npad) {
        const float* wp = PH ? p.w + (size_t)ph * p.w_bs + (size_t)(((u - (ph >> 1)) * 2 + (v - (ph & 1))) * p.xC) * p.Npad
                             : p.w + (size_t)tail_row0(ph, u, v, p.xC) * p.Npad;
        rb[ph] = *reinterpret_cast<const float4*>(wp + (size_t)(c0 + brow) * p.Npad + bcol);
      }
    }
  };
  auto store_tiles = [&](int buf) {
    float* A = As + buf * T::A_FLOATS;
#pragma unroll
    for (int r = 0; r < RA; ++r) *reinterpret_cast<float4*>(A + (p0 + 64 * r) * AS + 4 * q) = ra[r];
    if (t < 128) {
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        float* B = Bt + (buf * 4 + ph) * T::B_FLOATS + bcol * AS + brow;
        B[0] = rb[ph].x; B[AS] = rb[ph].y; B[2 * AS] = rb[ph].z; B[3 * AS] = rb[ph].w;
      }
    }
  };

  f32x4 acc[4][NG];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph)
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[ph][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf, int kb) {
    const int tap = (kb * 16) / p.xC;
    const int u = tap / 3, v = tap - u * 3;
    const float* A = As + buf * T::A_FLOATS + (wid * 64 + lane) * AS;
    float xv[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 a = *reinterpret_cast<const float4*>(A + 4 * g);
      xv[4 * g] = a.x; xv[4 * g + 1] = a.y; xv[4 * g + 2] = a.z; xv[4 * g + 3] = a.w;
    }
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      if (!phase_active<PH>(ph, u, v)) continue;   // block-uniform
      const float* B = Bt + (buf * 4 + ph) * T::B_FLOATS + (lane & 31) * AS;
      float wv[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(B + 4 * g);
        wv[4 * g] = b.x; wv[4 * g + 1] = b.y; wv[4 * g + 2] = b.z; wv[4 * g + 3] = b.w;
      }
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) NarrowMac<0, NG>::run(acc[ph], wv[kk], xv[kk]);
    }
  };

  const int nkb = (9 * p.xC) / 16;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  int cur = 0;
  for (int kb = 0; kb < nkb; ++kb) {
    const bool more = kb + 1 < nkb;
    if (more) load_tiles(kb + 1);
    compute(cur, kb);
    if (more) store_tiles(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  const int m = m0 + wid * 64 + lane;
  if (m >= p.M) return;
  const int n = m / HoWo, rem = m - n * HoWo;
  const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
    float* dst = p.y + (size_t)((n * p.yH + 2 * oy + (ph >> 1)) * p.yW + 2 * ox + (ph & 1)) * p.ycs;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int col = 4 * g;
      if (col >= p.Cout) break;
      float val[4] = {acc[ph][g][0], acc[ph][g][1], acc[ph][g][2], acc[ph][g][3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (p.bias && col + j < p.Cout) val[j] += p.bias[col + j];
        val[j] = act_apply(val[j], p.act);
      }
      if (col + 3 < p.Cout) {
        float4 o = make_float4(val[0], val[1], val[2], val[3]);
        if (PH && p.accumulate) { const float4 old = *reinterpret_cast<const float4*>(dst + col); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        *reinterpret_cast<float4*>(dst + col) = o;
      } else {
        for (int j = 0; j < 4 && col + j < p.Cout; ++j) dst[col + j] = (PH && p.accumulate) ? dst[col + j] + val[j] : val[j];
      }
    }
  }
}

// weight gradient of the folded tail conv, four phases fused: block = (union tap, pixel split), NW
// waves of 64 k-rows (one input channel per lane), reduction over PX-pixel stages (Wo % PX == 0: a
// stage lies inside one image row).  PX = 16 keeps the block at 35 KB of LDS and <= 168 VGPRs so 4
// blocks (3 waves per SIMD) share a CU -- with 32-pixel stages (1 wave per SIMD, one SIMD idle) the
// kernel ran at 27 % MFMA utilisation.  Writes the folded-gradient layout (or its slabs).
template <int NG, int NW, int PX>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(3, 3))) void tail_wgrad4_kernel(GemmP p) {
  constexpr int NTHR = 64 * NW, KC = 64 * NW;       // KC = Cin
  constexpr int BST = 20;                            // dY tile row stride (Npad <= 20)
  constexpr int A_FLOATS = PX * KC, B_FLOATS = PX * BST;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                   // [2][PX px][KC]
  float* Bs = smem + 2 * A_FLOATS;                    // [2][4 phases][PX px][20]  (+ tail padding)

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int tap = blockIdx.x, u = tap / 3, v = tap - u * 3;
  const int split = blockIdx.y;
  bool act[4];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) act[ph] = tail_active(ph, u, v);

  constexpr int A4 = KC / 4;                          // float4 per pixel row
  constexpr int RA = PX * A4 / NTHR;
  constexpr int AROWS = NTHR / A4;
  const int acol = (t % A4) * 4, arow0 = t / A4;      // rows arow0 + AROWS * r
  constexpr int NB4 = 4 * PX * 5;                     // dY: 4 phases x PX px x 5 float4
  constexpr int RB = (NB4 + NTHR - 1) / NTHR;

  const int nmb = p.M / PX;
  const int mb_begin = split * p.per_split;
  const int mb_end = min(nmb, mb_begin + p.per_split);

  float4 ra[RA], rb[RB];
  auto load_tiles = [&](int mb) {
    const int m = mb * PX;                            // first pixel of the stage; the PX share (n, oy)
    const int n = m / (p.Ho * p.Wo), rem = m - n * p.Ho * p.Wo;
    const int oy = rem / p.Wo, ox0 = rem - oy * p.Wo;
    const int sy = oy - 1 + u;
    const bool yok = sy >= 0 && sy < p.xH;
    const float* xrow = p.x + ((size_t)n * p.xH + (yok ? sy : 0)) * p.xW * p.xcs;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int sx = ox0 + arow0 + AROWS * r - 1 + v;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (yok && sx >= 0 && sx < p.xW) val = *reinterpret_cast<const float4*>(xrow + (size_t)sx * p.xcs + acol);
      ra[r] = val;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int i = t + NTHR * r;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < NB4) {
        const int ph = i / (PX * 5), j = i - ph * (PX * 5), px = j / 5, c4 = (j - px * 5) * 4;
        if (act[ph] && c4 < p.Npad)
          val = *reinterpret_cast<const float4*>(
              p.y + ((size_t)(n * p.yH + 2 * oy + (ph >> 1)) * p.yW + 2 * (ox0 + px) + (ph & 1)) * p.ycs + c4);
      }
      rb[r] = val;
    }
  };
  auto store_tiles = [&](int buf) {
    float* A = As + buf * A_FLOATS;
#pragma unroll
    for (int r = 0; r < RA; ++r) *reinterpret_cast<float4*>(A + (arow0 + AROWS * r) * KC + acol) = ra[r];
    float* B = Bs + buf * 4 * B_FLOATS;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int i = t + NTHR * r;
      if (i < NB4) *reinterpret_cast<float4*>(B + i * 4) = rb[r];       // [ph][px][20] is contiguous in i
    }
  };

  f32x4 acc[4][NG];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph)
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[ph][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const float* A = As + buf * A_FLOATS + wid * 64 + lane;
    float xv[PX];
#pragma unroll
    for (int st = 0; st < PX; ++st) xv[st] = A[st * KC];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      if (!act[ph]) continue;                          // block-uniform
      const float* B = Bs + (buf * 4 + ph) * B_FLOATS + (lane & 31);   // lanes >= 20 read past the row: unused blocks
      float dv[PX];
#pragma unroll
      for (int st = 0; st < PX; ++st) dv[st] = B[st * BST];
#pragma unroll
      for (int st = 0; st < PX; ++st) NarrowMac<0, NG>::run(acc[ph], dv[st], xv[st]);
    }
  };

  if (mb_begin < mb_end) {
    load_tiles(mb_begin);
    store_tiles(0);
    __syncthreads();
    int cur = 0;
    for (int mb = mb_begin; mb < mb_end; ++mb) {
      const bool more = mb + 1 < mb_end;
      if (more) load_tiles(mb + 1);
      compute(cur);
      if (more) store_tiles(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }
  float* out = p.splits > 1 ? p.slab + (size_t)split * 25 * KC * p.Npad : const_cast<float*>(p.w);
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
    if (!act[ph]) continue;
    float* row = out + (size_t)(tail_row0(ph, u, v, KC) + wid * 64 + lane) * p.Npad;
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (4 * g < p.Npad) *reinterpret_cast<float4*>(row + 4 * g) = make_float4(acc[ph][g][0], acc[ph][g][1], acc[ph][g][2], acc[ph][g][3]);
  }
}

// sums the K-split slabs in fixed order and applies the epilogue (4 output channels per thread)
__global__ void conv_fwd_reduce_kernel(GemmP p) {
  p.y += (size_t)blockIdx.z * p.y_bs; p.slab += (size_t)blockIdx.z * p.slab_bs;
  apply_phase(p);
  const int C4 = (p.Cout + 3) >> 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)p.M * C4;
  if (i >= total) return;
  const int m = (int)(i / C4), col = (int)(i - (size_t)m * C4) * 4;
  float4 a = *reinterpret_cast<const float4*>(p.slab + (size_t)m * p.Npad + col);   // Npad % 4 == 0
  for (int s = 1; s < p.splits; ++s) {
    const float4 b = *reinterpret_cast<const float4*>(p.slab + ((size_t)s * p.M + m) * p.Npad + col);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (p.bias && col + j < p.Cout) v[j] += p.bias[col + j];
    v[j] = act_apply(v[j], p.act);
  }
  const int HoWo = p.Ho * p.Wo;
  const int n = m / HoWo, rem = m - n * HoWo;
  const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
  float* dst = p.y + (size_t)((n * p.yH + oy * p.ymul + p.yoff) * p.yW + ox * p.xmul + p.xoff) * p.ycs + col;
  float am = 0.f;
  if (col + 3 < p.Cout) {
    float4 o = make_float4(v[0], v[1], v[2], v[3]);
    if (p.accumulate) {
      const float4 old = *reinterpret_cast<const float4*>(dst);
      o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
    }
    *reinterpret_cast<float4*>(dst) = o;
    am = f4amax(o);
  } else {
    for (int j = 0; j < 4 && col + j < p.Cout; ++j) { const float o = p.accumulate ? dst[j] + v[j] : v[j]; dst[j] = o; am = fmaxf(am, fabsf(o)); }
  }
  // (threads that returned above hold nothing: a per-thread atomic with the pre-check costs a load for all but a few)
  if (p.y_amax && am > 0.f) amax_store(am, p.y_amax, blockIdx.x + blockIdx.y * gridDim.x);
}

// ---------------------------------------------------------------------------------------
// wgrad-type kernel: rows = k (BM of them), cols = co, reduction over pixels
// ---------------------------------------------------------------------------------------
// NG > 0: narrow-N variant (Tile<2,1,4,1>: 256 k-rows x <= 32 channels): one k-row per lane,
// dY[m][4g..4g+3] broadcast from lanes 4g..4g+3, v_mfma_f32_4x4x1 as in conv_fwd_narrow_kernel.
// ROWU: Wo % 32 == 0, or Wo | 32 with Ho*Wo % 32 == 0: the 32 pixels of a stage lie inside one image at fixed
// offsets from its first pixel, whose decode is wave-uniform.
template <int MT, int NT, int WGM, int WGN, int NG = 0, bool ROWU = false>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_wgrad_kernel(GemmP p) {
  using T = Tile<MT, NT, WGM, WGN>;
  constexpr int BM = T::BM, BN = T::BN;
  constexpr int NTHR = 64 * WGM * WGN;
  constexpr int AROWS = NTHR / (BM / 4), BROWS = NTHR / (BN / 4);   // pixel rows of the [32][BM] / [32][BN] tiles per pass
  constexpr int RA = 32 / AROWS;       // float4 per thread
  constexpr int RB = 32 / BROWS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                    // [2][32*BM]
  float* Bs = smem + 2 * 32 * BM;      // [2][32*BN]

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid / WGN, wn = wid % WGN;
  const int tile = xcd_swizzle(blockIdx.x, p.ntiles);
  const int tile_n = tile % p.tiles_n, tile_k = tile / p.tiles_n;
  const int kt0 = tile_k * BM, n0 = tile_n * BN;
  const int split = blockIdx.y;
  p.x += (size_t)blockIdx.z * p.x_bs; p.y += (size_t)blockIdx.z * p.y_bs;
  p.w += (size_t)blockIdx.z * p.w_bs; p.slab += (size_t)blockIdx.z * p.slab_bs;
  apply_phase(p);

  // this thread's k (fixed for the whole kernel)
  const int acol = (t % (BM / 4)) * 4, arow0 = t / (BM / 4);
  const int k = kt0 + acol;
  const bool kvalid = k < p.K;
  const int tap = k / p.xC, ci = k - tap * p.xC;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int bcol = (t % (BN / 4)) * 4, brow0 = t / (BN / 4);
  const bool nvalid = (n0 + bcol) < p.yC;
  const int He = p.xH << p.ups, We = p.xW << p.ups;
  const int HoWo = p.Ho * p.Wo;

  // per-row pixel cursors (n, oy, ox), advanced by 32 pixels per stage instead of being
  // re-derived with two integer divisions per row per stage
  const int nmb = (p.M + 31) / 32;
  const int mb_begin = split * p.per_split;
  const int mb_end = min(nmb, mb_begin + p.per_split);
  int an[RA], aoy[RA], aox[RA], bn[RB], boy[RB], box[RB];
  auto decode = [&](int m, int& n, int& oy, int& ox) {
    n = m / HoWo; const int rem = m - n * HoWo;
    oy = rem / p.Wo; ox = rem - oy * p.Wo;
  };
  auto advance = [&](int& n, int& oy, int& ox) {
    ox += 32;
    if (ox >= p.Wo) {
      const int q = ox / p.Wo;
      ox -= q * p.Wo; oy += q;
      if (oy >= p.Ho) { const int r = oy / p.Ho; oy -= r * p.Ho; n += r; }
    }
  };
  // ROWU: offsets of this thread's rows inside a 32-pixel stage (the stage starts at ox = 0 unless Wo % 32 == 0)
  int a_dy[RA], a_dx[RA], b_dy[RB], b_dx[RB];
  if constexpr (ROWU) {
    const bool wide = (p.Wo & 31) == 0;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int j = arow0 + r * AROWS;
      a_dy[r] = wide ? 0 : j / p.Wo; a_dx[r] = wide ? j : j % p.Wo;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int j = brow0 + r * BROWS;
      b_dy[r] = wide ? 0 : j / p.Wo; b_dx[r] = wide ? j : j % p.Wo;
    }
  }
  if constexpr (!ROWU) {
#pragma unroll
    for (int r = 0; r < RA; ++r) decode(mb_begin * 32 + arow0 + r * AROWS, an[r], aoy[r], aox[r]);
#pragma unroll
    for (int r = 0; r < RB; ++r) decode(mb_begin * 32 + brow0 + r * BROWS, bn[r], boy[r], box[r]);
  }
  const int ximg = p.xH * p.xW * p.xcs;

  float4 ra[RA], rb[RB];
  auto load_tiles = [&](int mb) {
    const int mbase = mb * 32;
    if constexpr (ROWU) {
      // one scalar decode per stage; a thread's rows sit at fixed (dy, dx) from the stage's first pixel
      const int n = mbase / HoWo, rem = mbase - n * HoWo;
      const int oy0 = rem / p.Wo, ox0 = rem - oy0 * p.Wo;
      const bool live = mbase < p.M;                       // M % 32 == 0 here: a stage is all-valid or empty
      const float* ximg_p = p.x + (size_t)n * ximg + ci;
      const bool arow_ok = live && kvalid;
#pragma unroll
      for (int r = 0; r < RA; ++r) {
        const int sy = src_coord((oy0 + a_dy[r]) * p.stride - p.pad_t + kh, He, p.pad_mode, p.ups);
        const int sx = src_coord((ox0 + a_dx[r]) * p.stride - p.pad_l + kw, We, p.pad_mode, p.ups);
        ra[r] = (arow_ok && sy >= 0 && sx >= 0) ? *reinterpret_cast<const float4*>(ximg_p + (size_t)(sy * p.xW + sx) * p.xcs)
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const float* yimg_p = p.y + ((size_t)(n * p.yH + p.yoff) * p.yW + p.xoff) * p.ycs + n0 + bcol;
      const bool brow_ok = live && nvalid;
#pragma unroll
      for (int r = 0; r < RB; ++r)
        rb[r] = brow_ok ? *reinterpret_cast<const float4*>(
                              yimg_p + (size_t)((oy0 + b_dy[r]) * p.ymul * p.yW + (ox0 + b_dx[r]) * p.xmul) * p.ycs)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
      return;
    }
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int m = mbase + arow0 + r * AROWS;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < p.M && kvalid) {
        const int sy = src_coord(aoy[r] * p.stride - p.pad_t + kh, He, p.pad_mode, p.ups);
        const int sx = src_coord(aox[r] * p.stride - p.pad_l + kw, We, p.pad_mode, p.ups);
        if (sy >= 0 && sx >= 0)
          v = *reinterpret_cast<const float4*>(p.x + (size_t)an[r] * ximg + (size_t)(sy * p.xW + sx) * p.xcs + ci);
      }
      ra[r] = v;
      advance(an[r], aoy[r], aox[r]);
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int m = mbase + brow0 + r * BROWS;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < p.M && nvalid)
        v = *reinterpret_cast<const float4*>(
            p.y + (size_t)((bn[r] * p.yH + boy[r] * p.ymul + p.yoff) * p.yW + box[r] * p.xmul + p.xoff) * p.ycs + n0 + bcol);
      rb[r] = v;
      advance(bn[r], boy[r], box[r]);
    }
  };
  auto store_tiles = [&](int buf) {
    float* A = As + buf * 32 * BM;
#pragma unroll
    for (int r = 0; r < RA; ++r) *reinterpret_cast<float4*>(A + (arow0 + r * AROWS) * BM + acol) = ra[r];
    float* B = Bs + buf * 32 * BN;
#pragma unroll
    for (int r = 0; r < RB; ++r) *reinterpret_cast<float4*>(B + (brow0 + r * BROWS) * BN + bcol) = rb[r];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x4 nacc[NG > 0 ? NG : 1];
#pragma unroll
  for (int g = 0; g < (NG > 0 ? NG : 1); ++g) nacc[g] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    if constexpr (NG > 0) {
      const float* A = As + buf * 32 * BM + wid * 64 + lane;     // im2col column k of this lane
      const float* B = Bs + buf * 32 * BN + (lane & 31);         // dY[.][lane]
      float xv[32], dv[32];
#pragma unroll
      for (int st = 0; st < 32; ++st) { xv[st] = A[st * BM]; dv[st] = B[st * BN]; }
#pragma unroll
      for (int st = 0; st < 32; ++st) NarrowMac<0, (NG > 0 ? NG : 1)>::run(nacc, dv[st], xv[st]);
      return;
    }
    const float* A = As + buf * 32 * BM + (lane >> 5) * BM + wm * MT * 32 + (lane & 31);
    const float* B = Bs + buf * 32 * BN + (lane >> 5) * BN + wn * NT * 32 + (lane & 31);
    float af[MT][16], bf[NT][16];
#pragma unroll
    for (int st = 0; st < 16; ++st) {
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i][st] = A[2 * st * BM + i * 32];
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j][st] = B[2 * st * BN + j * 32];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < 16; ++st)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][st], bf[j][st], acc[i][j], 0, 0, 0);
  };

  if (mb_begin < mb_end) {
    load_tiles(mb_begin);
    store_tiles(0);
    __syncthreads();
    int cur = 0;
    for (int mb = mb_begin; mb < mb_end; ++mb) {
      const bool more = mb + 1 < mb_end;
      if (more) load_tiles(mb + 1);
      compute(cur);
      if (more) store_tiles(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }
  float* out = p.splits > 1 ? p.slab + (size_t)split * p.K * p.Npad : const_cast<float*>(p.w);
  if constexpr (NG > 0) {
    const int row = kt0 + wid * 64 + lane;
    if (row < p.K) {
#pragma unroll
      for (int g = 0; g < NG; ++g)
        if (4 * g < p.Npad)
          *reinterpret_cast<float4*>(out + (size_t)row * p.Npad + 4 * g) = make_float4(nacc[g][0], nacc[g][1], nacc[g][2], nacc[g][3]);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + wn * NT * 32 + j * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = kt0 + wm * MT * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < p.K && col < p.Npad) out[(size_t)row * p.Npad + col] = acc[i][j][e];
      }
    }
}


// ---------------------------------------------------------------------------------------
// wgrad-type kernel on the LDS-DMA ring: dW[k][n] = sum_m A[m][k] dY[m][n], tile = BMK k-rows x BN columns, reduction
// over 16-pixel stages.  Both LDS tiles are pixel-major ([16 px][BMK] and [16 px][BN]), i.e. plain images of what the
// lanes fetch: a lane owns one 16-byte chunk of k (fixed tap and channels for the whole kernel, so it works for any
// Cin % 4 == 0, also across taps) or of n, and one or more pixels of the stage.  A stage's 16 pixels lie in one image
// (Wo % 16 == 0, or Wo | 16 with Ho*Wo % 16 == 0): its first pixel is tracked by a scalar cursor and every lane adds a
// fixed (dy, dx).  Fragments are ds_read_b64 of two adjacent k-rows / columns (conflict-free, no transposes):
// a lane's MFMA blocks i = 0, 1 hold k-rows (2r, 2r+1), blocks j = 0, 1 columns (2c, 2c+1).
// ---------------------------------------------------------------------------------------
template <int WGM, int WGN>
struct DmaWgTile {
  static constexpr int NW = WGM * WGN, BMK = 64 * WGM, BN = 64 * WGN, PX = 16, NST = 3;
  static constexpr int A_FL = PX * BMK, B_FL = PX * BN, ST_FL = A_FL + B_FL;
  static constexpr int LPRA = BMK / 4, LPRB = BN / 4;              // lanes per pixel row
  static constexpr int PPIA = 64 / LPRA > 0 ? 64 / LPRA : 1, PPIB = 64 / LPRB;   // pixels per instruction
  static constexpr int AI = (PX * LPRA / 64) / NW, BI = (PX * LPRB / 64) / NW;   // instructions per wave per stage
  static constexpr int SMEM = NST * ST_FL * 4;
  static_assert(LPRA <= 64 && AI >= 1 && BI >= 1, "tile shape");
};

// SPLIT: 0 = v_mfma_f32_32x32x2_f32, 1 = three bf16 planes per operand (six MFMAs per product), 2 = two fp16 planes of the operands
// scaled by powers of two from their amax (three MFMAs; x_amax / dy_amax: 256 floats each whose maximum is the operand's amax),
// 3 = ONE fp16 plane of the scaled operands (one MFMA: the reduced-precision configuration)
// PLANE: both operands are plain [M][C] matrices (the batched Winograd-domain reductions dU[p] = V[p]^T dM[p]: 1x1 taps, stride 1,
// no padding, output map = the row index).  The generic loader recomputes the im2col source of every piece every stage (~90 VALU
// + ~60 SALU per wave and stage, measured: the wave spends 41 % of its time issuing, 123 % of a SIMD's port at 3 waves); here a
// piece's offset is a per-lane constant and the stage advances through the scalar offset of the buffer load.
// PAIR (bit 0: x, bit 1: dy): that operand is stored in pair form, x_amax / dy_amax then point at the int exponent of its producer
template <int WGM, int WGN, int SPLIT, bool PLANE = false, int PAIR = 0>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_wgrad_dma_kernel(GemmP p, DmaSched sc, const float* x_amax, const float* dy_amax) {
#if defined(__HIP_DEVICE_COMPILE__)
  using T = DmaWgTile<WGM, WGN>;
  constexpr int BMK = T::BMK, BN = T::BN, PX = T::PX, NST = T::NST, AI = T::AI, BI = T::BI;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wid / WGN, wn = wid % WGN;

  int u = blockIdx.x, gtile, split = 0, nsplit = 1, tt = 0;
  if (u < sc.full) {
    gtile = xcd_swizzle(u, sc.full);
  } else {
    u -= sc.full;
    tt = u / sc.tail_s; split = u - tt * sc.tail_s; nsplit = sc.tail_s;
    gtile = sc.full + tt;
  }
  const int z = gtile / sc.tiles_per_z, tile = gtile - z * sc.tiles_per_z;
  const int tile_n = tile % p.tiles_n, tile_k = tile / p.tiles_n;
  const int kt0 = tile_k * BMK, n0 = tile_n * BN;
  p.x += (size_t)z * p.x_bs; p.y += (size_t)z * p.y_bs; p.w += (size_t)z * p.w_bs;
  if (p.phases) { const int a = z >> 1, b = z & 1; p.pad_t -= a; p.pad_l -= b; p.yoff = a; p.xoff = b; }
  const int HoWo = p.Ho * p.Wo;
  const int nimg = p.M / HoWo;
  const int nmb_all = p.M / PX;
  const int mb_begin = nsplit > 1 ? split * sc.per_split : 0;
  const int mb_end = nsplit > 1 ? min(nmb_all, mb_begin + sc.per_split) : nmb_all;

  const unsigned x_bytes = (unsigned)((((size_t)p.xH * p.xW * nimg - 1) * p.xcs + p.xC) * 4);
  const unsigned y_bytes = (unsigned)((((size_t)p.yH * p.yW * nimg - 1) * p.ycs + p.yC) * 4);
  const i32x4 rsA = make_rsrc(p.x, x_bytes), rsB = make_rsrc(p.y, y_bytes);
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
  const int He = p.xH << p.ups, We = p.xW << p.ups;
  const bool wide = (p.Wo % PX) == 0;

  // ---- per-lane constants: A chunk -> (tap, ci); pixel offsets (dy, dx) of the lane's pixels inside a stage
  const int ak = kt0 + 4 * (lane % T::LPRA);
  const bool kvalid = ak < p.K;
  const int atap = ak / p.xC, aci = ak - atap * p.xC;
  const int akh = atap / p.KW, akw = atap - akh * p.KW;
  int a_dy[AI], a_dx[AI], b_dy[BI], b_dx[BI];
#pragma unroll
  for (int r = 0; r < AI; ++r) {
    const int j = (wid * AI + r) * T::PPIA + lane / T::LPRA;
    a_dy[r] = wide ? 0 : j / p.Wo; a_dx[r] = wide ? j : j % p.Wo;
  }
  const int bn = n0 + 4 * (lane % T::LPRB);
  const bool nvalid = bn < p.yC;
#pragma unroll
  for (int r = 0; r < BI; ++r) {
    const int j = (wid * BI + r) * T::PPIB + lane / T::LPRB;
    b_dy[r] = wide ? 0 : j / p.Wo; b_dx[r] = wide ? j : j % p.Wo;
  }
  // scalar cursor of the next stage to issue
  int c_n, c_oy, c_ox;
  {
    const int mbase = mb_begin * PX;
    c_n = mbase / HoWo; const int rem = mbase - c_n * HoWo;
    c_oy = rem / p.Wo; c_ox = rem - c_oy * p.Wo;
  }
  const int rows_per_stage = wide ? 0 : PX / p.Wo;
  // PLANE: per-lane byte offsets of the lane's pieces inside a stage (pixel j of the stage, its 16-byte chunk); out of range = zero fill
  unsigned pa_voff[AI], pb_voff[BI];
  unsigned p_stage = (unsigned)mb_begin;          // next stage to issue
  if constexpr (PLANE) {
#pragma unroll
    for (int r = 0; r < AI; ++r) pa_voff[r] = kvalid ? (unsigned)(a_dx[r] * p.xcs + ak) * 4u : DMA_OOB;
#pragma unroll
    for (int r = 0; r < BI; ++r) pb_voff[r] = nvalid ? (unsigned)(b_dx[r] * p.ycs + bn) * 4u : DMA_OOB;
  }
  auto issue = [&](int st) {
    const unsigned As = lds0 + (unsigned)(st * T::ST_FL) * 4u, Bs = As + T::A_FL * 4u;
    if constexpr (PLANE) {
      const unsigned sa = p_stage * (unsigned)(PX * 4) * (unsigned)p.xcs, sb = p_stage * (unsigned)(PX * 4) * (unsigned)p.ycs;
#pragma unroll
      for (int r = 0; r < AI; ++r) lds_dma16c(pa_voff[r], rsA, sa, As + (unsigned)(wid * AI + r) * 1024u);
#pragma unroll
      for (int r = 0; r < BI; ++r) lds_dma16c(pb_voff[r], rsB, sb, Bs + (unsigned)(wid * BI + r) * 1024u);
      p_stage += 1;
      return;
    }
    const int xin = c_n * p.xH * p.xW, yin = c_n * p.yH;
#pragma unroll
    for (int r = 0; r < AI; ++r) {
      unsigned off = DMA_OOB;
      const int sy = src_coord((c_oy + a_dy[r]) * p.stride - p.pad_t + akh, He, p.pad_mode, p.ups);
      const int sx = src_coord((c_ox + a_dx[r]) * p.stride - p.pad_l + akw, We, p.pad_mode, p.ups);
      if (kvalid && sy >= 0 && sx >= 0) off = (unsigned)((xin + sy * p.xW + sx) * p.xcs + aci) * 4u;
      lds_dma16(off, rsA, 0u, As + (unsigned)(wid * AI + r) * 1024u);
    }
#pragma unroll
    for (int r = 0; r < BI; ++r) {
      unsigned off = DMA_OOB;
      if (nvalid)
        off = (unsigned)(((yin + (c_oy + b_dy[r]) * p.ymul + p.yoff) * p.yW + (c_ox + b_dx[r]) * p.xmul + p.xoff) * p.ycs + bn) * 4u;
      lds_dma16(off, rsB, 0u, Bs + (unsigned)(wid * BI + r) * 1024u);
    }
    if (wide) {
      c_ox += PX;
      if (c_ox >= p.Wo) { c_ox = 0; c_oy += 1; if (c_oy >= p.Ho) { c_oy = 0; c_n += 1; } }
    } else {
      c_oy += rows_per_stage;
      if (c_oy >= p.Ho) { c_oy = 0; c_n += 1; }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int kx = 0, ky = 0;
  if constexpr (SPLIT >= 2) {
    if constexpr (PAIR & 1) kx = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(x_amax));
    else kx = __builtin_amdgcn_readfirstlane(scale_exp(amax256(x_amax, lane), PC_TOP_A));
    if constexpr (PAIR & 2) ky = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(dy_amax));
    else ky = __builtin_amdgcn_readfirstlane(scale_exp(amax256(dy_amax, lane), PC_TOP_A));
  }
  const float sx = pow2f(kx), sy = pow2f(ky);
  const int h = lane >> 5, l31 = lane & 31;
  const int a_rd = (8 * h) * BMK + wm * 64 + 2 * l31;
  const int b_rd = T::A_FL + (8 * h) * BN + wn * 64 + 2 * l31;
  // pixel order inside a stage: step s multiplies pixel s (lanes 0-31) and pixel 8 + s (lanes 32-63)
  auto compute = [&](int st) {
    const float* S = smem + st * T::ST_FL;
    float af[2][8], bf[2][8];
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const float2 a = *reinterpret_cast<const float2*>(S + a_rd + s8 * BMK);
      const float2 b = *reinterpret_cast<const float2*>(S + b_rd + s8 * BN);
      af[0][s8] = a.x; af[1][s8] = a.y; bf[0][s8] = b.x; bf[1][s8] = b.y;
    }
    if constexpr (SPLIT == 3) {
      split_mma_2x2_h1(acc, af, bf, sx, sy);
    } else if constexpr (SPLIT == 2 && PAIR != 0) {
      u32x4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if constexpr (PAIR & 1) pair8(af[i], ah[i], al[i]); else split8h(af[i], sx, ah[i], al[i]);
        if constexpr (PAIR & 2) pair8(bf[i], bh[i], bl[i]); else split8h_rn(bf[i], sy, bh[i], bl[i]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 c = acc[i][j];
          c = mma_f16(al[i], bh[j], c); c = mma_f16(ah[i], bl[j], c); c = mma_f16(ah[i], bh[j], c);
          acc[i][j] = c;
        }
    } else if constexpr (SPLIT == 2) {
      split_mma_2x2_h(acc, af, bf, sx, sy);
    } else {
      static_assert(SPLIT == 0, "0: f32 MFMA, 2: two fp16 planes, 3: one fp16 plane");
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s8], bf[j][s8], acc[i][j], 0, 0, 0);
    }
  };

  if (mb_begin < mb_end) {
    issue(0);
    if (mb_begin + 1 < mb_end) issue(1);
    int st = 0;
    for (int mb = mb_begin; mb < mb_end; ++mb) {
      if (mb + 1 < mb_end) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(AI + BI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int st2 = st + 2; if (st2 >= NST) st2 -= NST;
      if (mb + 2 < mb_end) issue(st2);
      compute(st);
      st = st + 1 == NST ? 0 : st + 1;
    }
  }

  if constexpr (SPLIT >= 2) {                              // remove the operand scales (two exact power-of-two factors)
    const float cx = pow2f(-kx), cy = pow2f(-ky);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = (acc[i][j][e] * cx) * cy;
  }
  // ---- epilogue: lane holds k-rows kt0 + wm*64 + 2*rr + i (rr = (e&3) + 8*(e>>2) + 4*h), columns n0 + wn*64 + 2*l31 + j
  const int colr = wn * 64 + 2 * l31;
  if (nsplit > 1) {
    float* slab = p.slab + ((size_t)(tt * nsplit + split) * BMK) * BN;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * 64 + 2 * ((e & 3) + 8 * (e >> 2) + 4 * h) + i;
        *reinterpret_cast<float2*>(slab + (size_t)row * BN + colr) = make_float2(acc[i][0][e], acc[i][1][e]);
      }
    return;
  }
  float* out = const_cast<float*>(p.w);
  const int col = n0 + colr;
  if (col < p.Npad) {                           // Npad % 4 == 0 and col even: col + 1 < Npad too
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = kt0 + wm * 64 + 2 * ((e & 3) + 8 * (e >> 2) + 4 * h) + i;
        if (row < p.K) *reinterpret_cast<float2*>(out + (size_t)row * p.Npad + col) = make_float2(acc[i][0][e], acc[i][1][e]);
      }
  }
#endif
}

template <int BMK, int BN>
__global__ __launch_bounds__(256) void wgrad_dma_reduce_kernel(GemmP p, DmaSched sc) {
  const int tt = blockIdx.y;
  const int e4 = blockIdx.x * 256 + threadIdx.x;
  if (e4 >= BMK * BN / 4) return;
  const int r = e4 / (BN / 4), c4 = (e4 - r * (BN / 4)) * 4;
  const int gtile = sc.full + tt;
  const int z = gtile / sc.tiles_per_z, tile = gtile - z * sc.tiles_per_z;
  const int tile_n = tile % p.tiles_n, tile_k = tile / p.tiles_n;
  const int row = tile_k * BMK + r, col = tile_n * BN + c4;
  if (row >= p.K || col >= p.Npad) return;
  const float* sl = p.slab + ((size_t)tt * sc.tail_s * BMK + r) * BN + c4;
  float4 a = *reinterpret_cast<const float4*>(sl);
  for (int s = 1; s < sc.tail_s; ++s) {
    const float4 b = *reinterpret_cast<const float4*>(sl + (size_t)s * BMK * BN);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  float* out = const_cast<float*>(p.w) + (size_t)z * p.w_bs;
  *reinterpret_cast<float4*>(out + (size_t)row * p.Npad + col) = a;
}

// sums the `splits` slabs of n floats: 16 outputs (float4) x 16 slab groups per block -- group g adds slabs g, g+16, ...
// in index order, then the 16 group sums are added in index order (fixed order: deterministic).  With hundreds of slabs
// of a small weight tensor (first-layer weight gradients: 250-500 slabs of 24 K floats) one thread per output walking
// all slabs is latency-bound (98 us); spreading the slab axis over the block makes it a 10 us kernel.
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* slab, float* out, size_t n, int splits, size_t slab_bs, size_t out_bs) {
  __shared__ float4 red[256];
  slab += (size_t)blockIdx.y * slab_bs; out += (size_t)blockIdx.y * out_bs;
  const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
  const size_t i = ((size_t)blockIdx.x * 16 + o) * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n)
    for (int sp = g; sp < splits; sp += 16) {
      const float4 b = *reinterpret_cast<const float4*>(slab + (size_t)sp * n + i);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
  red[threadIdx.x] = a;
  __syncthreads();
  if (g == 0 && i < n) {
    float4 t = red[o];
    for (int k = 1; k < 16; ++k) { const float4 b = red[k * 16 + o]; t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w; }
    *reinterpret_cast<float4*>(out + i) = t;
  }
}

// ---------------------------------------------------------------------------------------
// naive references (verification only)
// ---------------------------------------------------------------------------------------
__global__ void conv_fwd_naive_kernel(GemmP p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)p.M * p.Cout) return;
  const int m = (int)(i / p.Cout), co = (int)(i - (size_t)m * p.Cout);
  const int HoWo = p.Ho * p.Wo;
  const int n = m / HoWo, rem = m - n * HoWo;
  const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
  const int He = p.xH << p.ups, We = p.xW << p.ups;
  float acc = 0.f;
  for (int kh = 0; kh < p.KH; ++kh)
    for (int kw = 0; kw < p.KW; ++kw) {
      const int sy = src_coord(oy * p.stride - p.pad_t + kh, He, p.pad_mode, p.ups);
      const int sx = src_coord(ox * p.stride - p.pad_l + kw, We, p.pad_mode, p.ups);
      if (sy < 0 || sx < 0) continue;
      const float* xp = p.x + (size_t)n * p.xH * p.xW * p.xcs + (size_t)(sy * p.xW + sx) * p.xcs;
      const float* wp = p.w + (size_t)((kh * p.KW + kw) * p.xC) * p.Npad + co;
      for (int ci = 0; ci < p.xC; ++ci) acc = fmaf(xp[ci], wp[(size_t)ci * p.Npad], acc);
    }
  if (p.bias) acc += p.bias[co];
  acc = act_apply(acc, p.act);
  float* dst = p.y + (size_t)((n * p.yH + oy * p.ymul + p.yoff) * p.yW + ox * p.xmul + p.xoff) * p.ycs + co;
  if (p.accumulate) acc += *dst;
  *dst = acc;
}

__global__ void conv_wgrad_naive_kernel(GemmP p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)p.K * p.Npad) return;
  const int k = (int)(i / p.Npad), co = (int)(i - (size_t)k * p.Npad);
  float* dw = const_cast<float*>(p.w);
  if (co >= p.Cout) { dw[i] = 0.f; return; }
  const int tap = k / p.xC, ci = k - tap * p.xC;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int He = p.xH << p.ups, We = p.xW << p.ups;
  const int HoWo = p.Ho * p.Wo;
  double acc = 0.0;
  for (int m = 0; m < p.M; ++m) {
    const int n = m / HoWo, rem = m - n * HoWo;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const int sy = src_coord(oy * p.stride - p.pad_t + kh, He, p.pad_mode, p.ups);
    const int sx = src_coord(ox * p.stride - p.pad_l + kw, We, p.pad_mode, p.ups);
    if (sy < 0 || sx < 0) continue;
    const float xv = p.x[(size_t)n * p.xH * p.xW * p.xcs + (size_t)(sy * p.xW + sx) * p.xcs + ci];
    const float dv = p.y[(size_t)((n * p.yH + oy * p.ymul + p.yoff) * p.yW + ox * p.xmul + p.xoff) * p.ycs + co];
    acc += (double)xv * dv;
  }
  dw[i] = (float)acc;
}

// ---------------------------------------------------------------------------------------
// per-launch profiling (HIP events on the launch stream)
// ---------------------------------------------------------------------------------------
namespace {
struct ProfRec { std::string name; double flops; hipEvent_t a, b; };
int g_prof = 0;
std::vector<ProfRec> g_recs;
struct ProfScope {
  hipStream_t st; bool on; ProfRec r;
  ProfScope(const Stream& s, const char* name, double flops) : st(hs(s)), on(g_prof != 0) {
    if (route_on()) route_note(name);
    if (!on) return;
    r.name = name; r.flops = flops;
    (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.a, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.b, st);
    g_recs.push_back(r);
  }
};
}  // namespace
// ---- swn_probe_mfma: what the matrix pipe sustains for the ring kernels' instruction mix, operands in registers ----------------
__global__ __launch_bounds__(256, 4) void mfma_probe_kernel(int iters, int zeros, unsigned long long* clk, float* sink) {
  unsigned long long c0 = 0, r0 = 0;
  const bool me = blockIdx.x % 61 == 0 && blockIdx.x / 61 < 16 && threadIdx.x == 0;
  if (me) { c0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
  // fp16 bit patterns from a per-lane hash: sign, exponents 2^-3 .. 2^0, random mantissas (finite, products stay far from overflow)
  unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  auto word = [&]() {
    h = h * 1664525u + 1013904223u;
    const unsigned lo = (h >> 3) & 0x83ffu, hi = (h >> 17) & 0x83ffu;
    return zeros ? 0u : ((lo | 0x3000u | ((h & 3u) << 10)) | ((hi | 0x3000u | (((h >> 2) & 3u) << 10)) << 16));
  };
  u32x4 ah, al, bh[4], bl[4];
  for (int q = 0; q < 4; ++q) { ah[q] = word(); al[q] = word(); }
  for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) { bh[j][q] = word(); bl[j][q] = word(); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = mma_f16(al, bh[j], acc[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = mma_f16(ah, bl[j], acc[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = mma_f16(ah, bh[j], acc[j]);
    asm volatile("" ::: "memory");
  }
  float t = 0.f;
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) t += acc[j][e];
  if (t == 12345.678f) sink[0] = t;                                  // (keeps the accumulators alive)
  if (me) { clk[2 * (blockIdx.x / 61)] = __builtin_readcyclecounter() - c0; clk[2 * (blockIdx.x / 61) + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
void probe_mfma(Stream& s, int zeros, int iters, float* out4) {
  if (!s.ws || s.ws_bytes < 4096) throw Error(1, "probe_mfma: the stream scratch is missing");
  unsigned long long* clk = reinterpret_cast<unsigned long long*>(s.ws);
  float* sink = reinterpret_cast<float*>(s.ws + 512);
  const int blocks = 1024;
  hipEvent_t e0, e1;
  SWN_HIP_CHECK(hipEventCreate(&e0)); SWN_HIP_CHECK(hipEventCreate(&e1));
  // steady state, not a burst: the chip's power management settles over milliseconds (a single launch after an idle gap runs
  // 30-40 % faster than the same launch inside a train of them).  Twelve launches back to back; the last six are timed.
  constexpr int WARM = 6, TIMED = 6;
  SWN_HIP_CHECK(hipMemsetAsync(clk, 0, 256, hs(s)));
  for (int rep = 0; rep < WARM + TIMED; ++rep) {
    if (rep == WARM) SWN_HIP_CHECK(hipEventRecord(e0, hs(s)));
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, hs(s), iters, zeros, clk, sink);
  }
  SWN_HIP_CHECK(hipEventRecord(e1, hs(s)));
  SWN_HIP_CHECK(hipEventSynchronize(e1));
  float best = 0.f; SWN_HIP_CHECK(hipEventElapsedTime(&best, e0, e1));
  best /= TIMED;
  SWN_HIP_CHECK(hipEventDestroy(e0)); SWN_HIP_CHECK(hipEventDestroy(e1));
  unsigned long long h[32];
  SWN_HIP_CHECK(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
  int wall_khz = 0, dev = 0;
  SWN_HIP_CHECK(hipGetDevice(&dev));
  SWN_HIP_CHECK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev));
  double cs = 0, rs = 0;
  for (int i = 0; i < 16; ++i) { cs += (double)h[2 * i]; rs += (double)h[2 * i + 1]; }
  const double ghz = rs > 0 ? cs / rs * wall_khz * 1e-6 : 0.0;
  const double mfmas = (double)blocks * 4 * iters * 12;               // per launch
  out4[0] = (float)(mfmas * 32768.0 / (best * 1e-3) * 1e-12);
  out4[1] = (float)ghz;
  out4[2] = best;
  out4[3] = ghz > 0 ? (float)(mfmas * 32.0 / 1024.0 / (best * 1e-3 * ghz * 1e9)) : 0.f;
}

void prof_enable(int on) { g_prof = on; }
void prof_reset() {
  for (auto& r : g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_recs.clear();
}
int prof_report(char* buf, int len) {
  std::map<std::string, std::array<double, 3>> agg;
  for (auto& r : g_recs) {
    (void)hipEventSynchronize(r.b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    auto& a = agg[r.name];
    a[0] += 1; a[1] += ms; a[2] += r.flops;
  }
  std::string out;
  for (auto& kv : agg) {
    char line[256];
    snprintf(line, sizeof line, "%s %.0f %.6f %.6e\n", kv.first.c_str(), kv.second[0], kv.second[1], kv.second[2]);
    out += line;
  }
  if (buf && len > 0) { strncpy(buf, out.c_str(), len - 1); buf[len - 1] = 0; }
  return (int)out.size();
}

// ---------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------
static GemmP make_params(const TView& x, const Gather& g, const TView& y, OutMap om, int phases = 0, int batch = 1) {
  if (phases) {
    if (phases != 4 || batch > 1 || om.ymul != 2 || om.xmul != 2) throw Error(1, "conv: bad sub-pixel phase launch");
    om.yoff = om.xoff = 1;               // bounds check below against the farthest phase
  }
  if (x.C % 4 || x.cs % 4 || y.cs % 4) throw Error(1, "conv: channel counts/strides must be multiples of 4");
  if (((uintptr_t)x.p & 15) || ((uintptr_t)y.p & 15)) throw Error(1, "conv: views must be 16-byte aligned");
  GemmP p{};
  p.x = x.p; p.xH = x.H; p.xW = x.W; p.xC = x.C; p.xcs = x.cs;
  p.KH = g.KH; p.KW = g.KW; p.stride = g.stride; p.pad_t = g.pad_t; p.pad_l = g.pad_l;
  p.pad_mode = g.pad_mode; p.ups = g.ups; p.Ho = g.Ho; p.Wo = g.Wo;
  p.M = x.N * g.Ho * g.Wo;
  p.K = g.KH * g.KW * x.C;
  p.y = y.p; p.yH = y.H; p.yW = y.W; p.ycs = y.cs; p.yC = y.C;
  p.ymul = om.ymul; p.yoff = om.yoff; p.xmul = om.xmul; p.xoff = om.xoff;
  p.splits = 1; p.per_split = 1 << 30;
  p.phases = phases;
  if (phases) p.yoff = p.xoff = 0;
  if ((g.Ho - 1) * om.ymul + om.yoff >= y.H || (g.Wo - 1) * om.xmul + om.xoff >= y.W || y.N != x.N)
    throw Error(1, "conv: output map exceeds the output view");
  return p;
}

// Wave-quantisation-aware split factor.  `ntiles` output tiles, each `work` reduction steps
// long, run on `slots` concurrently resident workgroups (CUs x workgroups/CU).  A tile count just
// above a multiple of `slots` leaves most of the chip idle in the last round (e.g. 576 wgrad
// tiles of a resblock conv on 512 slots = 56 % efficiency); splitting the reduction s ways
// trades that for s slabs summed by a cheap second kernel.  Cost model: rounds x (steps per
// block + fixed prologue/epilogue) + slab traffic.
static int choose_splits(int ntiles, int work, int slots, int min_work, size_t slab_bytes, size_t ws_bytes) {
  int best = 1;
  double best_cost = 1e300;
  const int max_s = std::max(1, std::min(512, work / std::max(min_work, 1)));
  for (int sp = 1; sp <= max_s; ++sp) {
    if (sp > 1 && slab_bytes * sp > ws_bytes) break;
    const int per = ceil_div(work, sp);
    const int eff = ceil_div(work, per);
    const double rounds = std::ceil((double)ntiles * eff / slots);
    double cost = rounds * (per + 4.0);
    if (eff > 1) cost += 0.02 * eff * ((double)ntiles / slots) + 1.0;   // slab write + reduce launch
    if (cost < best_cost * 0.97) { best_cost = cost; best = eff; }
  }
  return best;
}

static bool prof_detail() {
  static const bool on = getenv("SWN_PROF_DETAIL") != nullptr;
  return on || route_on();
}

template <typename K>
static void set_smem(K kernel, int bytes) {
  SWN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
}

template <int MT, int NT, int WGM, int WGN>
static void launch_fwd(Stream& s, GemmP& p, bool fast, int batch) {
  using T = Tile<MT, NT, WGM, WGN>;
  const int tiles_m = ceil_div(p.M, T::BM);
  p.tiles_n = ceil_div(p.Npad, T::BN);
  p.ntiles = tiles_m * p.tiles_n;
  const int nkb = ceil_div(p.K, 32);
  const int slots = 256 * (T::SMEM_FWD > 80 * 1024 ? 1 : (T::SMEM_FWD > 64 * 1024 ? 2 : 3));
  p.slab_bs = (size_t)p.M * p.Npad;      // per batch, per split
  const int splits = choose_splits(p.ntiles * batch, nkb, slots, 8, (size_t)p.M * p.Npad * 4 * batch, s.ws_bytes);
  p.splits = splits;
  p.per_split = ceil_div(nkb, splits);
  p.splits = ceil_div(nkb, p.per_split);
  p.slab = reinterpret_cast<float*>(s.ws);
  p.slab_bs = (size_t)p.M * p.Npad * p.splits;
  dim3 grid(p.ntiles, p.splits, batch);
  char pname[96];
  if (prof_detail())
    snprintf(pname, sizeof pname, "conv_fwd_%dx%d_%s[M%d,N%d,K%d,s%d]", T::BM, T::BN, fast ? "fast" : "generic", p.M,
             p.Cout, p.K, p.splits);
  else
    snprintf(pname, sizeof pname, "conv_fwd_%dx%d_%s", T::BM, T::BN, fast ? "fast" : "generic");
  ProfScope prof(s, pname, 2.0 * p.M * p.Cout * p.K * batch);
  if (fast) {
    static bool once = (set_smem(conv_fwd_kernel<MT, NT, WGM, WGN, true>, T::SMEM_FWD), true);
    (void)once;
    hipLaunchKernelGGL((conv_fwd_kernel<MT, NT, WGM, WGN, true>), grid, dim3(64 * WGM * WGN), T::SMEM_FWD, hs(s), p);
  } else {
    static bool once = (set_smem(conv_fwd_kernel<MT, NT, WGM, WGN, false>, T::SMEM_FWD), true);
    (void)once;
    hipLaunchKernelGGL((conv_fwd_kernel<MT, NT, WGM, WGN, false>), grid, dim3(64 * WGM * WGN), T::SMEM_FWD, hs(s), p);
  }
  check_launch("conv_fwd");
  if (p.splits > 1) {
    const size_t total = (size_t)p.M * ((p.Cout + 3) / 4);
    hipLaunchKernelGGL(conv_fwd_reduce_kernel, dim3((unsigned)((total + 255) / 256), 1, batch), dim3(256), 0, hs(s), p);
    check_launch("conv_fwd_reduce");
  }
}

template <int NG>
static void launch_fwd_narrow(Stream& s, GemmP& p, bool fast, int batch) {
  using T = NarrowTile;
  p.tiles_n = 1;
  p.ntiles = ceil_div(p.M, T::BM);
  const int nkb = ceil_div(p.K, T::BK);
  const int slots = 256 * 3;
  const int splits = choose_splits(p.ntiles * batch, nkb, slots, 16, (size_t)p.M * p.Npad * 4 * batch, s.ws_bytes);
  p.per_split = ceil_div(nkb, splits);
  p.splits = ceil_div(nkb, p.per_split);
  p.slab = reinterpret_cast<float*>(s.ws);
  p.slab_bs = (size_t)p.M * p.Npad * p.splits;
  dim3 grid(p.ntiles, p.splits, batch);
  char pname[96];
  if (prof_detail())
    snprintf(pname, sizeof pname, "conv_fwd_narrow%d_%s[M%d,N%d,K%d,s%d]", 4 * NG, fast ? "fast" : "generic", p.M, p.Cout, p.K,
             p.splits);
  else
    snprintf(pname, sizeof pname, "conv_fwd_narrow%d_%s", 4 * NG, fast ? "fast" : "generic");
  ProfScope prof(s, pname, 2.0 * p.M * p.Cout * p.K * batch);
  if (fast) hipLaunchKernelGGL((conv_fwd_narrow_kernel<NG, true>), grid, dim3(256), T::SMEM, hs(s), p);
  else hipLaunchKernelGGL((conv_fwd_narrow_kernel<NG, false>), grid, dim3(256), T::SMEM, hs(s), p);
  check_launch("conv_fwd_narrow");
  if (p.splits > 1) {
    const size_t total = (size_t)p.M * ((p.Cout + 3) / 4);
    hipLaunchKernelGGL(conv_fwd_reduce_kernel, dim3((unsigned)((total + 255) / 256), 1, batch), dim3(256), 0, hs(s), p);
    check_launch("conv_fwd_reduce");
  }
}


// the four sub-pixel phases of a narrow (Cout <= 20) transposed conv in one launch: tail_fwd4_kernel<NG, true>
template <int NG>
static void launch_fwd_phase4(Stream& s, GemmP& p) {
  p.tiles_n = 1; p.ntiles = ceil_div(p.M, TailTile::BM);
  static bool once = (set_smem(tail_fwd4_kernel<NG, true>, TailTile::SMEM), true);
  (void)once;
  char pname[96];
  if (prof_detail()) snprintf(pname, sizeof pname, "conv_fwd_phase4_narrow%d[M%d,N%d,K%d]", 4 * NG, p.M, p.Cout, p.K);
  else snprintf(pname, sizeof pname, "conv_fwd_phase4_narrow%d", 4 * NG);
  ProfScope prof(s, pname, 2.0 * p.M * p.Cout * p.K * 4);
  hipLaunchKernelGGL((tail_fwd4_kernel<NG, true>), dim3(p.ntiles), dim3(256), TailTile::SMEM, hs(s), p);
  check_launch("conv_fwd_phase4");
}

// read per launch (tests and A/B measurements toggle it): 0 = v_mfma_f32_32x32x2_f32 main loop, default = bf16 split
static bool split_on() { return !(getenv("SWN_SPLIT") && atoi(getenv("SWN_SPLIT")) == 0); }
// ---- LDS-DMA forward kernel: schedule + launch ------------------------------------------------------------------
// T tiles of `work` stages on `slots` resident workgroups.  Whole rounds run one tile per unit; the remainder tiles
// (all tiles when T < slots) are split s ways along K so that the last round is as full as the others.
// `unit` = stage-time of this tile configuration relative to the 128x128 one (waves per SIMD / 3); *cost_out receives the
// estimated launch time in 128x128 stage units, comparable across tile configurations.
static DmaSched plan_dma(int tiles_total, int tiles_per_z, int work, int slots, size_t tile_bytes, size_t ws_bytes,
                         double* cost_out = nullptr, double unit = 1.0) {
  DmaSched sc{};
  sc.tiles_per_z = tiles_per_z;
  const int rounds = tiles_total / slots;
  int rem = tiles_total - rounds * slots;
  sc.full = rounds * slots; sc.tail_tiles = rem; sc.tail_s = 1; sc.per_split = work;
  if (cost_out) *cost_out = rounds * (work + 4.0) * unit;
  if (rem == 0) return sc;
  // A remainder behind at least one whole round runs UN-SPLIT (round 5).  The two-plane loops are bound by the chip's power, not
  // by issue (tools/tile_lab.hip: 1.13-1.24 GHz at > 85 % matrix-pipe occupancy on random operands, 2.1-2.4 GHz for the same
  // launch with the matrix instructions removed): a last round that occupies few CUs runs at nearly twice the clock, and the
  // split's slab round trip + reduce launch cost what it saves (36 Winograd planes of 512 x 1024 x 1024 = 1152 tiles on 1024
  // slots: 128 us un-split in the lab, 135 + 13 us split in the product; whole C2 step -0.29 ms, profiles/native_ab_r05.txt).
  // The same holds for a launch that fills at least half of the slots (two workgroups per CU in the 4-per-CU configuration): in the
  // lab 512 / 768 / 1024 tiles of the resblock shape take 51 / 72 / 95 us un-split -- the same 340-360 fp32-equivalent TFLOP/s, at
  // 1.51 / 1.36 / 1.10 GHz -- so splitting K to "fill the chip" buys nothing there either (below half, a CU runs too few waves to
  // keep its matrix pipe fed and the split still pays: 128 tiles 33 us).  SWN_TAIL_SPLIT=0 restores every split, =2 only the
  // behind-a-whole-round rule (A/B; read per launch).
  {
    const char* e = getenv("SWN_TAIL_SPLIT");
    const int mode = e ? atoi(e) : 1;
    if (mode != 0 && (rounds >= 1 || (mode == 1 && 2 * rem >= slots))) {
      sc.full = tiles_total; sc.tail_tiles = 0;
      if (cost_out) *cost_out += (work + 4.0) * unit * (rounds >= 1 ? 0.6 : (double)rem / slots);
      return sc;
    }
  }
  // cost in units of one stage-time of a resident workgroup (32 MFMAs per wave, three waves sharing a SIMD: ~2.7 us):
  // tail rounds x (stages per unit + ~4 of prologue / epilogue) + the slab round trip of the split tiles at ~5 TB/s
  // (13.5 MB per unit) + the reduce launch.  Calibrated on the Winograd-plane launches of the warp step (M 800 x 36
  // planes: 3 whole rounds 0.559 ms, 2 + a 3-way split tail 0.535 ms).
  int best = 1;
  double best_cost = 1e300;
  const int max_s = std::max(1, std::min(1024, work / 8));
  for (int sp = 1; sp <= max_s; ++sp) {
    if (sp > 1 && (size_t)rem * sp * tile_bytes > ws_bytes) break;
    const int per = ceil_div(work, sp), eff = ceil_div(work, per);
    const double tail_rounds = std::ceil((double)rem * eff / slots);
    double cost = tail_rounds * (per + 4.0);
    if (eff > 1) cost += (double)rem * eff * (double)tile_bytes * 2.0 / (13.5e6 * unit) + 2.0;
    if (cost < best_cost * 0.985) { best_cost = cost; best = eff; }
  }
  if (cost_out) *cost_out += best_cost * unit;
  sc.tail_s = best;
  sc.per_split = ceil_div(work, best);
  sc.tail_s = ceil_div(work, sc.per_split);
  if (sc.tail_s == 1) { sc.full = tiles_total; sc.tail_tiles = 0; }
  return sc;
}

// resident workgroups per CU: LDS (160 KB) and registers -- the kernels use ~104 VGPRs: 4 waves per SIMD (16 per CU) for
// the 8-wave tiles, and 3 per SIMD are kept for the 4-wave tiles (3 x 48 KB of LDS)
template <int WGM, int WGN>
static constexpr int dma_wg_per_cu() {
  using T = DmaTile<WGM, WGN>;
  return T::NW >= 8 ? std::min(160 * 1024 / T::SMEM, 16 / T::NW) : std::min(160 * 1024 / T::SMEM, 12 / T::NW);
}
template <int WGM, int WGN>
static DmaSched plan_fwd_dma(const GemmP& p, int nb, size_t ws_bytes, double* cost) {
  using T = DmaTile<WGM, WGN>;
  const int ntiles = ceil_div(p.M, T::BM) * ceil_div(p.Npad, T::BN);
  constexpr int wg = dma_wg_per_cu<WGM, WGN>();
  return plan_dma(ntiles * nb, ntiles, p.K / T::BK, 256 * wg, (size_t)T::BM * T::BN * 4, ws_bytes, cost, wg * T::NW / 12.0);
}

template <int WGM, int WGN>
static void launch_fwd_dma(Stream& s, GemmP& p, int nb) {
  using T = DmaTile<WGM, WGN>;
  const int tiles_m = ceil_div(p.M, T::BM);
  p.tiles_n = ceil_div(p.Npad, T::BN);
  p.ntiles = tiles_m * p.tiles_n;
  const DmaSched sc = plan_fwd_dma<WGM, WGN>(p, nb, s.ws_bytes, nullptr);
  p.slab = reinterpret_cast<float*>(s.ws);
  p.splits = sc.tail_s;
  static bool once = (set_smem(conv_fwd_dma_kernel<WGM, WGN, true>, T::SMEM), set_smem(conv_fwd_dma_kernel<WGM, WGN, false>, T::SMEM), true);
  (void)once;
  char pname[112];
  if (prof_detail())
    snprintf(pname, sizeof pname, "conv_fwd_dma_%dx%d[M%d,N%d,K%d,b%d,full%d,tail%dx%d]", T::BM, T::BN, p.M, p.Cout, p.K, nb,
             sc.full, sc.tail_tiles, sc.tail_s);
  else
    snprintf(pname, sizeof pname, "conv_fwd_dma_%dx%d", T::BM, T::BN);
  ProfScope prof(s, pname, 2.0 * p.M * p.Cout * p.K * nb);
  const int units = sc.full + sc.tail_tiles * sc.tail_s;
  if (split_on()) hipLaunchKernelGGL((conv_fwd_dma_kernel<WGM, WGN, true>), dim3(units), dim3(64 * T::NW), T::SMEM, hs(s), p, sc);
  else hipLaunchKernelGGL((conv_fwd_dma_kernel<WGM, WGN, false>), dim3(units), dim3(64 * T::NW), T::SMEM, hs(s), p, sc);
  check_launch("conv_fwd_dma");
  if (sc.tail_tiles > 0 && sc.tail_s > 1) {
    hipLaunchKernelGGL((conv_dma_reduce_kernel<T::BM, T::BN>), dim3(T::BM * T::BN / 4 / 256, sc.tail_tiles), dim3(256), 0, hs(s), p, sc);
    check_launch("conv_dma_reduce");
  }
}

static bool dma_on() {
  static const bool on = !(getenv("SWN_DMA") && atoi(getenv("SWN_DMA")) == 0);
  return on;
}

static int g_force_naive = 0;
void conv_force_naive(int on) { g_force_naive = on; }

// SWN_AMAX_FUSED=0: every launch takes the amax of its operands itself (A/B against the producer-side slots; read per launch)
static bool amax_fused_on() { return !(getenv("SWN_AMAX_FUSED") && atoi(getenv("SWN_AMAX_FUSED")) == 0); }
// ---- pre-cut ring kernel: schedule + launch ------------------------------------------------------------------------
static bool pc_on() {
  const bool on = !(getenv("SWN_PRECUT") && atoi(getenv("SWN_PRECUT")) == 0);      // read per launch (tests / A-B runs)
  return on;
}
// 2 (default): two fp16 planes per operand, three MFMAs per product; 1: the reduced-precision configuration (one fp16 plane per
// operand, one MFMA: bench.py --precision f16, never the headline).  Read once: the operands a model holds are cut for one form.
static int pc_planes() {
  static const int pl = [] {
    const int v = getenv("SWN_PC_PLANES") ? atoi(getenv("SWN_PC_PLANES")) : 2;
    return v == 1 ? 1 : 2;
  }();
  return pl;
}
int conv_precut_planes() { return pc_planes(); }
bool wino_pair_planes() {
  static const bool off = getenv("SWN_PAIR") && atoi(getenv("SWN_PAIR")) == 0;
  return !off && pc_planes() == 2 && dma_on() && split_on() && !g_force_naive && amax_fused_on();
}
// the last 2 KiB of a stream's scratch hold the partial maxima of the launch in flight (A operand) and of the operand a producer
// is cutting; the split-K slabs of the same launch stay below
constexpr size_t PC_WS_TAIL = 2048;
static float* ws_amax(Stream& s, int which) {
  if (!s.ws || s.ws_bytes < (1u << 20)) throw Error(1, "two-plane pre-cut kernels need the stream scratch");
  return reinterpret_cast<float*>(s.ws + s.ws_bytes - PC_WS_TAIL + (size_t)which * 1024);
}
static void amax_partials(Stream& s, const float* x, size_t rows, int C, size_t rs, int batch, size_t bs, float* out, int fold = 0,
                          float floor = 0.f) {
  if (C % 4 || rs % 4 || bs % 4 || ((uintptr_t)x & 15)) throw Error(1, "amax_partials: operand not 16-byte aligned");
  const int flat = rs == (size_t)C && (batch == 1 || bs == rows * (size_t)C);
  hipLaunchKernelGGL(amax_partials_kernel, dim3(256), dim3(1024), 0, hs(s), x, rows, C / 4, rs, batch, bs, flat, out, fold, floor);
  check_launch("amax_partials");
}
template <int WGM, int NB, int NSTG, int WGCU, int PL>
static void launch_fwd_pc(Stream& s, GemmP& p, int nb, const unsigned short* wpc, size_t wpc_bs, bool phases, const float* x_amax,
                          const int* x_pair_k = nullptr) {
  using T = PcTile<WGM, NB, NSTG, PL>;
  const int tiles_m = ceil_div(p.M, T::BM);
  p.tiles_n = ceil_div(p.Npad, T::BN);
  p.ntiles = tiles_m * p.tiles_n;
  constexpr int wg = WGCU;
  static_assert(wg * T::SMEM <= 160 * 1024, "tile does not fit a CU");
  const size_t ws_cap = s.ws_bytes - PC_WS_TAIL;
  const float* a_amax = nullptr;
  if (x_pair_k && PL != 2) throw Error(1, "conv_fwd: a pair-form operand needs the two-plane kernel");
  if (x_pair_k) {
    // (the producer scaled and cut the operand: nothing to take the amax of)
  } else if (x_amax && amax_fused_on()) {
    a_amax = x_amax;          // the producer of the operand left its amax (256 floats, maximum = amax) in a slot: no pass of our own
  } else {
    // |A|max over the whole input tensor of the launch (all images, all channels the gather reads; batched planes too)
    float* part = ws_amax(s, 0);
    amax_partials(s, p.x, (size_t)(p.M / (p.Ho * p.Wo)) * p.xH * p.xW, p.xC, (size_t)p.xcs, phases ? 1 : nb, p.x_bs, part);
    a_amax = part;
  }
  DmaSched sc = plan_dma(p.ntiles * nb, p.ntiles, p.K / T::BK, 256 * wg, (size_t)T::BM * T::BN * 4, ws_cap, nullptr,
                         wg * T::NW / 12.0);
  if (phases && !(getenv("SWN_PHASE_ZFAST") && atoi(getenv("SWN_PHASE_ZFAST")) == 0)) sc.zfast = nb;     // (A/B, read per launch)
  if (p.stat) {       // the statistics come out of the tile epilogue: every tile whole
    if (!(WGM == 4 && NB == 4) || nb != 1 || p.accumulate || p.act != ACT_NONE || (p.Ho * p.Wo) % T::BM)
      throw Error(1, "conv_fwd: stat_partial on a launch that cannot emit statistics (ask conv_fwd_stat_chunk first)");
    sc.full = p.ntiles * nb; sc.tail_tiles = 0; sc.tail_s = 1; sc.per_split = p.K / T::BK;
  }
  p.slab = reinterpret_cast<float*>(s.ws);
  p.splits = sc.tail_s;
  static bool once = (set_smem(conv_fwd_pc_kernel<WGM, NB, NSTG, WGCU, PL>, T::SMEM), true);
  (void)once;
  char pname[112];
  if (prof_detail())
    snprintf(pname, sizeof pname, "conv_fwd_pc_%dx%d%s[M%d,N%d,K%d,b%d,full%d,tail%dx%d]", T::BM, T::BN, x_pair_k ? "_ap" : "", p.M, p.Cout, p.K, nb, sc.full,
             sc.tail_tiles, sc.tail_s);
  else
    snprintf(pname, sizeof pname, "conv_fwd_pc_%dx%d", T::BM, T::BN);
  ProfScope prof(s, pname, 2.0 * p.M * p.Cout * p.K * nb);
  const int units = sc.full + sc.tail_tiles * sc.tail_s;
  if constexpr (PL == 2 && NSTG <= 3) {
    if (x_pair_k) {
      static bool once2 = (set_smem(conv_fwd_pc_kernel<WGM, NB, NSTG, WGCU, 2, true>, T::SMEM), true);
      (void)once2;
      hipLaunchKernelGGL((conv_fwd_pc_kernel<WGM, NB, NSTG, WGCU, 2, true>), dim3(units), dim3(64 * T::NW), T::SMEM, hs(s), p, sc, wpc, wpc_bs,
                         a_amax, x_pair_k);
    } else {
      hipLaunchKernelGGL((conv_fwd_pc_kernel<WGM, NB, NSTG, WGCU, PL>), dim3(units), dim3(64 * T::NW), T::SMEM, hs(s), p, sc, wpc, wpc_bs,
                         a_amax, x_pair_k);
    }
  } else {
    if (x_pair_k) throw Error(1, "conv_fwd: no pair-form instantiation of this tile configuration");
    hipLaunchKernelGGL((conv_fwd_pc_kernel<WGM, NB, NSTG, WGCU, PL>), dim3(units), dim3(64 * T::NW), T::SMEM, hs(s), p, sc, wpc, wpc_bs,
                       a_amax, x_pair_k);
  }
  check_launch("conv_fwd_pc");
  if (sc.tail_tiles > 0 && sc.tail_s > 1) {
    hipLaunchKernelGGL((conv_dma_reduce_kernel<T::BM, T::BN>), dim3(T::BM * T::BN / 4 / 256, sc.tail_tiles), dim3(256), 0, hs(s), p, sc);
    check_launch("conv_dma_reduce");
  }
}

// column tile of the pre-cut kernel by output width: 64 (256 x 64), 128 (128 x 128), or 192 for N in (128, 192] (the tail
// conv's input gradient into the 192-channel concat: one 128 x 192 tile instead of two 128-wide ones of which one is half empty)
static int pc_tile_for(int Npad) { return Npad <= 64 ? 64 : ((Npad > 128 && Npad <= 192) ? 192 : 128); }
// ops.h: which column tile a forward-type launch over an input with xC channels into Npad columns wants its weight operand
// pre-cut for (0 = the launch does not take the pre-cut ring kernel: no operand needs to be produced)
int conv_fwd_stat_chunk(int xC, int Npad, int HoWo, int nimg, int K) {
  if (pc_planes() != 2 || !pc_on() || conv_precut_tile(xC, Npad) != 128 || Npad <= 64 || HoWo % 128 || K % 16) return 0;
  // only where the launch runs every tile WHOLE anyway (C2 bs 32: 1024 / 2048 tiles; C3 bs 16: 512): the statistics come out of the
  // tile epilogue, and forcing a launch the planner would split along K onto whole tiles would trade its shorter fp32 accumulation
  // chains for one of K / 16 x 3 MFMAs per output -- measured on the MI355X at bs 2: the normalised output 4.3e-7 instead of 2.5e-7 from
  // float64, and the pinned gradients of the texture U-Net's deep levels 1e-4 instead of 2e-5 (tools/r06_stats_probe.py)
  const int tiles = (int)((size_t)nimg * HoWo / 128) * ceil_div(Npad, 128);
  const DmaSched sc = plan_dma(tiles, tiles, K / 16, 256 * 4, (size_t)128 * 128 * 4, (size_t)1 << 30, nullptr, 4 * 4 / 12.0);
  if (sc.tail_tiles > 0 && sc.tail_s > 1) return 0;
  return 128;
}
int conv_precut_tile(int xC, int Npad) {
  static const bool off = getenv("SWN_PRECUT") && atoi(getenv("SWN_PRECUT")) == 0;
  if (off || g_force_naive || !dma_on() || !split_on() || xC % 16 || Npad <= 32) return 0;
  return pc_tile_for(Npad);
}
size_t conv_precut_elems(int K, int Npad, int bn) {
  return (size_t)(K / 16) * ceil_div(Npad, bn) * 2 * pc_planes() * bn * 8 + PC_TRAILER;
}
const float* conv_precut_amax(Stream& s, const float* src, size_t rows, int C, int batch, size_t bs) {
  float* part = ws_amax(s, 1);
  amax_partials(s, src, rows, C, (size_t)C, batch, bs, part);
  return part;
}
void conv_precut(Stream& s, const float* w, int K, int Npad, int bn, int batch, size_t w_bs, uint16_t* out, const float** amax_io) {
  if (K % 16 || (bn != 64 && bn != 128 && bn != 192)) throw Error(1, "conv_precut: K must be a multiple of 16, tile 64, 128 or 192");
  const size_t total = (size_t)(K / 8) * ceil_div(Npad, bn) * bn;
  // two-plane form: one scale for all `batch` panels of the launch (they are cut from one weight tensor)
  const float* wamax = (amax_io && *amax_io) ? *amax_io : conv_precut_amax(s, w, (size_t)K, Npad, batch, w_bs);
  if (amax_io) *amax_io = wamax;
  hipLaunchKernelGGL(conv_precut_kernel, dim3((unsigned)((total + 255) / 256), batch), dim3(256), 0, hs(s), w, out, K, Npad, bn, w_bs,
                     conv_precut_elems(K, Npad, bn), wamax, pc_planes());
  check_launch("conv_precut");
}

void tensor_amax(Stream& s, const TView& x, float* slot, float floor) {
  amax_partials(s, x.p, x.pixels(), x.C, (size_t)x.cs, 1, 0, slot, 0, floor);
}
// ConvFwdArgs::y_amax on a launch whose kernel has no folding epilogue: a pass over the output view
static void fold_output_amax(Stream& s, const ConvFwdArgs& a) {
  amax_partials(s, a.y.p, a.y.pixels(), a.y.C, (size_t)a.y.cs, 1, 0, a.y_amax, 1);
}

// the LDS-DMA kernel addresses activations through 32-bit buffer offsets and marks padding with offsets >= 2^31
static bool dma_ok(const ConvFwdArgs& a, const GemmP& p) {
  if (!dma_on() || a.x.C % 16 || a.Npad <= 32) return false;
  const size_t xbytes = (size_t)a.x.N * a.x.H * a.x.W * a.x.cs * 4, wbytes = (size_t)p.K * a.Npad * 4;
  return xbytes < ((size_t)1 << 31) && wbytes < ((size_t)1 << 31);
}

static bool narrow_on() {
  const bool on = !(getenv("SWN_NARROW") && atoi(getenv("SWN_NARROW")) == 0);     // read per launch: tests toggle it
  return on;
}


// per-phase form of a tail4 launch (ops.h): the reference path and the fallback of the fused kernels
template <class Args>
static Args tail_phase_args(const Args& a, int ph, size_t woff) {
  Args c = a;
  c.tail4 = 0;
  c.g.KH = 2 + (ph >> 1); c.g.KW = 2 + (ph & 1); c.g.stride = 1; c.g.pad_t = 1; c.g.pad_l = 1;
  c.om.ymul = 2; c.om.xmul = 2; c.om.yoff = ph >> 1; c.om.xoff = ph & 1;
  (void)woff;
  return c;
}
static size_t tail_panel_off(int ph, int xC, int Npad) {
  const int pre = ph == 0 ? 0 : (ph == 1 ? 4 : (ph == 2 ? 10 : 16));
  return (size_t)pre * xC * Npad;
}
static void check_tail4(const Gather& g, const OutMap& om, int batch, int phases) {
  if (g.KH != 3 || g.KW != 3 || g.stride != 1 || g.pad_t != 1 || g.pad_l != 1 || g.ups || g.pad_mode != PAD_ZERO ||
      om.ymul != 2 || om.xmul != 2 || batch > 1 || phases)
    throw Error(1, "conv: bad tail4 launch");
}

void conv_fwd(Stream& s, const ConvFwdArgs& a) {
  if (a.tail4) {
    check_tail4(a.g, a.om, a.batch, a.phases);
    const bool fused = !(getenv("SWN_TAIL4") && atoi(getenv("SWN_TAIL4")) == 0);
    if (g_force_naive || !fused || a.x.C % 32 || a.Npad > 20 || a.accumulate) {
      for (int ph = 0; ph < 4; ++ph) {
        ConvFwdArgs c = tail_phase_args(a, ph, 0);
        c.w = a.w + tail_panel_off(ph, a.x.C, a.Npad);
        conv_fwd(s, c);
      }
      return;
    }
    OutMap om = a.om; om.yoff = om.xoff = 1;          // bounds check against the farthest phase
    GemmP p = make_params(a.x, a.g, a.y, om);
    p.w = a.w; p.Npad = a.Npad; p.bias = a.bias; p.act = a.act; p.Cout = a.Cout; p.tail4 = 1;
    p.tiles_n = 1; p.ntiles = ceil_div(p.M, TailTile::BM);
    static bool once = (set_smem(tail_fwd4_kernel<5>, TailTile::SMEM), true);
    (void)once;
    ProfScope prof(s, "tail_fwd4", 2.0 * p.M * p.Cout * 25 * a.x.C);
    hipLaunchKernelGGL(tail_fwd4_kernel<5>, dim3(p.ntiles), dim3(256), TailTile::SMEM, hs(s), p);
    check_launch("tail_fwd4");
    return;
  }
  if (g_force_naive) { conv_fwd_naive(s, a); if (a.y_amax) fold_output_amax(s, a); return; }
  GemmP p = make_params(a.x, a.g, a.y, a.om, a.phases, a.batch);
  p.w = a.w; p.Npad = a.Npad; p.bias = a.bias; p.act = a.act; p.accumulate = a.accumulate; p.Cout = a.Cout;
  p.y_amax = a.y_amax;
  p.stat = a.stat_partial;
  if (a.stat_partial && !(a.wpc && pc_planes() == 2 && pc_on() && a.Npad > 64 && (a.g.Ho * a.g.Wo) % 128 == 0 && a.wpc_bn == 128 && !a.phases && a.batch <= 1))
    throw Error(1, "conv_fwd: stat_partial on a launch outside the 128 x 128 pre-cut kernel");
  if (a.Npad % 4 || a.Cout > a.Npad) throw Error(1, "conv_fwd: bad Npad/Cout");
  if (a.accumulate && a.act != ACT_NONE) throw Error(1, "conv_fwd: accumulate with activation");
  const bool fast = (a.x.C % 32) == 0;
  constexpr int big = 1;
  p.x_bs = a.x_bs; p.w_bs = a.w_bs; p.y_bs = a.y_bs;
  const int nb = a.phases ? a.phases : std::max(a.batch, 1);
  constexpr int t192 = 1;
  // N in (128, 192] (the tail conv's input gradient into the 192-channel concat): a 128x192 tile instead
  // of two 128-wide column tiles of which the second is half empty
  if (dma_ok(a, p)) {
    // weight operand handed over pre-cut (conv_precut) for this launch's column tile: the round-3 kernel
    if (a.wpc && pc_on() && split_on() && a.wpc_bn == pc_tile_for(a.Npad) &&
        (size_t)(p.K / 16) * ceil_div(a.Npad, a.wpc_bn) * 12 * a.wpc_bn * 8 < ((size_t)1 << 31)) {
      const bool ph = a.phases != 0;
      if (pc_planes() == 1) {
        if (a.wpc_bn == 192) launch_fwd_pc<4, 6, 2, 2, 1>(s, p, nb, a.wpc, a.wpc_bs, ph, a.x_amax);
        else if (a.Npad > 64) launch_fwd_pc<4, 4, 2, 4, 1>(s, p, nb, a.wpc, a.wpc_bs, ph, a.x_amax);
        else launch_fwd_pc<8, 2, 3, 2, 1>(s, p, nb, a.wpc, a.wpc_bs, ph, a.x_amax);
        return;
      }
      if (pc_planes() == 2) {
        if (a.wpc_bn == 192) launch_fwd_pc<4, 6, 2, 2, 2>(s, p, nb, a.wpc, a.wpc_bs, ph, a.x_amax, a.x_pair_k);
        else if (a.Npad > 64) {
          // 128 x 128: two LDS stages, four workgroups per CU (3- and 4-stage rings measured the same within 0.4 %: rounds 3 / 4)
          launch_fwd_pc<4, 4, 2, 4, 2>(s, p, nb, a.wpc, a.wpc_bs, ph, a.x_amax, a.x_pair_k);
        }
        else launch_fwd_pc<8, 2, 3, 2, 2>(s, p, nb, a.wpc, a.wpc_bs, ph, a.x_amax, a.x_pair_k);
        return;
      }
      throw Error(1, "conv_fwd: unknown plane count");
    }
    if (!a.w) throw Error(1, "conv_fwd: the weight operand exists in pre-cut form only, but this launch cannot take the pre-cut "
                             "kernel (SWN_SPLIT / SWN_PRECUT / SWN_DMA must not change after a model is built)");
    if (a.x_pair_k) throw Error(1, "conv_fwd: pair-form operand on a launch outside the pre-cut ring kernel");
    p.y_amax = nullptr;
    if (a.Npad > 64) {
      // 128 x 128 (4 waves, 3 workgroups / CU) unless the 128 x 256 tile (8 waves, 2 / CU: 512 slots instead of 768)
      // quantises the launch better: the resblock input gradient (M 800, N 1024 x 36 planes) is 2016 tiles = 2.6
      // rounds of 768 but 1008 = 1.97 rounds of 512
      const int wide = getenv("SWN_DMA_WIDE") ? atoi(getenv("SWN_DMA_WIDE")) : 1;   // 0 never, 1 by cost, 2 always (tests); per launch
      double c22 = 0, c24 = 0;
      // (with the split main loop the 8-wave tile spills at its 128-VGPR budget and measures slower: f32-MFMA form only)
      if (wide && (wide == 2 || !split_on()) && a.Npad % 256 == 0) { plan_fwd_dma<2, 2>(p, nb, s.ws_bytes, &c22); plan_fwd_dma<2, 4>(p, nb, s.ws_bytes, &c24); }
      if (c24 > 0 && (wide == 2 || c24 < 0.95 * c22)) launch_fwd_dma<2, 4>(s, p, nb);
      else launch_fwd_dma<2, 2>(s, p, nb);
    }
    else launch_fwd_dma<4, 1>(s, p, nb);                 // 256 x 64
    if (a.y_amax) fold_output_amax(s, a);
    return;
  }
  if (!a.w) throw Error(1, "conv_fwd: pre-cut-only weight operand on a launch outside the ring kernel's shapes");
  if (a.x_pair_k) throw Error(1, "conv_fwd: pair-form operand on a launch outside the pre-cut ring kernel");
  p.y_amax = nullptr;
  // every register-staged route: the amax of the output, if asked for, by a pass behind the launch
  if (t192 && a.Npad > 128 && a.Npad <= 192) launch_fwd<2, 3, 2, 2>(s, p, fast, nb);
  else if (a.Npad > 64 && big && fast && p.M >= 2048) launch_fwd<2, 2, 4, 2>(s, p, fast, nb);
  else if (a.Npad > 64) launch_fwd<2, 2, 2, 2>(s, p, fast, nb);
  else if (a.Npad > 32) launch_fwd<2, 1, 2, 2>(s, p, fast, nb);   // (a 2-wave 128x64 tile with 64x64 wave tiles measured 4 % slower)
  else if (!narrow_on()) launch_fwd<1, 1, 4, 1>(s, p, fast, nb);
  else if (a.phases == 4 && a.Npad <= 20 && p.KH == 2 && p.KW == 2 && p.stride == 1 && !p.ups && p.pad_mode == PAD_ZERO && a.x.C % 16 == 0 &&
           !(getenv("SWN_PHASE4") && atoi(getenv("SWN_PHASE4")) == 0)) {
    // (SWN_PHASE4=0, read per launch: one narrow launch per phase, as before round 5)
    if (a.Npad <= 4) launch_fwd_phase4<1>(s, p);
    else if (a.Npad <= 8) launch_fwd_phase4<2>(s, p);
    else if (a.Npad <= 16) launch_fwd_phase4<4>(s, p);
    else launch_fwd_phase4<5>(s, p);
  }
  else if (a.Npad <= 4) launch_fwd_narrow<1>(s, p, fast, nb);
  else if (a.Npad <= 8) launch_fwd_narrow<2>(s, p, fast, nb);
  else if (a.Npad <= 16) launch_fwd_narrow<4>(s, p, fast, nb);
  else if (a.Npad <= 20) launch_fwd_narrow<5>(s, p, fast, nb);
  else if (a.Npad <= 24) launch_fwd_narrow<6>(s, p, fast, nb);
  else launch_fwd_narrow<8>(s, p, fast, nb);
  if (a.y_amax) fold_output_amax(s, a);
}

template <int MT, int NT, int WGM, int WGN, int NG = 0, bool ROWU = false>
static void launch_wgrad(Stream& s, GemmP& p, int batch) {
  if constexpr (!ROWU) {
    // a 32-pixel stage lies inside one image and starts at a row start (or inside one row)
    const bool rowu = p.M % 32 == 0 && (p.Wo % 32 == 0 || (32 % p.Wo == 0 && (p.Ho * p.Wo) % 32 == 0));
    if (rowu) { launch_wgrad<MT, NT, WGM, WGN, NG, true>(s, p, batch); return; }
  }
  using T = Tile<MT, NT, WGM, WGN>;
  const int tiles_k = ceil_div(p.K, T::BM);
  p.tiles_n = ceil_div(p.Npad, T::BN);
  p.ntiles = tiles_k * p.tiles_n;
  const int nmb = ceil_div(p.M, 32);
  const int slots = 256 * (T::SMEM_WG > 80 * 1024 ? 1 : (T::SMEM_WG >= 64 * 1024 ? 2 : (T::SMEM_WG >= 48 * 1024 ? 3 : 4)));
  const int splits = choose_splits(p.ntiles * batch, nmb, slots, 8, (size_t)p.K * p.Npad * 4 * batch, s.ws_bytes);
  p.per_split = ceil_div(nmb, splits);
  p.splits = ceil_div(nmb, p.per_split);
  p.slab = reinterpret_cast<float*>(s.ws);
  p.slab_bs = (size_t)p.K * p.Npad * p.splits;
  static bool once = (set_smem(conv_wgrad_kernel<MT, NT, WGM, WGN, NG, ROWU>, T::SMEM_WG), true);
  (void)once;
  char pname[96];
  const int bn = NG > 0 ? 4 * NG : T::BN;
  if (prof_detail())
    snprintf(pname, sizeof pname, "conv_wgrad_%dx%d[M%d,N%d,K%d,s%d]", T::BM, bn, p.M, p.Cout, p.K, p.splits);
  else
    snprintf(pname, sizeof pname, "conv_wgrad_%dx%d", T::BM, bn);
  ProfScope prof(s, pname, 2.0 * p.M * p.Cout * p.K * batch);
  hipLaunchKernelGGL((conv_wgrad_kernel<MT, NT, WGM, WGN, NG, ROWU>), dim3(p.ntiles, p.splits, batch), dim3(64 * WGM * WGN),
                     T::SMEM_WG, hs(s), p);
  check_launch("conv_wgrad");
  if (p.splits > 1) {
    const size_t n = (size_t)p.K * p.Npad;
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n / 4 + 15) / 16), batch), dim3(256), 0, hs(s), p.slab,
                       const_cast<float*>(p.w), n, p.splits, p.slab_bs, p.w_bs);
    check_launch("slab_sum");
  }
}


// 2 (default): the weight-gradient ring kernel on two fp16 planes per operand, both scaled by powers of two from their amax (three
// MFMAs per product; tools/ring_lab.hip variants 18 / 19: 157-167 -> 266-268 fp32-equivalent TFLOP/s at 4.5e-7); 1: the
// reduced-precision configuration (one plane, bench.py --precision f16).  Read per launch (tests).
static int wgrad_planes() {
  const int v = getenv("SWN_WGRAD_PLANES") ? atoi(getenv("SWN_WGRAD_PLANES")) : 2;
  return v == 1 ? 1 : 2;
}
template <int WGM, int WGN>
static void launch_wgrad_dma(Stream& s, GemmP& p, int nb, const ConvWgradArgs& a) {
  using T = DmaWgTile<WGM, WGN>;
  const int tiles_k = ceil_div(p.K, T::BMK);
  p.tiles_n = ceil_div(p.Npad, T::BN);
  p.ntiles = tiles_k * p.tiles_n;
  const int nmb = p.M / T::PX;
  const int wg_per_cu = std::min(160 * 1024 / T::SMEM, 12 / T::NW);
  const int wpl = wgrad_planes();
  const bool two = split_on() && wpl <= 2 && s.ws && s.ws_bytes >= (1u << 20);        // fp16 planes: scaled operands
  const float *xa = nullptr, *ya = nullptr;
  if (two) {
    // both operands are activations: their amax over the whole tensors the gather / the dY rows come from -- left in a slot by
    // whoever produced the tensor (ConvWgradArgs::x_amax / dy_amax), else taken here
    const int nbb = a.phases ? 1 : nb;
    const size_t nimg = (size_t)(p.M / (p.Ho * p.Wo));
    const bool fused = amax_fused_on();
    if (a.x_pair_k) xa = reinterpret_cast<const float*>(a.x_pair_k);
    else if (a.x_amax && fused) xa = a.x_amax;
    else { float* px = ws_amax(s, 0); amax_partials(s, a.x.p, nimg * a.x.H * a.x.W, a.x.C, (size_t)a.x.cs, nbb, a.x_bs, px); xa = px; }
    if (a.dy_pair_k) ya = reinterpret_cast<const float*>(a.dy_pair_k);
    else if (a.dy_amax && fused) ya = a.dy_amax;
    else { float* py = ws_amax(s, 1); amax_partials(s, a.dy.p, nimg * a.dy.H * a.dy.W, a.dy.C, (size_t)a.dy.cs, nbb, a.dy_bs, py); ya = py; }
  }
  const DmaSched sc = plan_dma(p.ntiles * nb, p.ntiles, nmb, 256 * wg_per_cu, (size_t)T::BMK * T::BN * 4, two ? s.ws_bytes - PC_WS_TAIL : s.ws_bytes);
  p.slab = reinterpret_cast<float*>(s.ws);
  p.splits = sc.tail_s;
  static bool once = (set_smem(conv_wgrad_dma_kernel<WGM, WGN, 2>, T::SMEM), set_smem(conv_wgrad_dma_kernel<WGM, WGN, 0>, T::SMEM),
                      set_smem(conv_wgrad_dma_kernel<WGM, WGN, 3>, T::SMEM), true);
  (void)once;
  // plain [M][C] operands (batched Winograd planes): the loader without im2col arithmetic.  SWN_WGRAD_PLANE=0: generic (A/B runs)
  const bool plane = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && !p.ups && !p.phases && p.Ho == 1 &&
                     p.Wo % T::PX == 0 && p.xH == 1 && p.yH == 1 && p.xW == p.Wo && p.yW == p.Wo && p.ymul == 1 && p.xmul == 1 &&
                     p.yoff == 0 && p.xoff == 0 && p.M == p.Wo;
  const bool plane_name = two && wpl == 2 && plane;
  const int pair_name = (a.x_pair_k ? 1 : 0) | (a.dy_pair_k ? 2 : 0);
  char pname[128];
  if (prof_detail())
    snprintf(pname, sizeof pname, "conv_wgrad_dma_%dx%d%s[M%d,N%d,K%d,b%d,full%d,tail%dx%d]", T::BMK, T::BN, two ? (wpl == 1 ? "_h1" : (plane_name ? (pair_name == 3 ? "_h2pp" : (pair_name ? "_h2p1" : "_h2p")) : "_h2")) : "", p.M, p.Cout, p.K, nb,
             sc.full, sc.tail_tiles, sc.tail_s);
  else
    snprintf(pname, sizeof pname, "conv_wgrad_dma_%dx%d", T::BMK, T::BN);
  ProfScope prof(s, pname, 2.0 * p.M * p.Cout * p.K * nb);
  const int units = sc.full + sc.tail_tiles * sc.tail_s;
  const int pairm = (a.x_pair_k ? 1 : 0) | (a.dy_pair_k ? 2 : 0);
  if (pairm && !(two && wpl == 2 && plane)) throw Error(1, "conv_wgrad: pair-form operands need the two-plane plane-form kernel");
  if (two && wpl == 2 && plane) {
    static bool once2 = (set_smem(conv_wgrad_dma_kernel<WGM, WGN, 2, true, 0>, T::SMEM), set_smem(conv_wgrad_dma_kernel<WGM, WGN, 2, true, 1>, T::SMEM),
                         set_smem(conv_wgrad_dma_kernel<WGM, WGN, 2, true, 2>, T::SMEM), set_smem(conv_wgrad_dma_kernel<WGM, WGN, 2, true, 3>, T::SMEM), true);
    (void)once2;
    const dim3 g(units), b(64 * T::NW);
    if (pairm == 3) hipLaunchKernelGGL((conv_wgrad_dma_kernel<WGM, WGN, 2, true, 3>), g, b, T::SMEM, hs(s), p, sc, xa, ya);
    else if (pairm == 2) hipLaunchKernelGGL((conv_wgrad_dma_kernel<WGM, WGN, 2, true, 2>), g, b, T::SMEM, hs(s), p, sc, xa, ya);
    else if (pairm == 1) hipLaunchKernelGGL((conv_wgrad_dma_kernel<WGM, WGN, 2, true, 1>), g, b, T::SMEM, hs(s), p, sc, xa, ya);
    else hipLaunchKernelGGL((conv_wgrad_dma_kernel<WGM, WGN, 2, true, 0>), g, b, T::SMEM, hs(s), p, sc, xa, ya);
  }
  else if (two && wpl == 1) hipLaunchKernelGGL((conv_wgrad_dma_kernel<WGM, WGN, 3>), dim3(units), dim3(64 * T::NW), T::SMEM, hs(s), p, sc, xa, ya);
  else if (two) hipLaunchKernelGGL((conv_wgrad_dma_kernel<WGM, WGN, 2>), dim3(units), dim3(64 * T::NW), T::SMEM, hs(s), p, sc, xa, ya);
  else hipLaunchKernelGGL((conv_wgrad_dma_kernel<WGM, WGN, 0>), dim3(units), dim3(64 * T::NW), T::SMEM, hs(s), p, sc, xa, ya);
  check_launch("conv_wgrad_dma");
  if (sc.tail_tiles > 0 && sc.tail_s > 1) {
    hipLaunchKernelGGL((wgrad_dma_reduce_kernel<T::BMK, T::BN>), dim3(T::BMK * T::BN / 4 / 256, sc.tail_tiles), dim3(256), 0, hs(s), p, sc);
    check_launch("wgrad_dma_reduce");
  }
}
// ops.h: would a batched plane launch with these dimensions take the kernels that read pair-form operands?  (The engine decides the
// storage form of a layer's Winograd planes with these when the layer is built; the launchers re-check and throw on a mismatch.)
bool conv_fwd_takes_pairs(int xC, int Npad) { return wino_pair_planes() && conv_precut_tile(xC, Npad) != 0; }
bool conv_wgrad_takes_pairs(size_t T, int K, int Npad) {
  constexpr bool plane_off = false;
  if (plane_off || !wino_pair_planes() || wgrad_planes() != 2 || Npad <= 32 || K % 4 || Npad % 4 || T % 16 || T * (size_t)std::max(K, Npad) * 4 >= ((size_t)1 << 31))
    return false;
  const int bmk = Npad > 64 ? 128 : 256;
  const double fillf = (double)K / (double)(ceil_div(K, bmk) * bmk);
  return fillf > 0.8 || (bmk == 128 && fillf >= 0.75);
}
// stage geometry the LDS-DMA wgrad kernel needs: 16 consecutive pixels inside one image at fixed offsets from the first
static bool wgrad_dma_ok(const ConvWgradArgs& a, const GemmP& p) {
  if (!dma_on() || a.Npad <= 32 || a.dy.C % 4 || a.x.C % 4) return false;
  const bool geom = (p.Wo % 16 == 0) || (16 % p.Wo == 0 && (p.Ho * p.Wo) % 16 == 0);
  const size_t xbytes = (size_t)a.x.N * a.x.H * a.x.W * a.x.cs * 4, ybytes = (size_t)a.dy.N * a.dy.H * a.dy.W * a.dy.cs * 4;
  // the k-tile is 128 rows (N > 64) or 256 rows (N <= 64): shapes that would leave a quarter or more of the tile rows
  // empty (the K = 64 / 320 / 384 first-layer weight gradients) stay on the 128-row register-staged kernel
  const int bmk = a.Npad > 64 ? 128 : 256;
  // (128-row tiles from 0.75: K = 192, the tail conv's Winograd-domain weight gradient, is 1.5 tiles of a long reduction)
  const double fillf = (double)p.K / (double)(ceil_div(p.K, bmk) * bmk);
  const bool fill = fillf > 0.8 || (bmk == 128 && fillf >= 0.75);
  return geom && fill && p.M % 16 == 0 && xbytes < ((size_t)1 << 31) && ybytes < ((size_t)1 << 31);
}

void conv_wgrad(Stream& s, const ConvWgradArgs& a) {
  if (a.tail4) {
    check_tail4(a.g, a.om, a.batch, a.phases);
    const bool fused = !(getenv("SWN_TAIL4") && atoi(getenv("SWN_TAIL4")) == 0);
    if (g_force_naive || !fused || a.x.C != 192 || a.Npad > 20 || a.g.Wo % 16 || a.g.Wo != a.x.W || a.g.Ho != a.x.H) {
      for (int ph = 0; ph < 4; ++ph) {
        ConvWgradArgs c = tail_phase_args(a, ph, 0);
        c.dw = a.dw + tail_panel_off(ph, a.x.C, a.Npad);
        conv_wgrad(s, c);
      }
      return;
    }
    OutMap om = a.om; om.yoff = om.xoff = 1;
    GemmP p = make_params(a.x, a.g, a.dy, om);
    p.w = a.dw; p.Npad = a.Npad; p.Cout = a.Cout; p.tail4 = 1;
    constexpr int NW = 3, KC = 64 * NW, PX = 16;
    constexpr int smem = (2 * PX * KC + 2 * 4 * PX * 20 + 32) * 4;
    const int nmb = p.M / PX;
    // 9 tap blocks of unequal length (1-4 active phases): many short splits keep the chip balanced
    const size_t slab_bytes = (size_t)25 * KC * p.Npad * 4;
    int splits = std::max(1, std::min(nmb / 32, 456));
    while (splits > 1 && slab_bytes * splits > s.ws_bytes) --splits;
    p.per_split = ceil_div(nmb, splits);
    p.splits = ceil_div(nmb, p.per_split);
    p.slab = reinterpret_cast<float*>(s.ws);
    ProfScope prof(s, "tail_wgrad4", 2.0 * p.M * p.Cout * 25 * a.x.C);
    hipLaunchKernelGGL((tail_wgrad4_kernel<5, NW, PX>), dim3(9, p.splits), dim3(64 * NW), smem, hs(s), p);
    check_launch("tail_wgrad4");
    if (p.splits > 1) {
      const size_t n = (size_t)25 * KC * p.Npad;
      hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n / 4 + 15) / 16), 1), dim3(256), 0, hs(s), p.slab,
                         const_cast<float*>(p.w), n, p.splits, (size_t)0, (size_t)0);
      check_launch("slab_sum");
    }
    return;
  }
  if (g_force_naive) { conv_wgrad_naive(s, a); return; }
  GemmP p = make_params(a.x, a.g, a.dy, a.om, a.phases, a.batch);
  p.w = a.dw; p.Npad = a.Npad; p.Cout = a.Cout;
  if (a.Npad % 4 || a.Cout > a.Npad || a.dy.C % 4) throw Error(1, "conv_wgrad: bad Npad/Cout");
  p.x_bs = a.x_bs; p.y_bs = a.dy_bs; p.w_bs = a.dw_bs;
  const int nb = a.phases ? a.phases : std::max(a.batch, 1);
  constexpr int big = 1;
  if (wgrad_dma_ok(a, p)) {
    if (a.Npad > 64) launch_wgrad_dma<2, 2>(s, p, nb, a);    // 128 k-rows x 128 columns
    else launch_wgrad_dma<4, 1>(s, p, nb, a);                // 256 x 64
    return;
  }
  if (a.x_pair_k || a.dy_pair_k) throw Error(1, "conv_wgrad: pair-form operands on a launch outside the ring kernel");
  // 8-wave 256x128 tile: +3 % on the single-GEMM layers, -7 % on the batched Winograd planes (measured)
  if (a.Npad > 64 && big && p.K >= 512 && nb == 1) launch_wgrad<2, 2, 4, 2>(s, p, nb);
  else if (a.Npad > 64) launch_wgrad<2, 2, 2, 2>(s, p, nb);
  else if (a.Npad > 32) launch_wgrad<2, 1, 2, 2>(s, p, nb);
  // narrow variant only where it measured faster (N <= 8: PatchGAN's 1-channel head); at N = 19 both
  // forms are bound by the im2col load path (2 N FLOP per loaded float), not by the matrix pipe
  else if (narrow_on() && a.Npad <= 4) launch_wgrad<2, 1, 4, 1, 1>(s, p, nb);
  else if (narrow_on() && a.Npad <= 8) launch_wgrad<2, 1, 4, 1, 2>(s, p, nb);
  else launch_wgrad<1, 1, 4, 1>(s, p, nb);
}

static void host_phase(GemmP& q, int ph) {
  if (!q.phases) return;
  q.pad_t -= ph >> 1; q.pad_l -= ph & 1; q.yoff = ph >> 1; q.xoff = ph & 1;
  q.phases = 0;
}

void conv_fwd_naive(Stream& s, const ConvFwdArgs& a) {
  GemmP p = make_params(a.x, a.g, a.y, a.om, a.phases, a.batch);
  p.w = a.w; p.Npad = a.Npad; p.bias = a.bias; p.act = a.act; p.accumulate = a.accumulate; p.Cout = a.Cout;
  const size_t total = (size_t)p.M * p.Cout;
  for (int b = 0; b < (a.phases ? a.phases : std::max(a.batch, 1)); ++b) {
    GemmP q = p;
    host_phase(q, b);
    q.x += (size_t)b * a.x_bs; q.w += (size_t)b * a.w_bs; q.y += (size_t)b * a.y_bs;
    hipLaunchKernelGGL(conv_fwd_naive_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, hs(s), q);
  }
  check_launch("conv_fwd_naive");
}

void conv_wgrad_naive(Stream& s, const ConvWgradArgs& a) {
  GemmP p = make_params(a.x, a.g, a.dy, a.om, a.phases, a.batch);
  p.w = a.dw; p.Npad = a.Npad; p.Cout = a.Cout;
  const size_t total = (size_t)p.K * p.Npad;
  for (int b = 0; b < (a.phases ? a.phases : std::max(a.batch, 1)); ++b) {
    GemmP q = p;
    host_phase(q, b);
    q.x += (size_t)b * a.x_bs; q.y += (size_t)b * a.dy_bs; q.w += (size_t)b * a.dw_bs;
    hipLaunchKernelGGL(conv_wgrad_naive_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, hs(s), q);
  }
  check_launch("conv_wgrad_naive");
}

}  // namespace swn
