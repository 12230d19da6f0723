// swapnet_amd -- InstanceNorm2d (eps 1e-5, biased variance, no affine) + activation +
// dropout (+ residual add), forward and backward, on NHWC views.  HBM-bound kernels:
// every access is a 16-byte channel-contiguous load/store, statistics are accumulated in
// fp64 (no E[x^2]-E[x]^2 cancellation), partial sums are combined in a fixed order so the
// result is run-to-run deterministic.
// Reference: modules/layers.py:12-44,126-144 (UNetDown / UNetUp / ResidualBlock),
// modules/__init__.py:66-69 (InstanceNorm2d(affine=False, track_running_stats=False)).
#include "hip_util.h"

namespace swn {

namespace {

constexpr float IN_EPS = 1e-5f;

__device__ __forceinline__ float act_grad_from_in(float v, int act) {
  switch (act) {
    case ACT_LRELU: return v > 0.f ? 1.f : 0.2f;
    case ACT_RELU: return v > 0.f ? 1.f : 0.f;
    case ACT_TANH: { const float th = tanhf(v); return 1.f - th * th; }
    default: return 1.f;
  }
}

struct NAp {
  const float* x; int xcs;
  float* y; int ycs;            // fwd: y ; bwd: dx
  const float* dy; int dycs;    // bwd only
  const float* res; int rescs;  // fwd residual
  float* stats;                 // [N][C][2]
  double* colsum;               // bwd, optional: [N][C] column sums of dx
  double* partial;              // [N][nchunk][C][2]
  double* sums;                 // bwd: [N][C][2] = mean(dxh), mean(dxh*xh)
  int N, HW, C, nchunk, chunk;
  int norm, act;
  float drop_p; uint64_t seed;
  float* amax_out;              // optional amax slot of what the apply kernels write (fwd: y, bwd: dx)
  const uint64_t* seed_base;    // captured step: the step seed lives in device memory; seed = *seed_base * 0x9E3779B1 + salt
  uint64_t salt;
  int pair_xcd;                 // in_fused_group
};
__device__ __forceinline__ uint64_t na_seed(const NAp& p) { return p.seed_base ? *p.seed_base * 0x9E3779B1ull + p.salt : p.seed; }

// partial sums over a pixel chunk: mode 0 -> (sum x, sum x^2); mode 1 -> (sum dxh, sum dxh*xh)
template <int MODE>
__global__ __launch_bounds__(256) void in_partial_kernel(NAp p) {
  const uint64_t drop_seed_v = na_seed(p);
  __shared__ double red[256 * 8];
  const int C4 = p.C >> 2;
  const int rows = 256 / C4;
  const int t = threadIdx.x, tx = t % C4, ty = t / C4;
  const int n = blockIdx.y, ch = blockIdx.x;
  const int c0 = ch * p.chunk, c1 = min(p.HW, c0 + p.chunk);
  double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  if (ty < rows) {
    float mean[4] = {0, 0, 0, 0}, rstd[4] = {1, 1, 1, 1};
    if (MODE == 1 && p.norm) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        mean[j] = p.stats[((size_t)n * p.C + tx * 4 + j) * 2];
        rstd[j] = p.stats[((size_t)n * p.C + tx * 4 + j) * 2 + 1];
      }
    }
    for (int pix = c0 + ty; pix < c1; pix += rows) {
      const size_t e = (size_t)n * p.HW + pix;
      const float4 xv = *reinterpret_cast<const float4*>(p.x + e * p.xcs + tx * 4);
      const float xa[4] = {xv.x, xv.y, xv.z, xv.w};
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[j] += xa[j]; ss[j] += (double)xa[j] * xa[j]; }
      } else {
        const float4 gv = *reinterpret_cast<const float4*>(p.dy + e * p.dycs + tx * 4);
        const float ga[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (xa[j] - mean[j]) * rstd[j];                 // the forward's value (selects the activation branch)
          float g = ga[j] * act_grad_from_in(xh, p.act);
          if (p.drop_p > 0.f) g *= drop_scale(drop_seed_v, e * p.C + tx * 4 + j, p.drop_p);
          s[j] += g; ss[j] += (double)g * (((double)xa[j] - (double)mean[j]) * (double)rstd[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[t * 8 + j] = s[j]; red[t * 8 + 4 + j] = ss[j]; }
  __syncthreads();
  if (ty == 0) {
    for (int r = 1; r < rows; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) red[t * 8 + j] += red[(r * C4 + tx) * 8 + j];
    double* o = p.partial + (((size_t)n * p.nchunk + ch) * p.C + tx * 4) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j * 2] = red[t * 8 + j]; o[j * 2 + 1] = red[t * 8 + 4 + j]; }
  }
}

template <int MODE>
__global__ void in_finalize_kernel(NAp p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;     // over N*C
  if (i >= p.N * p.C) return;
  const int n = i / p.C, c = i - n * p.C;
  double a = 0, b = 0;
  for (int ch = 0; ch < p.nchunk; ++ch) {
    const double* o = p.partial + (((size_t)n * p.nchunk + ch) * p.C + c) * 2;
    a += o[0]; b += o[1];
  }
  if (MODE == 0) {
    const double mean = a / p.HW;
    double var = b / p.HW - mean * mean;
    if (var < 0) var = 0;
    p.stats[(size_t)i * 2] = (float)mean;
    p.stats[(size_t)i * 2 + 1] = (float)(1.0 / sqrt(var + (double)IN_EPS));
  } else {
    p.sums[(size_t)i * 2] = a / p.HW;
    p.sums[(size_t)i * 2 + 1] = b / p.HW;
  }
}

__global__ __launch_bounds__(256) void norm_act_apply_kernel(NAp p) {
  const uint64_t drop_seed_v = na_seed(p);
  const int C4 = p.C >> 2;
  const size_t total = (size_t)p.N * p.HW * C4;
  float am = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i / C4;
    const int c = (int)(i - e * C4) * 4;
    const int n = (int)(e / p.HW);
    const float4 xv = *reinterpret_cast<const float4*>(p.x + e * p.xcs + c);
    float v[4] = {xv.x, xv.y, xv.z, xv.w};
    if (p.norm) {
      const float4 s0 = *reinterpret_cast<const float4*>(p.stats + ((size_t)n * p.C + c) * 2);
      const float4 s1 = *reinterpret_cast<const float4*>(p.stats + ((size_t)n * p.C + c) * 2 + 4);
      v[0] = (v[0] - s0.x) * s0.y; v[1] = (v[1] - s0.z) * s0.w;
      v[2] = (v[2] - s1.x) * s1.y; v[3] = (v[3] - s1.z) * s1.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = act_apply(v[j], p.act);
      if (p.drop_p > 0.f) v[j] *= drop_scale(drop_seed_v, e * p.C + c + j, p.drop_p);
    }
    if (p.res) {
      const float4 r = *reinterpret_cast<const float4*>(p.res + e * p.rescs + c);
      v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
    }
    const float4 o4 = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p.y + e * p.ycs + c) = o4;
    am = fmaxf(am, f4amax(o4));
  }
  amax_fold(am, p.amax_out);
}

__global__ __launch_bounds__(256) void norm_act_bwd_apply_kernel(NAp p) {
  const uint64_t drop_seed_v = na_seed(p);
  const int C4 = p.C >> 2;
  const size_t total = (size_t)p.N * p.HW * C4;
  float am = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i / C4;
    const int c = (int)(i - e * C4) * 4;
    const int n = (int)(e / p.HW);
    const float4 xv = *reinterpret_cast<const float4*>(p.x + e * p.xcs + c);
    const float4 gv = *reinterpret_cast<const float4*>(p.dy + e * p.dycs + c);
    const float xa[4] = {xv.x, xv.y, xv.z, xv.w};
    const float ga[4] = {gv.x, gv.y, gv.z, gv.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float mean = 0.f, rstd = 1.f;
      double m1 = 0.0, m2 = 0.0;
      if (p.norm) {
        mean = p.stats[((size_t)n * p.C + c + j) * 2];
        rstd = p.stats[((size_t)n * p.C + c + j) * 2 + 1];
        m1 = p.sums[((size_t)n * p.C + c + j) * 2];
        m2 = p.sums[((size_t)n * p.C + c + j) * 2 + 1];
      }
      const float xh = (xa[j] - mean) * rstd;
      float g = ga[j] * act_grad_from_in(xh, p.act);
      if (p.drop_p > 0.f) g *= drop_scale(drop_seed_v, e * p.C + c + j, p.drop_p);
      // g - mean(g) - xh * mean(g xh) cancels heavily (the deep layers lose 3 digits here): the combination is done in
      // double -- the kernel is HBM-bound (20 B per element), the handful of fp64 operations is free
      o[j] = p.norm ? (float)((double)rstd * ((double)g - m1 - (((double)xa[j] - (double)mean) * (double)rstd) * m2)) : g;
    }
    const float4 o4 = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(p.y + e * p.ycs + c) = o4;
    am = fmaxf(am, f4amax(o4));
  }
  amax_fold(am, p.amax_out);
}

// ---- second-order step through act(IN(x)) -- ops.h norm_act_bwd2 ------------------------------------------------
struct NA2p {
  const float *u, *gy, *x; int ucs, gycs, xcs;
  float *uy, *ax; int uycs, axcs;
  const float* stats;
  double* partial;      // [N][nchunk][C][5]
  double* sums;         // [N][C][5] = <u>, <gm>, <u xh>, <gm xh>, <u gm>
  int N, HW, C, nchunk, chunk, act;
};
__global__ __launch_bounds__(256) void in2_partial_kernel(NA2p p) {
  __shared__ double red[256 * 5];
  const int n = blockIdx.y, ch = blockIdx.x;
  const int c0 = ch * p.chunk, c1 = min(p.HW, c0 + p.chunk);
  // thread = one channel of one pixel row: consecutive threads walk consecutive channels (coalesced scalar loads)
  const int cpb = min(p.C, 256), rows = 256 / cpb;
  const int tx = threadIdx.x % cpb, ty = threadIdx.x / cpb;
  for (int cb = 0; cb < p.C; cb += cpb) {
    const int c = cb + tx;
    double s[5] = {0, 0, 0, 0, 0};
    if (ty < rows && c < p.C) {
      const float mean = p.stats[((size_t)n * p.C + c) * 2], rstd = p.stats[((size_t)n * p.C + c) * 2 + 1];
      for (int pix = c0 + ty; pix < c1; pix += rows) {
        const size_t e = (size_t)n * p.HW + pix;
        const float xv = p.x[e * p.xcs + c];
        const float xhf = (xv - mean) * rstd;
        const double xh = ((double)xv - (double)mean) * (double)rstd;
        const double uu = p.u[e * p.ucs + c];
        const double gm = (double)(p.gy[e * p.gycs + c] * act_grad_from_in(xhf, p.act));
        s[0] += uu; s[1] += gm; s[2] += uu * xh; s[3] += gm * xh; s[4] += uu * gm;
      }
    }
    for (int j = 0; j < 5; ++j) red[threadIdx.x * 5 + j] = s[j];
    __syncthreads();
    if (ty == 0 && c < p.C) {
      for (int r = 1; r < rows; ++r)
        for (int j = 0; j < 5; ++j) red[threadIdx.x * 5 + j] += red[(r * cpb + tx) * 5 + j];
      double* o = p.partial + (((size_t)n * p.nchunk + ch) * p.C + c) * 5;
      for (int j = 0; j < 5; ++j) o[j] = red[threadIdx.x * 5 + j];
    }
    __syncthreads();
  }
}
__global__ void in2_finalize_kernel(NA2p p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.N * p.C) return;
  const int n = i / p.C, c = i - n * p.C;
  double a[5] = {0, 0, 0, 0, 0};
  for (int ch = 0; ch < p.nchunk; ++ch) {
    const double* o = p.partial + (((size_t)n * p.nchunk + ch) * p.C + c) * 5;
    for (int j = 0; j < 5; ++j) a[j] += o[j];
  }
  for (int j = 0; j < 5; ++j) p.sums[(size_t)i * 5 + j] = a[j] / p.HW;
}
__global__ __launch_bounds__(256) void in2_apply_kernel(NA2p p) {
  const size_t total = (size_t)p.N * p.HW * p.C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i / p.C;
    const int c = (int)(i - e * p.C);
    const int n = (int)(e / p.HW);
    const float mean = p.stats[((size_t)n * p.C + c) * 2], rstdf = p.stats[((size_t)n * p.C + c) * 2 + 1];
    const double* m = p.sums + ((size_t)n * p.C + c) * 5;
    const double mu = m[0], mg = m[1], muh = m[2], mgh = m[3], mug = m[4];
    const float xv = p.x[e * p.xcs + c];
    const float ad = act_grad_from_in((xv - mean) * rstdf, p.act);
    const double rstd = rstdf, xh = ((double)xv - (double)mean) * rstd;
    const double uu = p.u[e * p.ucs + c], gm = (double)(p.gy[e * p.gycs + c] * ad);
    const double ju = uu - mu - xh * muh;
    p.uy[e * p.uycs + c] = (float)((double)ad * rstd * ju);
    p.ax[e * p.axcs + c] = (float)(-rstd * rstd * (xh * (mug - mu * mg - muh * mgh) + mgh * ju + muh * (gm - mg - xh * mgh)));
  }
}


// ---------------------------------------------------------------------------------------
// Register-resident InstanceNorm for maps of up to 1024 pixels (every IN of the two networks below 64x64: the resblocks,
// the deep encoder / decoder levels, PatchGAN's 32x32 and 31x31 maps).  One block owns one image x CG channels and keeps
// the whole H*W slab in registers: statistics and apply are ONE pass over HBM (read x once, write y once; backward: read x
// and dy once, write dx once) and one launch, where the chunked three-kernel path reads x twice (thrice backward) and
// launches three kernels per layer.  Same arithmetic: fp64 sums, a fixed reduction tree (deterministic), mean / rstd rounded
// to fp32 exactly like in_finalize_kernel.
// ---------------------------------------------------------------------------------------
template <int ROWS, int NV>
__device__ __forceinline__ void block_tree_sum(double* red, double (&v)[NV], int tx, int ty, int C4) {
  // red[(ty * C4 + tx) * NV + j]; result broadcast from row 0
#pragma unroll
  for (int j = 0; j < NV; ++j) red[(ty * C4 + tx) * NV + j] = v[j];
  __syncthreads();
#pragma unroll
  for (int stride = ROWS / 2; stride >= 1; stride >>= 1) {
    if (ty < stride) {
#pragma unroll
      for (int j = 0; j < NV; ++j) red[(ty * C4 + tx) * NV + j] += red[((ty + stride) * C4 + tx) * NV + j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = red[tx * NV + j];
}

// Channel group of a block.  A 16-channel slab reads 64 bytes per pixel: half a 128-byte line, whose other half belongs to the
// neighbouring group.  Workgroups go round-robin over the 8 XCDs in launch order, so groups g and g + 1 land on different L2s and the
// line is fetched twice (FETCH_SIZE: 2.0 x the operands of the <16, 16> launches, 1.0 x for the contiguous apply kernels).  Within
// every 16 consecutive blocks the order is permuted so that line partners are 8 launches apart -- the same XCD, microseconds apart.
// SWN_IN_PAIR_XCD=0 (A/B, read per launch) keeps the plain order.
template <int CG>
__device__ __forceinline__ int in_fused_group(int pair_xcd) {
  const int x = blockIdx.x;
  if (CG != 16 || !pair_xcd) return x;
  const int r = x & 15;
  return (x & ~15) + ((r & 7) << 1) + (r >> 3);
}

template <int CG, int NP>
__global__ __launch_bounds__(256) void in_fused_fwd_kernel(NAp p) {
  const uint64_t drop_seed_v = na_seed(p);
  constexpr int C4 = CG / 4, ROWS = 256 / C4;
  __shared__ double red[256 * 8];
  const int t = threadIdx.x, tx = t % C4, ty = t / C4;
  const int n = blockIdx.y, c = in_fused_group<CG>(p.pair_xcd) * CG + tx * 4;
  float4 v[NP];
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // all loads first, unconditionally (rows past the map re-read its last pixel and are masked below): inside `if (pix < HW)`
  // every load sits in its own basic block and the NP round trips to HBM serialise
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pix = min(ty + i * ROWS, p.HW - 1);
    v[i] = *reinterpret_cast<const float4*>(p.x + ((size_t)n * p.HW + pix) * p.xcs + c);
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const float k = ty + i * ROWS < p.HW ? 1.f : 0.f;
    const float a0 = v[i].x * k, a1 = v[i].y * k, a2 = v[i].z * k, a3 = v[i].w * k;
    s[0] += a0; s[1] += a1; s[2] += a2; s[3] += a3;
    s[4] += (double)a0 * a0; s[5] += (double)a1 * a1; s[6] += (double)a2 * a2; s[7] += (double)a3 * a3;
  }
  block_tree_sum<ROWS, 8>(red, s, tx, ty, C4);
  float mean[4], rstd[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double m = s[j] / p.HW;
    double var = s[4 + j] / p.HW - m * m;
    if (var < 0) var = 0;
    mean[j] = (float)m; rstd[j] = (float)(1.0 / sqrt(var + (double)IN_EPS));
  }
  if (ty == 0) {
    float* st = p.stats + ((size_t)n * p.C + c) * 2;
    *reinterpret_cast<float4*>(st) = make_float4(mean[0], rstd[0], mean[1], rstd[1]);
    *reinterpret_cast<float4*>(st + 4) = make_float4(mean[2], rstd[2], mean[3], rstd[3]);
  }
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pix = ty + i * ROWS;
    if (pix >= p.HW) continue;
    const size_t e = (size_t)n * p.HW + pix;
    float o[4] = {(v[i].x - mean[0]) * rstd[0], (v[i].y - mean[1]) * rstd[1], (v[i].z - mean[2]) * rstd[2], (v[i].w - mean[3]) * rstd[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = act_apply(o[j], p.act);
      if (p.drop_p > 0.f) o[j] *= drop_scale(drop_seed_v, e * p.C + c + j, p.drop_p);
    }
    if (p.res) {
      const float4 r = *reinterpret_cast<const float4*>(p.res + e * p.rescs + c);
      o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
    }
    const float4 o4 = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(p.y + e * p.ycs + c) = o4;
    am = fmaxf(am, f4amax(o4));
  }
  amax_fold(am, p.amax_out);
}

template <int CG, int NP>
__global__ __launch_bounds__(256) void in_fused_bwd_kernel(NAp p) {
  const uint64_t drop_seed_v = na_seed(p);
  constexpr int C4 = CG / 4, ROWS = 256 / C4;
  __shared__ double red[256 * 8];
  const int t = threadIdx.x, tx = t % C4, ty = t / C4;
  const int n = blockIdx.y, c = in_fused_group<CG>(p.pair_xcd) * CG + tx * 4;
  float mean[4], rstd[4];
  {
    const float* st = p.stats + ((size_t)n * p.C + c) * 2;
    const float4 s0 = *reinterpret_cast<const float4*>(st), s1 = *reinterpret_cast<const float4*>(st + 4);
    mean[0] = s0.x; rstd[0] = s0.y; mean[1] = s0.z; rstd[1] = s0.w; mean[2] = s1.x; rstd[2] = s1.y; mean[3] = s1.z; rstd[3] = s1.w;
  }
  float4 xv[NP], gv[NP];
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < NP; ++i) {          // all loads first, unconditionally (see the forward kernel)
    const size_t e = (size_t)n * p.HW + min(ty + i * ROWS, p.HW - 1);
    xv[i] = *reinterpret_cast<const float4*>(p.x + e * p.xcs + c);
    gv[i] = *reinterpret_cast<const float4*>(p.dy + e * p.dycs + c);
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pix = ty + i * ROWS;
    if (pix < p.HW) {
      const size_t e = (size_t)n * p.HW + pix;
      const float4 d = gv[i];
      const float xa[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
      float g[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (xa[j] - mean[j]) * rstd[j];                  // the forward's value (selects the activation branch)
        g[j] *= act_grad_from_in(xh, p.act);
        if (p.drop_p > 0.f) g[j] *= drop_scale(drop_seed_v, e * p.C + c + j, p.drop_p);
        s[j] += g[j];
        s[4 + j] += (double)g[j] * (((double)xa[j] - (double)mean[j]) * (double)rstd[j]);
      }
      gv[i] = make_float4(g[0], g[1], g[2], g[3]);
    }
  }
  block_tree_sum<ROWS, 8>(red, s, tx, ty, C4);
  double m1[4], m2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { m1[j] = s[j] / p.HW; m2[j] = s[4 + j] / p.HW; }
  double cs[4] = {0, 0, 0, 0};
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pix = ty + i * ROWS;
    if (pix >= p.HW) continue;
    const size_t e = (size_t)n * p.HW + pix;
    const float xa[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
    const float g[4] = {gv[i].x, gv[i].y, gv[i].z, gv[i].w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = (float)((double)rstd[j] * ((double)g[j] - m1[j] - (((double)xa[j] - (double)mean[j]) * (double)rstd[j]) * m2[j]));
      cs[j] += o[j];
    }
    const float4 o4 = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(p.y + e * p.ycs + c) = o4;
    am = fmaxf(am, f4amax(o4));
  }
  amax_fold(am, p.amax_out);
  if (p.colsum) {                      // (block-uniform) bias gradient of the producing conv: sum of dx over this image's pixels
    block_tree_sum<ROWS, 4>(red, cs, tx, ty, C4);
    if (ty == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) p.colsum[(size_t)n * p.C + c + j] = cs[j];
    }
  }
}

static bool fused_in_on() {
  static const bool on = true;        // (the round-3 A/B switch SWN_FUSED_IN is gone)
  return on;
}

struct EWp {
  const float* a; int acs;
  const float* b; int bcs;
  float* o; int ocs;
  size_t pixels; int C;
  int act; int accumulate; float alpha; float shift;
  float* amax_out;              // optional amax slot of what is written
};

// MODE 0: o = act(a);  1: o (+)= a * act'(.) expressed through b = the activation OUTPUT;  2: o (+)= alpha*a
template <int MODE>
__global__ __launch_bounds__(256) void ew_kernel(EWp p) {
  const int C4 = p.C >> 2;
  const size_t total = p.pixels * C4;
  float am = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i / C4;
    const int c = (int)(i - e * C4) * 4;
    const float4 av = *reinterpret_cast<const float4*>(p.a + e * p.acs + c);
    float v[4] = {av.x, av.y, av.z, av.w};
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = act_apply(v[j], p.act);
    } else if (MODE == 1) {
      const float4 bv = *reinterpret_cast<const float4*>(p.b + e * p.bcs + c);
      v[0] *= act_grad_from_out(bv.x, p.act); v[1] *= act_grad_from_out(bv.y, p.act);
      v[2] *= act_grad_from_out(bv.z, p.act); v[3] *= act_grad_from_out(bv.w, p.act);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = v[j] * p.alpha + p.shift;
    }
    float* dst = p.o + e * p.ocs + c;
    if (MODE != 0 && p.accumulate) {
      const float4 d = *reinterpret_cast<const float4*>(dst);
      v[0] += d.x; v[1] += d.y; v[2] += d.z; v[3] += d.w;
    }
    const float4 o4 = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(dst) = o4;
    am = fmaxf(am, f4amax(o4));
  }
  amax_fold(am, p.amax_out);
}

// column sums of dy (bias gradient): stage 1 per pixel-chunk partials (fp64), stage 2 sum
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* dy, int cs, size_t pixels, int C,
                                                             size_t chunk, double* partial) {
  __shared__ double red[256 * 4];
  const int C4 = C >> 2, rows = 256 / C4;
  const int t = threadIdx.x, tx = t % C4, ty = t / C4;
  const size_t p0 = (size_t)blockIdx.x * chunk, p1 = min(pixels, p0 + chunk);
  double s[4] = {0, 0, 0, 0};
  if (ty < rows)
    for (size_t e = p0 + ty; e < p1; e += rows) {
      const float4 v = *reinterpret_cast<const float4*>(dy + e * cs + tx * 4);
      s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[t * 4 + j] = s[j];
  __syncthreads();
  if (ty == 0) {
    for (int r = 1; r < rows; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[t * 4 + j] += red[(r * C4 + tx) * 4 + j];
#pragma unroll
    for (int j = 0; j < 4; ++j) partial[(size_t)blockIdx.x * C + tx * 4 + j] = red[t * 4 + j];
  }
}
// 16 channels x 16 lanes per block: each lane sums every 16th chunk partial, then a fixed-order
// LDS sum over the 16 lanes (deterministic)
__global__ __launch_bounds__(256) void colsum_final_kernel(const double* partial, int nchunk, int C, float* out) {
  __shared__ double red[256];
  const int cl = threadIdx.x & 15, ln = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double a = 0;
  if (c < C)
    for (int i = ln; i < nchunk; i += 16) a += partial[(size_t)i * C + c];
  red[threadIdx.x] = a;
  __syncthreads();
  if (ln == 0 && c < C) {
    double t = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red[j * 16 + cl];
    out[c] = (float)t;
  }
}

__global__ __launch_bounds__(256) void reflect_fold_kernel(const float* src, int scs, float* dst, int dcs, int N,
                                                           int H, int W, int C, int accumulate) {
  const int C4 = C >> 2;
  const size_t total = (size_t)N * H * W * C4;
  const int Hp = H + 2, Wp = W + 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i / C4;
    const int c = (int)(i - e * C4) * 4;
    const int n = (int)(e / ((size_t)H * W));
    const int rem = (int)(e - (size_t)n * H * W);
    const int y = rem / W, x = rem - y * W;
    // padded coordinates u with reflect(u-1) == y : u = y+1, plus u=0 when y==1, u=H+1 when y==H-2
    // (H,W >= 4 so the two border rows never fold onto the same row)
    int ys[2], xs[2], ny = 1, nx = 1;
    ys[0] = y + 1; xs[0] = x + 1;
    if (y == 1) ys[ny++] = 0;
    if (y == H - 2) ys[ny++] = H + 1;
    if (x == 1) xs[nx++] = 0;
    if (x == W - 2) xs[nx++] = W + 1;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) {
        const float4 v = *reinterpret_cast<const float4*>(src + ((size_t)(n * Hp + ys[a]) * Wp + xs[b]) * scs + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    float* d = dst + e * dcs + c;
    if (accumulate) {
      const float4 o = *reinterpret_cast<const float4*>(d);
      acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    *reinterpret_cast<float4*>(d) = acc;
  }
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(int N, int HW, int C, float p, uint64_t seed, float* out) {
  const size_t total = (size_t)N * C * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int pix = (int)(i % HW); const size_t q = i / HW;
    const int c = (int)(q % C); const size_t n = q / C;
    out[i] = drop_scale(seed, (n * HW + pix) * C + c, p);
  }
}

inline unsigned ew_grid(size_t total) {
  size_t b = (total + 255) / 256;
  return (unsigned)std::min<size_t>(std::max<size_t>(b, 1), 256 * 16);
}

void check_view4(const TView& v, const char* what) {
  if (v.C % 4 || v.cs % 4 || ((uintptr_t)v.p & 15)) throw Error(1, std::string(what) + ": view not 16-byte tileable");
}

void plan_chunks(int HW, int N, int C, int& nchunk, int& chunk) {
  const int rows = std::max(1, 256 / (C / 4));
  int want = std::max(1, 1024 / std::max(N, 1));                   // ~1024 blocks in flight
  int maxchunks = std::max(1, HW / std::max(rows * 4, 1));          // >= 4 pixels per thread row
  nchunk = std::min(std::min(want, maxchunks), 256);
  chunk = ceil_div(HW, nchunk);
  nchunk = ceil_div(HW, chunk);
}

}  // namespace

void norm_act_fwd(Stream& s, const NormActArgs& a) {
  check_view4(a.x, "norm_act_fwd x"); check_view4(a.y, "norm_act_fwd y");
  if (a.x.C > 1024) throw Error(1, "norm_act: C > 1024 unsupported");
  NAp p{};
  p.x = a.x.p; p.xcs = a.x.cs; p.y = a.y.p; p.ycs = a.y.cs;
  p.res = a.residual ? a.residual->p : nullptr; p.rescs = a.residual ? a.residual->cs : 0;
  p.stats = a.stats; p.N = a.x.N; p.HW = a.x.H * a.x.W; p.C = a.x.C;
  p.norm = a.norm; p.act = a.act; p.drop_p = a.drop_p; p.seed = a.seed;
  p.amax_out = a.amax_out; p.seed_base = a.seed_base; p.salt = a.salt;
  if (a.norm && !a.stats) throw Error(1, "norm_act_fwd: stats buffer required");
  p.pair_xcd = (p.C % 256 == 0 && !(getenv("SWN_IN_PAIR_XCD") && atoi(getenv("SWN_IN_PAIR_XCD")) == 0)) ? 1 : 0;
  if (a.norm && !a.partial_in && p.HW <= 1024 && p.C % 32 == 0 && fused_in_on()) {
    const dim3 grid(p.C / 32, p.N);
    // (above 512 pixels: 16-channel slabs of 64 KB, so several blocks share a CU and one block's load phase runs under
    // another's store phase -- a 128 KB slab per block leaves one block per CU and the two phases serialise chip-wide)
    if (p.HW <= 64) hipLaunchKernelGGL((in_fused_fwd_kernel<32, 2>), grid, dim3(256), 0, hs(s), p);
    else if (p.HW <= 256) hipLaunchKernelGGL((in_fused_fwd_kernel<32, 8>), grid, dim3(256), 0, hs(s), p);
    else if (p.HW <= 512) hipLaunchKernelGGL((in_fused_fwd_kernel<32, 16>), grid, dim3(256), 0, hs(s), p);
    else hipLaunchKernelGGL((in_fused_fwd_kernel<16, 16>), dim3(p.C / 16, p.N), dim3(256), 0, hs(s), p);
    check_launch("norm_act_fwd (fused)");
    return;
  }
  if (a.norm && route_on()) route_note(a.partial_in ? "norm_act[statistics from the conv epilogue]" : "norm_act[statistics pass]");
  if (a.norm && a.partial_in) {
    // Conv + InstanceNorm fusion: the producing conv's epilogue left the partial sums (ops.h ConvFwdArgs::stat_partial) -- no
    // statistics pass over x, finalize + apply only
    if (a.partial_chunks <= 0 || p.HW % a.partial_chunks) throw Error(1, "norm_act_fwd: bad partial_chunks");
    p.nchunk = a.partial_chunks; p.chunk = p.HW / a.partial_chunks;
    p.partial = const_cast<double*>(a.partial_in);
    hipLaunchKernelGGL(in_finalize_kernel<0>, dim3(ceil_div(p.N * p.C, 256)), dim3(256), 0, hs(s), p);
  } else if (a.norm) {
    plan_chunks(p.HW, p.N, p.C, p.nchunk, p.chunk);
    p.partial = reinterpret_cast<double*>(s.ws);
    if ((size_t)p.N * p.nchunk * p.C * 16 > s.ws_bytes) throw Error(1, "norm_act: workspace too small");
    hipLaunchKernelGGL(in_partial_kernel<0>, dim3(p.nchunk, p.N), dim3(256), 0, hs(s), p);
    hipLaunchKernelGGL(in_finalize_kernel<0>, dim3(ceil_div(p.N * p.C, 256)), dim3(256), 0, hs(s), p);
  }
  hipLaunchKernelGGL(norm_act_apply_kernel, dim3(ew_grid((size_t)p.N * p.HW * (p.C / 4))), dim3(256), 0, hs(s), p);
  check_launch("norm_act_fwd");
}

void norm_act_bwd(Stream& s, const NormActBwdArgs& a) {
  check_view4(a.x, "norm_act_bwd x"); check_view4(a.dy, "norm_act_bwd dy"); check_view4(a.dx, "norm_act_bwd dx");
  NAp p{};
  p.x = a.x.p; p.xcs = a.x.cs; p.dy = a.dy.p; p.dycs = a.dy.cs; p.y = a.dx.p; p.ycs = a.dx.cs;
  p.stats = const_cast<float*>(a.stats); p.N = a.x.N; p.HW = a.x.H * a.x.W; p.C = a.x.C;
  p.norm = a.norm; p.act = a.act; p.drop_p = a.drop_p; p.seed = a.seed;
  p.colsum = a.colsum;
  p.amax_out = a.amax_out; p.seed_base = a.seed_base; p.salt = a.salt;
  if (a.colsum && !(a.norm && norm_act_bwd_emits_colsum(p.HW, p.C))) throw Error(1, "norm_act_bwd: colsum requested on the chunked path");
  p.pair_xcd = (p.C % 256 == 0 && !(getenv("SWN_IN_PAIR_XCD") && atoi(getenv("SWN_IN_PAIR_XCD")) == 0)) ? 1 : 0;
  if (a.norm && p.HW <= 1024 && p.C % 32 == 0 && fused_in_on()) {
    if (p.HW <= 64) hipLaunchKernelGGL((in_fused_bwd_kernel<32, 2>), dim3(p.C / 32, p.N), dim3(256), 0, hs(s), p);
    else if (p.HW <= 256) hipLaunchKernelGGL((in_fused_bwd_kernel<32, 8>), dim3(p.C / 32, p.N), dim3(256), 0, hs(s), p);
    else if (p.HW <= 512) hipLaunchKernelGGL((in_fused_bwd_kernel<32, 16>), dim3(p.C / 32, p.N), dim3(256), 0, hs(s), p);
    else hipLaunchKernelGGL((in_fused_bwd_kernel<16, 16>), dim3(p.C / 16, p.N), dim3(256), 0, hs(s), p);
    check_launch("norm_act_bwd (fused)");
    return;
  }
  if (a.norm) {
    plan_chunks(p.HW, p.N, p.C, p.nchunk, p.chunk);
    p.partial = reinterpret_cast<double*>(s.ws);
    const size_t pbytes = (size_t)p.N * p.nchunk * p.C * 16;
    p.sums = reinterpret_cast<double*>(s.ws + round_up((int)pbytes, 256));
    if (pbytes + 256 + (size_t)p.N * p.C * 16 > s.ws_bytes) throw Error(1, "norm_act: workspace too small");
    hipLaunchKernelGGL(in_partial_kernel<1>, dim3(p.nchunk, p.N), dim3(256), 0, hs(s), p);
    hipLaunchKernelGGL(in_finalize_kernel<1>, dim3(ceil_div(p.N * p.C, 256)), dim3(256), 0, hs(s), p);
  }
  hipLaunchKernelGGL(norm_act_bwd_apply_kernel, dim3(ew_grid((size_t)p.N * p.HW * (p.C / 4))), dim3(256), 0, hs(s), p);
  check_launch("norm_act_bwd");
}

bool norm_act_bwd_emits_colsum(int HW, int C) {
  static const bool on = true;
  return on && HW <= 1024 && C % 32 == 0;
}
void bias_grad_from_colsums(Stream& s, const double* partial, int N, int C, float* db) {
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(C, 16)), dim3(256), 0, hs(s), partial, N, C, db);
  check_launch("bias_grad_from_colsums");
}
void dropout_mask(Stream& s, int N, int H, int W, int C, float p, uint64_t seed, float* out_nchw) {
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(ew_grid((size_t)N * H * W * C)), dim3(256), 0, hs(s), N, H * W, C, p, seed,
                     out_nchw);
  check_launch("dropout_mask");
}

void norm_act_bwd2(Stream& s, const NormActBwd2Args& a) {
  NA2p p{};
  p.u = a.u.p; p.ucs = a.u.cs; p.gy = a.gy.p; p.gycs = a.gy.cs; p.x = a.x.p; p.xcs = a.x.cs;
  p.uy = a.uy.p; p.uycs = a.uy.cs; p.ax = a.ax.p; p.axcs = a.ax.cs; p.stats = a.stats;
  p.N = a.x.N; p.HW = a.x.H * a.x.W; p.C = a.x.C; p.act = a.act;
  if (!a.stats) throw Error(1, "norm_act_bwd2: stats required");
  plan_chunks(p.HW, p.N, std::max(p.C, 4), p.nchunk, p.chunk);
  p.partial = reinterpret_cast<double*>(s.ws);
  const size_t pbytes = (size_t)p.N * p.nchunk * p.C * 5 * 8;
  p.sums = reinterpret_cast<double*>(s.ws + (pbytes + 255) / 256 * 256);
  if ((pbytes + 255) / 256 * 256 + (size_t)p.N * p.C * 5 * 8 > s.ws_bytes) throw Error(1, "norm_act_bwd2: workspace too small");
  hipLaunchKernelGGL(in2_partial_kernel, dim3(p.nchunk, p.N), dim3(256), 0, hs(s), p);
  hipLaunchKernelGGL(in2_finalize_kernel, dim3(ceil_div(p.N * p.C, 256)), dim3(256), 0, hs(s), p);
  hipLaunchKernelGGL(in2_apply_kernel, dim3(ew_grid((size_t)p.N * p.HW * p.C)), dim3(256), 0, hs(s), p);
  check_launch("norm_act_bwd2");
}

static EWp ew_params(const TView& a, const TView* b, const TView& o) {
  check_view4(a, "elementwise a"); check_view4(o, "elementwise out");
  if (b) check_view4(*b, "elementwise b");
  if (a.pixels() != o.pixels() || a.C != o.C) throw Error(1, "elementwise: shape mismatch");
  EWp p{};
  p.a = a.p; p.acs = a.cs; p.b = b ? b->p : nullptr; p.bcs = b ? b->cs : 0;
  p.o = o.p; p.ocs = o.cs; p.pixels = a.pixels(); p.C = a.C; p.alpha = 1.f;
  return p;
}

void act_fwd(Stream& s, const TView& x, const TView& y, int act, float* amax_out) {
  EWp p = ew_params(x, nullptr, y); p.act = act; p.amax_out = amax_out;
  hipLaunchKernelGGL(ew_kernel<0>, dim3(ew_grid(p.pixels * (p.C / 4))), dim3(256), 0, hs(s), p);
  check_launch("act_fwd");
}
void act_bwd(Stream& s, const TView& dy, const TView& y, const TView& dx, int act, int accumulate, float* amax_out) {
  EWp p = ew_params(dy, &y, dx); p.act = act; p.accumulate = accumulate; p.amax_out = amax_out;
  hipLaunchKernelGGL(ew_kernel<1>, dim3(ew_grid(p.pixels * (p.C / 4))), dim3(256), 0, hs(s), p);
  check_launch("act_bwd");
}
void axpy(Stream& s, const TView& src, const TView& dst, float alpha, int accumulate, float shift, float* amax_out) {
  EWp p = ew_params(src, nullptr, dst); p.alpha = alpha; p.accumulate = accumulate; p.shift = shift; p.amax_out = amax_out;
  hipLaunchKernelGGL(ew_kernel<2>, dim3(ew_grid(p.pixels * (p.C / 4))), dim3(256), 0, hs(s), p);
  check_launch("axpy");
}

void bias_grad(Stream& s, const TView& dy, float* db) {
  check_view4(dy, "bias_grad dy");
  const size_t pixels = dy.pixels();
  const int rows = std::max(1, 256 / (dy.C / 4));
  int nchunk = (int)std::min<size_t>(512, std::max<size_t>(1, pixels / (rows * 4)));
  const size_t chunk = (pixels + nchunk - 1) / nchunk;
  nchunk = (int)((pixels + chunk - 1) / chunk);
  double* partial = reinterpret_cast<double*>(s.ws);
  if ((size_t)nchunk * dy.C * 8 > s.ws_bytes) throw Error(1, "bias_grad: workspace too small");
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nchunk), dim3(256), 0, hs(s), dy.p, dy.cs, pixels, dy.C, chunk, partial);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(dy.C, 16)), dim3(256), 0, hs(s), partial, nchunk, dy.C, db);
  check_launch("bias_grad");
}

void reflect_fold(Stream& s, const TView& dxpad, const TView& dx, int accumulate) {
  check_view4(dxpad, "reflect_fold src"); check_view4(dx, "reflect_fold dst");
  if (dxpad.H != dx.H + 2 || dxpad.W != dx.W + 2 || dxpad.C != dx.C || dx.H < 4 || dx.W < 4)
    throw Error(1, "reflect_fold: shape mismatch");
  hipLaunchKernelGGL(reflect_fold_kernel, dim3(ew_grid(dx.pixels() * (dx.C / 4))), dim3(256), 0, hs(s), dxpad.p,
                     dxpad.cs, dx.p, dx.cs, dx.N, dx.H, dx.W, dx.C, accumulate);
  check_launch("reflect_fold");
}

}  // namespace swn
