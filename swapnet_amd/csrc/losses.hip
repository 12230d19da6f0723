// swapnet_amd -- loss reductions fused with their gradients (one pass over the data
// produces the scalar and dL/dx).  HBM-bound; fp64 block partials combined in fixed order.
// Reference: modules/loss.py:12-130 (GANLoss), models/warp_model.py:147-150 (CE on tanh
// outputs vs argmax(target)), models/texture_model.py:168-170 (L1),
// modules/losses/perceptual.py:6-10,49-79 (content / style).
#include "hip_util.h"

namespace swn {
namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over a 256-thread block; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0;
  if (threadIdx.x == 0) r = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return r;
}

// one 64-lane block: lane l sums partial[l], partial[l+64], ...; lane 0 then adds the 64 lane sums in
// index order (fixed order -> deterministic)
__global__ __launch_bounds__(64) void finalize_kernel(const double* partial, int n, double mul, float* out) {
  __shared__ double sh[64];
  double a = 0;
  for (int i = threadIdx.x; i < n; i += 64) a += partial[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int i = 0; i < 64; ++i) t += sh[i];
    out[0] = (float)(t * mul);
  }
}

inline int loss_grid(size_t work) { return (int)std::min<size_t>(std::max<size_t>((work + 255) / 256, 1), 1024); }

// ---- GAN losses on a 1-channel prediction map (channel 0 of a C-padded view) -----------
// MODE 0 BCE-with-logits, 1 LSGAN (MSE), 2 WGAN (sign * mean)
template <int MODE>
__global__ __launch_bounds__(256) void gan_loss_kernel(const float* pred, int pcs, size_t numel, float label,
                                                       float gscale, float* dpred, int dcs, double* partial, const float* label_dev) {
  __shared__ double sh[4];
  if (label_dev) label = *label_dev;          // the label of a captured step lives in device memory (ops.h bce_logits_loss)
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (size_t)gridDim.x * 256) {
    const float x = pred[i * pcs];
    float l, g;
    if (MODE == 0) {
      // max(x,0) - x*t + log1p(exp(-|x|))   (numerically stable BCEWithLogits)
      l = fmaxf(x, 0.f) - x * label + log1pf(expf(-fabsf(x)));
      g = 1.f / (1.f + expf(-x)) - label;
    } else if (MODE == 1) {
      const float d = x - label;
      l = d * d; g = 2.f * d;
    } else {
      l = label * x; g = label;        // label carries the sign (+1 fake, -1 real)
    }
    acc += l;
    if (dpred) dpred[i * dcs] = g * gscale;
  }
  const double r = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// ---- cross entropy vs argmax(target) ----------------------------------------------------
// one thread per pixel; logits / target / gradient rows are moved as float4 (the views are
// 16-byte aligned with C padded to a multiple of 4), C <= 32
__global__ __launch_bounds__(256) void ce_kernel(const float* logits, int lcs, const float* target, int tcs, int C,
                                                 size_t pixels, float gscale, float* dl, int dcs, int accumulate,
                                                 double* partial) {
  __shared__ double sh[4];
  double acc = 0;
  const int C4 = (C + 3) >> 2;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < pixels; e += (size_t)gridDim.x * 256) {
    float l[32], tg[32];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < C4) {
        const float4 a = *reinterpret_cast<const float4*>(logits + e * lcs + 4 * j);
        const float4 b = *reinterpret_cast<const float4*>(target + e * tcs + 4 * j);
        l[4 * j] = a.x; l[4 * j + 1] = a.y; l[4 * j + 2] = a.z; l[4 * j + 3] = a.w;
        tg[4 * j] = b.x; tg[4 * j + 1] = b.y; tg[4 * j + 2] = b.z; tg[4 * j + 3] = b.w;
      }
    int label = 0; float tmax = tg[0], lmax = l[0];
#pragma unroll
    for (int c = 1; c < 32; ++c)
      if (c < C) {
        if (tg[c] > tmax) { tmax = tg[c]; label = c; }     // first maximal index (torch.argmax)
        lmax = fmaxf(lmax, l[c]);
      }
    float se = 0.f, ll = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c)
      if (c < C) {
        const float d = l[c] - lmax;
        if (c == label) ll = d;
        l[c] = expf(d); se += l[c];
      }
    // -log softmax[label] = log(se) - (logit[label] - lmax)
    acc += (double)(logf(se) - ll);
    if (dl) {
      const float inv = gscale / se;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < C4) {
          float* dp = dl + e * dcs + 4 * j;
          float g[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int c = 4 * j + i;
            g[i] = c < C ? l[c] * inv - (c == label ? gscale : 0.f) : 0.f;
          }
          if (accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(dp);
            g[0] += o.x; g[1] += o.y; g[2] += o.z; g[3] += o.w;
          }
          *reinterpret_cast<float4*>(dp) = make_float4(g[0], g[1], g[2], g[3]);
        }
    }
  }
  const double r = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// ---- L1 ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l1_kernel(const float* a, int acs, const float* b, int bcs, int C, size_t pixels,
                                                 float gscale, float* da, int dcs, int accumulate, double* partial) {
  __shared__ double sh[4];
  double acc = 0;
  const size_t total = pixels * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t e = i / C; const int c = (int)(i - e * C);
    const float d = a[e * acs + c] - b[e * bcs + c];
    acc += fabsf(d);
    if (da) {
      float g = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * gscale;
      float* dp = da + e * dcs + c;
      if (accumulate) g += *dp;
      *dp = g;
    }
  }
  const double r = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// ---- content: MSE of channel-L2-normalised features, one wave per pixel -------------------
// f/(|f|+1e-8) vs t/(|t|+1e-8); C <= 512 (2 float4 per lane)
__global__ __launch_bounds__(256) void normed_mse_kernel(const float* f, int fcs, const float* t, int tcs, int C,
                                                         size_t pixels, float gscale, float* df, int dcs,
                                                         int accumulate, double* partial) {
  __shared__ double sh[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int C4 = C >> 2;
  double acc = 0;
  for (size_t e = (size_t)blockIdx.x * 4 + wv; e < pixels; e += (size_t)gridDim.x * 4) {
    float4 fv[2], tv[2];
    float sf = 0.f, st = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int c4 = lane + 64 * r;
      fv[r] = make_float4(0, 0, 0, 0); tv[r] = make_float4(0, 0, 0, 0);
      if (c4 < C4) {
        fv[r] = *reinterpret_cast<const float4*>(f + e * fcs + c4 * 4);
        tv[r] = *reinterpret_cast<const float4*>(t + e * tcs + c4 * 4);
      }
      sf += fv[r].x * fv[r].x + fv[r].y * fv[r].y + fv[r].z * fv[r].z + fv[r].w * fv[r].w;
      st += tv[r].x * tv[r].x + tv[r].y * tv[r].y + tv[r].z * tv[r].z + tv[r].w * tv[r].w;
    }
    sf = sqrtf(wave_sum_f(sf)); st = sqrtf(wave_sum_f(st));
    const float inf = 1.f / (sf + 1e-8f), intt = 1.f / (st + 1e-8f);
    float l = 0.f, dot = 0.f;
    float4 g[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      g[r].x = fv[r].x * inf - tv[r].x * intt; g[r].y = fv[r].y * inf - tv[r].y * intt;
      g[r].z = fv[r].z * inf - tv[r].z * intt; g[r].w = fv[r].w * inf - tv[r].w * intt;
      l += g[r].x * g[r].x + g[r].y * g[r].y + g[r].z * g[r].z + g[r].w * g[r].w;
      dot += g[r].x * fv[r].x + g[r].y * fv[r].y + g[r].z * fv[r].z + g[r].w * fv[r].w;
    }
    l = wave_sum_f(l);
    if (lane == 0) acc += l;
    if (df) {
      dot = wave_sum_f(dot);
      // y = x/(s+eps): dx = g/(s+eps) - x * (x.g) / (s (s+eps)^2)   (s>0; s==0 -> first term only)
      const float k2 = sf > 0.f ? dot * inf * inf / sf : 0.f;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int c4 = lane + 64 * r;
        if (c4 < C4) {
          float4 o;
          o.x = (g[r].x * inf - fv[r].x * k2) * gscale; o.y = (g[r].y * inf - fv[r].y * k2) * gscale;
          o.z = (g[r].z * inf - fv[r].z * k2) * gscale; o.w = (g[r].w * inf - fv[r].w * k2) * gscale;
          float* dp = df + e * dcs + c4 * 4;
          if (accumulate) {
            const float4 d = *reinterpret_cast<const float4*>(dp);
            o.x += d.x; o.y += d.y; o.z += d.z; o.w += d.w;
          }
          *reinterpret_cast<float4*>(dp) = o;
        }
      }
    }
  }
  const double r = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// ---- style: Gram of the raw images viewed (N*C) x (H*W) ----------------------------------
// stage 1: per pixel-chunk partial Gram of both images (R = N*C rows, R <= 128)
__global__ __launch_bounds__(256) void gram_partial_kernel(const float* a, int acs, const float* b, int bcs, int N,
                                                           int HW, int C, int chunk, float* partial) {
  extern __shared__ float xs[];        // [2][R][chunk + 1]  (+1: rows of different r2 land in different banks)
  const int R = N * C, cst = chunk + 1;
  const int p0 = blockIdx.x * chunk;
  const int np = min(chunk, HW - p0);
  for (int i = threadIdx.x; i < R * chunk; i += 256) {
    const int r = i / chunk, pp = i - r * chunk;
    const int n = r / C, c = r - n * C;
    float va = 0.f, vb = 0.f;
    if (pp < np) {
      va = a[((size_t)n * HW + p0 + pp) * acs + c];
      vb = b[((size_t)n * HW + p0 + pp) * bcs + c];
    }
    xs[r * cst + pp] = va; xs[R * cst + r * cst + pp] = vb;
  }
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * 2 * R * R;
  for (int i = threadIdx.x; i < R * R; i += 256) {
    const int r1 = i / R, r2 = i - r1 * R;
    float s1 = 0.f, s2 = 0.f;
    for (int pp = 0; pp < chunk; ++pp) {
      s1 = fmaf(xs[r1 * cst + pp], xs[r2 * cst + pp], s1);
      s2 = fmaf(xs[R * cst + r1 * cst + pp], xs[R * cst + r2 * cst + pp], s2);
    }
    out[i] = s1; out[R * R + i] = s2;
  }
}
// stage 2: G = sum of partials (fp64), dG = gscale * 2 * (Ga - Gb); loss partial.
// 16 Gram entries x 16 lanes per block: lane l sums chunks l, l+16, ...; fixed-order LDS sum.
__global__ __launch_bounds__(256) void gram_final_kernel(const float* partial, int nchunk, int R, float gscale,
                                                         float* dG, double* losspartial) {
  __shared__ double sa[256], sb[256];
  const int el = threadIdx.x & 15, ln = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + el;
  double ga = 0, gb = 0;
  if (i < R * R)
    for (int ch = ln; ch < nchunk; ch += 16) {
      ga += partial[(size_t)ch * 2 * R * R + i];
      gb += partial[(size_t)ch * 2 * R * R + R * R + i];
    }
  sa[threadIdx.x] = ga; sb[threadIdx.x] = gb;
  __syncthreads();
  if (threadIdx.x < 16) {
    double ta = 0, tb = 0;
    for (int j = 0; j < 16; ++j) { ta += sa[j * 16 + el]; tb += sb[j * 16 + el]; }
    const double d = ta - tb;
    sa[el] = (i < R * R) ? d * d : 0.0;
    if (i < R * R) dG[i] = (float)(2.0 * d) * gscale;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int j = 0; j < 16; ++j) t += sa[j];
    losspartial[blockIdx.x] = t;
  }
}
// stage 3: dA[r][p] (+)= sum_r' (dG[r][r'] + dG[r'][r]) * A[r'][p]
__global__ __launch_bounds__(256) void gram_bwd_kernel(const float* a, int acs, const float* dG, int N, int HW, int C,
                                                       float* da, int dcs, int accumulate) {
  extern __shared__ float g[];       // [R][R] symmetrised
  const int R = N * C;
  for (int i = threadIdx.x; i < R * R; i += 256) {
    const int r1 = i / R, r2 = i - r1 * R;
    g[i] = dG[i] + dG[r2 * R + r1];
  }
  __syncthreads();
  const size_t total = (size_t)HW * R;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int pp = (int)(i / R), r = (int)(i - (size_t)pp * R);
    const int n = r / C, c = r - n * C;
    float s = 0.f;
    for (int r2 = 0; r2 < R; ++r2) {
      const int n2 = r2 / C, c2 = r2 - n2 * C;
      s = fmaf(g[r * R + r2], a[((size_t)n2 * HW + pp) * acs + c2], s);
    }
    float* dp = da + ((size_t)n * HW + pp) * dcs + c;
    if (accumulate) s += *dp;
    *dp = s;
  }
}

// ---- generic image-Gram kernels: any R = N*C (large batches, data-parallel global batches) -------------------------
// G = X X^T for both images as 32 x 32 output tiles, the pixel axis split over blockIdx.y (deterministic partials, same
// layout as gram_partial_kernel: partial[split][2][R][R])
__global__ __launch_bounds__(256) void gram_tile_kernel(const float* a, int acs, const float* b, int bcs, int N, int HW, int C,
                                                        int per_split, float* partial) {
  __shared__ float xa[2][32][65], xb[2][32][65];     // [row panel | column panel][32 rows][64 px (+1)]
  const int R = N * C, tr = (R + 31) / 32;
  const int t0 = blockIdx.x / tr, t1 = blockIdx.x % tr;
  const int r0 = t0 * 32, c0 = t1 * 32;
  const int p_begin = blockIdx.y * per_split, p_end = min(HW, p_begin + per_split);
  const int ty = threadIdx.x >> 5, tx = threadIdx.x & 31;          // thread owns entries (ty + 8 k, tx), k = 0..3
  float sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
  for (int p0 = p_begin; p0 < p_end; p0 += 64) {
    for (int i = threadIdx.x; i < 2 * 32 * 64; i += 256) {
      const int panel = i >> 11, rr = (i >> 6) & 31, pp = i & 63;
      const int r = (panel ? c0 : r0) + rr, pix = p0 + pp;
      float va = 0.f, vb = 0.f;
      if (r < R && pix < p_end) {
        const int n = r / C, c = r - n * C;
        va = a[((size_t)n * HW + pix) * acs + c];
        vb = b[((size_t)n * HW + pix) * bcs + c];
      }
      xa[panel][rr][pp] = va; xb[panel][rr][pp] = vb;
    }
    __syncthreads();
#pragma unroll 4
    for (int pp = 0; pp < 64; ++pp) {
      const float ca = xa[1][tx][pp], cb = xb[1][tx][pp];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sa[k] = fmaf(xa[0][ty + 8 * k][pp], ca, sa[k]);
        sb[k] = fmaf(xb[0][ty + 8 * k][pp], cb, sb[k]);
      }
    }
    __syncthreads();
  }
  float* out = partial + (size_t)blockIdx.y * 2 * R * R;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    if (r < R && c < R) { out[(size_t)r * R + c] = sa[k]; out[(size_t)R * R + (size_t)r * R + c] = sb[k]; }
  }
}
// dA[r][p] (+)= sum_r' (dG[r][r'] + dG[r'][r]) A[r'][p] for the local rows r in [row0, row0 + Rloc): block = 32 pixels
__global__ __launch_bounds__(256) void gram_bwd_generic_kernel(const float* a, int acs, const float* dG, int N, int HW, int C,
                                                               int row0, int Rloc, float* da, int dcs, int accumulate) {
  extern __shared__ float xs[];       // [R][33]
  const int R = N * C;
  const int p0 = blockIdx.x * 32;
  for (int i = threadIdx.x; i < R * 32; i += 256) {
    const int r = i >> 5, pp = i & 31;
    const int n = r / C, c = r - n * C;
    xs[r * 33 + pp] = p0 + pp < HW ? a[((size_t)n * HW + p0 + pp) * acs + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Rloc * 32; i += 256) {
    const int rl = i >> 5, pp = i & 31;
    if (p0 + pp >= HW) continue;
    const int r = row0 + rl;
    float s = 0.f;
    for (int r2 = 0; r2 < R; ++r2) s = fmaf(dG[(size_t)r * R + r2] + dG[(size_t)r2 * R + r], xs[r2 * 33 + pp], s);
    const int nl = rl / C, c = rl - nl * C;
    float* dp = da + ((size_t)nl * HW + p0 + pp) * dcs + c;
    if (accumulate) s += *dp;
    *dp = s;
  }
}

// ---- gradient penalty helpers (ops.h) ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gp_interp_kernel(const float* a, int acs, const float* b, int bcs, const float* alpha,
                                                        const float* beta, int becs, const float* half_std, float* out,
                                                        int ocs, int N, size_t HW, int C) {
  const size_t total = (size_t)N * HW * C;
  const float hs = half_std ? half_std[0] : 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t e = i / C;
    const int c = (int)(i - e * C);
    const int n = (int)(e / HW);
    const float av = a[e * acs + c];
    const float bv = b ? b[e * bcs + c] : av + hs * beta[e * becs + c];
    out[e * ocs + c] = av + alpha[n] * (bv - av);
  }
}
__global__ __launch_bounds__(256) void gp_sumsq_kernel(const float* g, int gcs, size_t per_sample, int C, int nchunk,
                                                       double* partial, int with_sum) {
  // grid (nchunk, N): partial[(n*nchunk + ch)*2] = sum x^2 (and sum x) over a strided share of sample n
  __shared__ double sh[4];
  const int n = blockIdx.y;
  double s2 = 0, s1 = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per_sample; i += (size_t)gridDim.x * 256) {
    const size_t e = i / C;
    const int c = (int)(i - e * C);
    const double v = g[((size_t)n * (per_sample / C) + e) * gcs + c];
    s2 += v * v; s1 += v;
  }
  const double r2 = block_sum(s2, sh);
  const double r1 = with_sum ? block_sum(s1, sh) : 0.0;
  if (threadIdx.x == 0) { partial[((size_t)n * nchunk + blockIdx.x) * 2] = r2; partial[((size_t)n * nchunk + blockIdx.x) * 2 + 1] = r1; }
}
// one block: per-sample norms -> penalty value and the per-sample gradient coefficients
__global__ __launch_bounds__(64) void gp_finalize_kernel(const double* partial, int N, int nchunk, int lp, float scale,
                                                         float* loss_out, float* coef) {
  if (threadIdx.x != 0) return;
  double loss = 0;
  for (int n = 0; n < N; ++n) {
    double s2 = 0;
    for (int ch = 0; ch < nchunk; ++ch) s2 += partial[((size_t)n * nchunk + ch) * 2];
    const double norm = sqrt(s2);
    double d = norm - 1.0;
    if (lp && d < 0) d = 0;
    loss += d * d;
    coef[n] = norm > 0 ? (float)((double)scale * 2.0 * d / ((double)N * norm)) : 0.f;
  }
  loss_out[0] = (float)(loss / N);
}
__global__ __launch_bounds__(256) void gp_scale_kernel(const float* g, int gcs, const float* coef, float* u, int ucs, int N,
                                                       size_t HW, int C) {
  const size_t total = (size_t)N * HW * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t e = i / C;
    const int c = (int)(i - e * C);
    u[e * ucs + c] = coef[e / HW] * g[e * gcs + c];
  }
}
__global__ __launch_bounds__(64) void gp_std_finalize_kernel(const double* partial, int nblocks, double numel, float* out) {
  if (threadIdx.x != 0) return;
  double s2 = 0, s1 = 0;
  for (int i = 0; i < nblocks; ++i) { s2 += partial[(size_t)i * 2]; s1 += partial[(size_t)i * 2 + 1]; }
  const double mean = s1 / numel;
  double var = (s2 - numel * mean * mean) / (numel - 1.0);
  if (var < 0) var = 0;
  out[0] = (float)(0.5 * sqrt(var));
}
__global__ __launch_bounds__(256) void gp_uniform_kernel(float* v, int cs, size_t pixels, int C, int Clog, uint64_t seed,
                                                          const int32_t* cimap) {
  const size_t total = pixels * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t e = i / C;
    const int c = (int)(i - e * C);
    const bool live = c < Clog && (!cimap || cimap[c] >= 0);
    v[e * cs + c] = live ? (float)(mix64(seed * 0xD1342543DE82EF95ull + i) & 0xFFFFFFu) * (1.0f / 16777216.0f) : 0.f;
  }
}

__global__ void scalar_axpby_kernel(const float* a, float ca, const float* b, float cb, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (a ? a[0] * ca : 0.f) + (b ? b[0] * cb : 0.f);
}

template <int MODE>
void gan_loss(Stream& s, const TView& pred, float label, float scale, float* loss_out, const TView* dpred, const float* label_dev) {
  const size_t numel = pred.pixels();
  const int grid = loss_grid(numel);
  double* partial = reinterpret_cast<double*>(s.ws);
  const float gs = scale / (float)numel;
  hipLaunchKernelGGL(gan_loss_kernel<MODE>, dim3(grid), dim3(256), 0, hs(s), pred.p, pred.cs, numel, label, gs,
                     dpred ? dpred->p : nullptr, dpred ? dpred->cs : 0, partial, label_dev);
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, hs(s), partial, grid, 1.0 / (double)numel, loss_out);
  check_launch("gan_loss");
}

}  // namespace

void bce_logits_loss(Stream& s, const TView& pred, float label, float scale, float* loss_out, const TView* dpred, const float* label_dev) {
  gan_loss<0>(s, pred, label, scale, loss_out, dpred, label_dev);
}
void lsgan_loss(Stream& s, const TView& pred, float label, float scale, float* loss_out, const TView* dpred, const float* label_dev) {
  gan_loss<1>(s, pred, label, scale, loss_out, dpred, label_dev);
}
void wgan_loss(Stream& s, const TView& pred, float sign, float scale, float* loss_out, const TView* dpred) {
  gan_loss<2>(s, pred, sign, scale, loss_out, dpred, nullptr);
}

void ce_argmax_loss(Stream& s, const TView& logits, const TView& target, int C, float scale, float* loss_out,
                    const TView* dlogits, int accumulate) {
  const size_t pixels = logits.pixels();
  if (C > 32 || logits.cs % 4 || target.cs % 4 || ((uintptr_t)logits.p & 15) || ((uintptr_t)target.p & 15) ||
      (dlogits && (dlogits->cs % 4 || ((uintptr_t)dlogits->p & 15))))
    throw Error(1, "ce_argmax_loss: needs C <= 32 and 16-byte aligned views");
  const int grid = loss_grid(pixels);
  double* partial = reinterpret_cast<double*>(s.ws);
  hipLaunchKernelGGL(ce_kernel, dim3(grid), dim3(256), 0, hs(s), logits.p, logits.cs, target.p, target.cs, C, pixels,
                     scale / (float)pixels, dlogits ? dlogits->p : nullptr, dlogits ? dlogits->cs : 0, accumulate, partial);
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, hs(s), partial, grid, 1.0 / (double)pixels, loss_out);
  check_launch("ce_argmax_loss");
}

void l1_loss(Stream& s, const TView& a, const TView& b, int C, float scale, float* loss_out, const TView* da,
             int accumulate) {
  const size_t pixels = a.pixels();
  const size_t numel = pixels * C;
  const int grid = loss_grid(numel);
  double* partial = reinterpret_cast<double*>(s.ws);
  hipLaunchKernelGGL(l1_kernel, dim3(grid), dim3(256), 0, hs(s), a.p, a.cs, b.p, b.cs, C, pixels, scale / (float)numel,
                     da ? da->p : nullptr, da ? da->cs : 0, accumulate, partial);
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, hs(s), partial, grid, 1.0 / (double)numel, loss_out);
  check_launch("l1_loss");
}

void normed_mse_loss(Stream& s, const TView& f, const TView& t, float scale, float* loss_out, const TView* df,
                     int accumulate) {
  if (f.C % 4 || f.C > 512 || t.C != f.C) throw Error(1, "normed_mse_loss: C must be a multiple of 4, <= 512");
  const size_t pixels = f.pixels();
  const size_t numel = pixels * f.C;
  const int grid = (int)std::min<size_t>((pixels + 3) / 4, 2048);
  double* partial = reinterpret_cast<double*>(s.ws);
  hipLaunchKernelGGL(normed_mse_kernel, dim3(grid), dim3(256), 0, hs(s), f.p, f.cs, t.p, t.cs, f.C, pixels,
                     2.f * scale / (float)numel, df ? df->p : nullptr, df ? df->cs : 0, accumulate, partial);
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, hs(s), partial, grid, 1.0 / (double)numel, loss_out);
  check_launch("normed_mse_loss");
}

void gram_style_loss(Stream& s, const TView& a, const TView& b, int C, float scale, float* loss_out, const TView* da,
                     int accumulate, int n0, int nloc) {
  const int R = a.N * C, HW = a.H * a.W;
  if (nloc < 0) { n0 = 0; nloc = a.N; }
  const double numel = (double)R * R;
  const int fgrid = ceil_div(R * R, 16);
  if (R <= 128 && n0 == 0 && nloc == a.N) {        // whole Gram in LDS (the single-GPU default: 16 x 3 = 48 rows)
    const int chunk = 64;
    const int nchunk = ceil_div(HW, chunk);
    const size_t pbytes = (size_t)nchunk * 2 * R * R * 4;
    const size_t off_dG = (pbytes + 255) / 256 * 256;
    const size_t off_lp = off_dG + (size_t)R * R * 4 + 256;
    if (off_lp + 256 + (size_t)fgrid * 8 > s.ws_bytes) throw Error(1, "gram_style_loss: workspace too small");
    float* partial = reinterpret_cast<float*>(s.ws);
    float* dG = reinterpret_cast<float*>(s.ws + off_dG);
    double* lp = reinterpret_cast<double*>(s.ws + (off_lp + 255) / 256 * 256);
    hipLaunchKernelGGL(gram_partial_kernel, dim3(nchunk), dim3(256), 2 * R * (chunk + 1) * 4, hs(s), a.p, a.cs, b.p, b.cs,
                       a.N, HW, C, chunk, partial);
    hipLaunchKernelGGL(gram_final_kernel, dim3(fgrid), dim3(256), 0, hs(s), partial, nchunk, R, (float)(scale / numel),
                       dG, lp);
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, hs(s), lp, fgrid, 1.0 / numel, loss_out);
    if (da) {
      const size_t total = (size_t)HW * R;
      hipLaunchKernelGGL(gram_bwd_kernel, dim3(loss_grid(total)), dim3(256), R * R * 4, hs(s), a.p, a.cs, dG, a.N, HW, C,
                         da->p, da->cs, accumulate);
    }
    check_launch("gram_style_loss");
    return;
  }
  // generic path: large batches (R > 128) and the data-parallel form (global Gram, gradient for the local samples)
  if (R > 1024) throw Error(1, "gram_style_loss: N*C > 1024 unsupported");
  const int nsplit = std::max(1, std::min(64, HW / 256));
  const int per_split = ceil_div(ceil_div(HW, nsplit), 64) * 64;
  const int ns = ceil_div(HW, per_split);
  const size_t pbytes = (size_t)ns * 2 * R * R * 4;
  const size_t off_dG = (pbytes + 255) / 256 * 256;
  const size_t off_lp = off_dG + ((size_t)R * R * 4 + 255) / 256 * 256;
  if (off_lp + 256 + (size_t)fgrid * 8 > s.ws_bytes) throw Error(1, "gram_style_loss: workspace too small");
  float* partial = reinterpret_cast<float*>(s.ws);
  float* dG = reinterpret_cast<float*>(s.ws + off_dG);
  double* lp = reinterpret_cast<double*>(s.ws + off_lp);
  const int tr = ceil_div(R, 32);
  hipLaunchKernelGGL(gram_tile_kernel, dim3(tr * tr, ns), dim3(256), 0, hs(s), a.p, a.cs, b.p, b.cs, a.N, HW, C, per_split, partial);
  hipLaunchKernelGGL(gram_final_kernel, dim3(fgrid), dim3(256), 0, hs(s), partial, ns, R, (float)(scale / numel), dG, lp);
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, hs(s), lp, fgrid, 1.0 / numel, loss_out);
  if (da) {
    const size_t smem = (size_t)R * 33 * 4;
    // R * 33 floats of LDS (up to 135 KB at R = 1024): fits gfx950's 160 KB only -- fail with a clear message elsewhere
    static int lds_limit = -1;
    if (lds_limit < 0) {
      int dev = 0, lim = 0;
      SWN_HIP_CHECK(hipGetDevice(&dev));
      SWN_HIP_CHECK(hipDeviceGetAttribute(&lim, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
      SWN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gram_bwd_generic_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, std::min(lim, 1024 * 33 * 4)));
      lds_limit = lim;
    }
    if (smem > (size_t)lds_limit)
      throw Error(1, "gram_style_loss: the style gradient over R = " + std::to_string(R) + " Gram rows needs " + std::to_string(smem) +
                         " bytes of LDS, this device offers " + std::to_string(lds_limit));
    hipLaunchKernelGGL(gram_bwd_generic_kernel, dim3(ceil_div(HW, 32)), dim3(256), smem, hs(s), a.p, a.cs, dG, a.N, HW, C,
                       n0 * C, nloc * C, da->p, da->cs, accumulate);
  }
  check_launch("gram_style_loss");
}

void gp_interpolate(Stream& s, const TView& a, const TView* b, const float* alpha, const TView* beta, const float* half_std,
                    const TView& out) {
  if (!b && !(beta && half_std)) throw Error(1, "gp_interpolate: dragan form needs beta and half_std");
  const size_t HW = (size_t)a.H * a.W;
  hipLaunchKernelGGL(gp_interp_kernel, dim3(loss_grid((size_t)a.N * HW * a.C)), dim3(256), 0, hs(s), a.p, a.cs, b ? b->p : nullptr,
                     b ? b->cs : 0, alpha, beta ? beta->p : nullptr, beta ? beta->cs : 0, half_std, out.p, out.cs, a.N, HW, a.C);
  check_launch("gp_interpolate");
}
void gp_half_std(Stream& s, const TView& a, size_t numel, float* out) {
  const size_t per = (size_t)a.H * a.W * a.C;
  const int nchunk = 64;
  double* partial = reinterpret_cast<double*>(s.ws);
  hipLaunchKernelGGL(gp_sumsq_kernel, dim3(nchunk, a.N), dim3(256), 0, hs(s), a.p, a.cs, per, a.C, nchunk, partial, 1);
  hipLaunchKernelGGL(gp_std_finalize_kernel, dim3(1), dim3(64), 0, hs(s), partial, nchunk * a.N, (double)numel, out);
  check_launch("gp_half_std");
}
void gp_penalty(Stream& s, const TView& g, int lp, float scale, float* loss_out, const TView& u) {
  const size_t HW = (size_t)g.H * g.W, per = HW * g.C;
  const int nchunk = 64;
  double* partial = reinterpret_cast<double*>(s.ws);
  float* coef = reinterpret_cast<float*>(s.ws + (size_t)nchunk * g.N * 16 + 256);
  hipLaunchKernelGGL(gp_sumsq_kernel, dim3(nchunk, g.N), dim3(256), 0, hs(s), g.p, g.cs, per, g.C, nchunk, partial, 0);
  hipLaunchKernelGGL(gp_finalize_kernel, dim3(1), dim3(64), 0, hs(s), partial, g.N, nchunk, lp, scale, loss_out, coef);
  hipLaunchKernelGGL(gp_scale_kernel, dim3(loss_grid((size_t)g.N * per)), dim3(256), 0, hs(s), g.p, g.cs, coef, u.p, u.cs, g.N, HW, g.C);
  check_launch("gp_penalty");
}
void gp_uniform(Stream& s, const TView& v, int Clog, uint64_t seed, const int32_t* cimap) {
  hipLaunchKernelGGL(gp_uniform_kernel, dim3(loss_grid(v.pixels() * v.C)), dim3(256), 0, hs(s), v.p, v.cs, v.pixels(), v.C, Clog, seed,
                     cimap);
  check_launch("gp_uniform");
}

void scalar_axpby(Stream& s, const float* a, float ca, const float* b, float cb, float* out) {
  hipLaunchKernelGGL(scalar_axpby_kernel, dim3(1), dim3(64), 0, hs(s), a, ca, b, cb, out);
  check_launch("scalar_axpby");
}

}  // namespace swn
