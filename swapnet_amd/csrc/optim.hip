// swapnet_amd -- fused AdamW over a whole parameter arena + weight layout transforms.
// Reference: optimizers/__init__.py:37-60 -> torch.optim.AdamW (decoupled weight decay,
// eps 1e-8, amsgrad off); state-dict weight layouts of Conv2d (Co,Ci,KH,KW) and
// ConvTranspose2d (Ci,Co,4,4) (modules/layers.py:15,31).
// AdamW is 28 B/param of HBM traffic (read p,g,m,v; write p,m,v) -> purely bandwidth bound:
// one launch over the contiguous arena, 16-byte accesses, grid-stride.
#include "hip_util.h"

namespace swn {
namespace {

__global__ __launch_bounds__(256) void adamw_kernel(AdamWArgs a, float decay, float step_size, float inv_sqrt_bc2) {
  if (a.sched_dev) { step_size = a.sched_dev[0]; inv_sqrt_bc2 = a.sched_dev[1]; }      // captured step: this step's bias corrections
  const size_t n4 = a.n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 p = reinterpret_cast<float4*>(a.p)[i];
    const float4 g = reinterpret_cast<const float4*>(a.g)[i];
    float4 m = reinterpret_cast<float4*>(a.m)[i];
    float4 v = reinterpret_cast<float4*>(a.v)[i];
    float* pp = &p.x; const float* gp = &g.x; float* mp = &m.x; float* vp = &v.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // torch.optim.adamw single-tensor order: p*=1-lr*wd; m=b1*m+(1-b1)g; v=b2*v+(1-b2)g*g;
      // denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) * m/denom
      float pj = pp[j] * decay;
      const float mj = mp[j] * a.beta1 + (1.f - a.beta1) * gp[j];
      const float vj = vp[j] * a.beta2 + (1.f - a.beta2) * gp[j] * gp[j];
      const float denom = sqrtf(vj) * inv_sqrt_bc2 + a.eps;
      pj -= step_size * (mj / denom);
      pp[j] = pj; mp[j] = mj; vp[j] = vj;
    }
    reinterpret_cast<float4*>(a.p)[i] = p;
    reinterpret_cast<float4*>(a.m)[i] = m;
    reinterpret_cast<float4*>(a.v)[i] = v;
  }
}

struct PackP {
  WShape w;
  const float* src; float* dst;
  int mode, Cop, Ndg;
  size_t total;
};

__device__ __forceinline__ int ref_channel(const WShape& w, int cb) {
  if (w.cimap) return w.cimap[cb];
  return cb < w.Ci ? cb : -1;
}

// dst = packed ; src = NCHW(torch) weights
__global__ void pack_kernel(PackP p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.total) return;
  const WShape& w = p.w;
  float v = 0.f;
  if (w.kind == WK_CONV) {
    const int k = (int)(i / w.Npad), n = (int)(i - (size_t)k * w.Npad);
    const int tap = k / w.Cip, cb = k - tap * w.Cip;
    const int ci = ref_channel(w, cb);
    const int kh = tap / w.KW, kw = tap - kh * w.KW;
    if (n < w.Co && ci >= 0) v = p.src[(((size_t)n * w.Ci + ci) * w.KH + kh) * w.KW + kw];
  } else {
    const int Kp = 4 * w.Cip;
    const size_t per = (size_t)Kp * w.Npad;
    const int ph = (int)(i / per);
    const size_t rem = i - (size_t)ph * per;
    const int k = (int)(rem / w.Npad), n = (int)(rem - (size_t)k * w.Npad);
    const int t = k / w.Cip, cb = k - t * w.Cip;
    const int ci = ref_channel(w, cb);
    const int a = ph >> 1, b = ph & 1, dy = t >> 1, dx = t & 1;
    const int ky = 3 - a - 2 * dy, kx = 3 - b - 2 * dx;
    if (n < w.Co && ci >= 0) v = p.src[(((size_t)ci * w.Co + n) * 4 + ky) * 4 + kx];
  }
  p.dst[i] = v;
}

// dst = NCHW(torch) ; src = packed       (total = Co*Ci*KH*KW)
__global__ void unpack_kernel(PackP p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.total) return;
  const WShape& w = p.w;
  if (w.kind == WK_CONV) {
    const int kw = (int)(i % w.KW); size_t r = i / w.KW;
    const int kh = (int)(r % w.KH); r /= w.KH;
    const int ci = (int)(r % w.Ci); const int co = (int)(r / w.Ci);
    int cb = ci;
    if (w.cimap) { cb = -1; for (int j = 0; j < w.Cip; ++j) if (w.cimap[j] == ci) { cb = j; break; } }
    p.dst[i] = cb >= 0 ? p.src[((size_t)(kh * w.KW + kw) * w.Cip + cb) * w.Npad + co] : 0.f;
  } else {
    const int kx = (int)(i % 4); size_t r = i / 4;
    const int ky = (int)(r % 4); r /= 4;
    const int co = (int)(r % w.Co); const int ci = (int)(r / w.Co);
    const int a = (3 - ky) & 1, dy = (3 - ky) >> 1, b = (3 - kx) & 1, dx = (3 - kx) >> 1;
    const int Kp = 4 * w.Cip;
    int cb = ci;
    if (w.cimap) { cb = -1; for (int j = 0; j < w.Cip; ++j) if (w.cimap[j] == ci) { cb = j; break; } }
    p.dst[i] = cb >= 0 ? p.src[((size_t)(a * 2 + b) * Kp + (dy * 2 + dx) * w.Cip + cb) * w.Npad + co] : 0.f;
  }
}

// dgrad operand from forward-packed weights (dest-indexed)
__global__ void repack_dgrad_kernel(PackP p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.total) return;
  const WShape& w = p.w;
  const int Cop = p.Cop, Ndg = p.Ndg;
  float v = 0.f;
  if (p.mode == 0) {          // conv k4 s2: phase blocks [(dy*2+dx)*Cop+co][ci]
    const size_t per = (size_t)4 * Cop * Ndg;
    const int ph = (int)(i / per);
    const size_t rem = i - (size_t)ph * per;
    const int k = (int)(rem / Ndg), ci = (int)(rem - (size_t)k * Ndg);
    const int t = k / Cop, co = k - t * Cop;
    const int a = ph >> 1, b = ph & 1, dy = t >> 1, dx = t & 1;
    const int ky = 3 - a - 2 * dy, kx = 3 - b - 2 * dx;
    if (co < w.Co && ci < w.Cip) v = p.src[((size_t)(ky * w.KW + kx) * w.Cip + ci) * w.Npad + co];
  } else if (p.mode == 1) {   // conv stride 1: flipped taps
    const int k = (int)(i / Ndg), ci = (int)(i - (size_t)k * Ndg);
    const int t = k / Cop, co = k - t * Cop;
    const int khf = t / w.KW, kwf = t - khf * w.KW;
    const int kh = w.KH - 1 - khf, kw = w.KW - 1 - kwf;
    if (co < w.Co && ci < w.Cip) v = p.src[((size_t)(kh * w.KW + kw) * w.Cip + ci) * w.Npad + co];
  } else if (p.mode == 2) {   // convT k4 s2: [(ky*4+kx)*Cop+co][ci]
    const int k = (int)(i / Ndg), ci = (int)(i - (size_t)k * Ndg);
    const int t = k / Cop, co = k - t * Cop;
    const int ky = t >> 2, kx = t & 3;
    const int a = (3 - ky) & 1, dy = (3 - ky) >> 1, b = (3 - kx) & 1, dx = (3 - kx) >> 1;
    const int Kp = 4 * w.Cip;
    if (co < w.Co && ci < w.Cip)
      v = p.src[((size_t)(a * 2 + b) * Kp + (dy * 2 + dx) * w.Cip + ci) * w.Npad + co];
  } else {                    // tail: 5x5 stride-2 effective kernel
    const int k = (int)(i / Ndg), ci = (int)(i - (size_t)k * Ndg);
    const int t = k / Cop, co = k - t * Cop;
    const int r = t / 5, c = t - r * 5;
    if (co < w.Co && ci < w.Cip) {
      for (int a = 0; a < 2; ++a) {
        const int ky = a + 3 - r;
        if (ky < 0 || ky > 3) continue;
        for (int b = 0; b < 2; ++b) {
          const int kx = b + 3 - c;
          if (kx < 0 || kx > 3) continue;
          v += p.src[((size_t)(ky * 4 + kx) * w.Cip + ci) * w.Npad + co];
        }
      }
    }
  }
  p.dst[i] = v;
}

__device__ __forceinline__ int fold_tap(int a, int k) { return a ? (k + 1) >> 1 : k >> 1; }

// MODE 0: folded[p][r][c][ci][n] = sum of the packed taps that fold onto (r,c); MODE 1: unfold grads
template <int MODE>
__global__ void tail_fold_kernel(WShape w, const float* src, float* dst, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t per_tap = (size_t)w.Cip * w.Npad;
  if (MODE == 0) {
    // locate phase
    size_t off = i; int ph = 0;
    for (; ph < 4; ++ph) {
      const size_t sz = (size_t)(2 + (ph >> 1)) * (2 + (ph & 1)) * per_tap;
      if (off < sz) break;
      off -= sz;
    }
    const int a = ph >> 1, b = ph & 1, KWp = 2 + b;
    const int tap = (int)(off / per_tap); const size_t rem = off - (size_t)tap * per_tap;
    const int r = tap / KWp, c = tap - r * KWp;
    float v = 0.f;
    for (int ky = 0; ky < 4; ++ky) {
      if (fold_tap(a, ky) != r) continue;
      for (int kx = 0; kx < 4; ++kx)
        if (fold_tap(b, kx) == c) v += src[(size_t)(ky * 4 + kx) * per_tap + rem];
    }
    dst[i] = v;
  } else {
    const int tap = (int)(i / per_tap); const size_t rem = i - (size_t)tap * per_tap;
    const int ky = tap >> 2, kx = tap & 3;
    float v = 0.f;
    size_t base = 0;
    for (int ph = 0; ph < 4; ++ph) {
      const int a = ph >> 1, b = ph & 1, KWp = 2 + b;
      v += src[base + (size_t)(fold_tap(a, ky) * KWp + fold_tap(b, kx)) * per_tap + rem];
      base += (size_t)(2 + a) * KWp * per_tap;
    }
    dst[i] = v;
  }
}

inline unsigned blocks(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

void adamw_schedule(float lr, float beta1, float beta2, int step, float out[2]) {
  const double bc1 = 1.0 - pow((double)beta1, step);
  const double bc2 = 1.0 - pow((double)beta2, step);
  out[0] = (float)(lr / bc1);
  out[1] = (float)(1.0 / sqrt(bc2));
}
void adamw_step(Stream& s, const AdamWArgs& a) {
  if (a.n % 4) throw Error(1, "adamw_step: arena size must be a multiple of 4");
  float sched[2];
  adamw_schedule(a.lr, a.beta1, a.beta2, a.step, sched);
  const float decay = 1.f - a.lr * a.weight_decay;
  const float step_size = sched[0];
  const float inv_sqrt_bc2 = sched[1];
  const unsigned grid = (unsigned)std::min<size_t>(std::max<size_t>((a.n / 4 + 255) / 256, 1), 256 * 16);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, hs(s), a, decay, step_size, inv_sqrt_bc2);
  check_launch("adamw_step");
}

size_t packed_elems(const WShape& w) {
  if (w.kind == WK_CONV) return (size_t)w.KH * w.KW * w.Cip * w.Npad;
  return (size_t)4 * 4 * w.Cip * w.Npad;
}

void pack_weight(Stream& s, const WShape& w, const float* nchw, float* packed) {
  PackP p{}; p.w = w; p.src = nchw; p.dst = packed; p.total = packed_elems(w);
  hipLaunchKernelGGL(pack_kernel, dim3(blocks(p.total)), dim3(256), 0, hs(s), p);
  check_launch("pack_weight");
}
void unpack_weight(Stream& s, const WShape& w, const float* packed, float* nchw) {
  PackP p{}; p.w = w; p.src = packed; p.dst = nchw; p.total = (size_t)w.Co * w.Ci * w.KH * w.KW;
  hipLaunchKernelGGL(unpack_kernel, dim3(blocks(p.total)), dim3(256), 0, hs(s), p);
  check_launch("unpack_weight");
}

size_t tail_fold_offset(const WShape& w, int phase) {
  size_t off = 0;
  for (int p = 0; p < phase && p < 4; ++p) off += (size_t)(2 + (p >> 1)) * (2 + (p & 1)) * w.Cip * w.Npad;
  return off;
}
void tail_fold_weights(Stream& s, const WShape& w, const float* packed, float* folded) {
  const size_t total = tail_fold_offset(w, 4);
  hipLaunchKernelGGL(tail_fold_kernel<0>, dim3(blocks(total)), dim3(256), 0, hs(s), w, packed, folded, total);
  check_launch("tail_fold_weights");
}
void tail_unfold_wgrad(Stream& s, const WShape& w, const float* dfolded, float* dpacked) {
  const size_t total = (size_t)16 * w.Cip * w.Npad;
  hipLaunchKernelGGL(tail_fold_kernel<1>, dim3(blocks(total)), dim3(256), 0, hs(s), w, dfolded, dpacked, total);
  check_launch("tail_unfold_wgrad");
}

size_t dgrad_elems(const WShape& w, int mode, int Cop, int Ndgpad) {
  switch (mode) {
    case 0: return (size_t)4 * 4 * Cop * Ndgpad;
    case 1: return (size_t)w.KH * w.KW * Cop * Ndgpad;
    case 2: return (size_t)16 * Cop * Ndgpad;
    default: return (size_t)25 * Cop * Ndgpad;
  }
}
void repack_dgrad(Stream& s, const WShape& w, int mode, int Cop, int Ndgpad, const float* packed, float* dg) {
  PackP p{}; p.w = w; p.src = packed; p.dst = dg; p.mode = mode; p.Cop = Cop; p.Ndg = Ndgpad;
  p.total = dgrad_elems(w, mode, Cop, Ndgpad);
  hipLaunchKernelGGL(repack_dgrad_kernel, dim3(blocks(p.total)), dim3(256), 0, hs(s), p);
  check_launch("repack_dgrad");
}

}  // namespace swn
