// swapnet_amd -- gather / resampling / layout / integer kernels (all HBM- or gather-bound).
// Reference: torchvision.ops.RoIAlign @0.4.0 (call site modules/swapnet_modules.py:166-168,234;
// SURVEY.md Appendix B), nn.functional.interpolate nearest (:244-247), MaxPool2d(2,2) of
// VGG16 (modules/losses/perceptual.py:26-42), util/decode_labels.py:24-55,
// datasets/data_utils.py:311-343.
#include "hip_util.h"

namespace swn {
namespace {

inline unsigned egrid(size_t total) {
  return (unsigned)std::min<size_t>(std::max<size_t>((total + 255) / 256, 1), 256 * 32);
}

struct RoiSample { int yl, yh, xl, xh; float w1, w2, w3, w4; bool valid; };

// legacy (unaligned) RoIAlign sample for bin (ph,pw), sampling_ratio 1, spatial_scale 1.
// fp32 arithmetic in the published order, contraction disabled so the truncations that
// produce the integer corner indices are bit-identical to the CPU reference.
__device__ __forceinline__ RoiSample roi_sample(const float* roi, int ph, int pw, int PH, int PW, int H, int W) {
  RoiSample r;
  const float sw = roi[0], sh = roi[1], ew = roi[2], eh = roi[3];
  const float rw = fmaxf(__fsub_rn(ew, sw), 1.0f);
  const float rh = fmaxf(__fsub_rn(eh, sh), 1.0f);
  const float bw = __fdiv_rn(rw, (float)PW);
  const float bh = __fdiv_rn(rh, (float)PH);
  float y = __fadd_rn(__fadd_rn(sh, __fmul_rn((float)ph, bh)), __fdiv_rn(__fmul_rn(0.5f, bh), 1.0f));
  float x = __fadd_rn(__fadd_rn(sw, __fmul_rn((float)pw, bw)), __fdiv_rn(__fmul_rn(0.5f, bw), 1.0f));
  r.valid = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
  y = fmaxf(y, 0.f); x = fmaxf(x, 0.f);
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else { yh = yl + 1; }
  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else { xh = xl + 1; }
  const float ly = __fsub_rn(y, (float)yl), lx = __fsub_rn(x, (float)xl);
  const float hy = __fsub_rn(1.f, ly), hx = __fsub_rn(1.f, lx);
  r.yl = yl; r.yh = yh; r.xl = xl; r.xh = xh;
  r.w1 = __fmul_rn(hy, hx); r.w2 = __fmul_rn(hy, lx); r.w3 = __fmul_rn(ly, hx); r.w4 = __fmul_rn(ly, lx);
  return r;
}

// one thread per (b, ph, pw, roi): the 12 ROI threads of a pixel write 36 contiguous floats
__global__ __launch_bounds__(256) void roi_align_kernel(const float* tex, int tcs, int H, int W, int C,
                                                        const float* rois, int B, int R, float* out, int ocs, int PH,
                                                        int PW) {
  const size_t total = (size_t)B * PH * PW * R;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i % R); size_t q = i / R;
    const int pw = (int)(q % PW); q /= PW;
    const int ph = (int)(q % PH); const int b = (int)(q / PH);
    const RoiSample sm = roi_sample(rois + ((size_t)b * R + r) * 4, ph, pw, PH, PW, H, W);
    float* o = out + (((size_t)b * PH + ph) * PW + pw) * ocs + r * C;
    const float* img = tex + (size_t)b * H * W * tcs;
    for (int c = 0; c < C; ++c) {
      float v = 0.f;
      if (sm.valid) {
        const float v1 = img[((size_t)sm.yl * W + sm.xl) * tcs + c], v2 = img[((size_t)sm.yl * W + sm.xh) * tcs + c];
        const float v3 = img[((size_t)sm.yh * W + sm.xl) * tcs + c], v4 = img[((size_t)sm.yh * W + sm.xh) * tcs + c];
        // (w1*v1 + w2*v2 + w3*v3 + w4*v4) evaluated left to right, unfused (ROIAlign_cpu.cpp)
        v = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(sm.w1, v1), __fmul_rn(sm.w2, v2)), __fmul_rn(sm.w3, v3)),
                      __fmul_rn(sm.w4, v4));
      }
      o[c] = v;
    }
  }
}

// The same gather with wavefront primitives (north_star: "ROIAlign/texture-pool as a wavefront-primitive gather kernel").  The
// DEFAULT since round 5 (bit-identical to the scalar kernel on the MI355X, tests/test_ops.py
// test_wavefront_gather_roi_align_is_bit_identical); SWN_ROI_WAVE=0 selects the scalar kernel.  One wavefront per (image, output row ph, 64 output columns),
// the ROIs of the pixel in a loop.  Per ROI the sample ROW is wave-uniform (y, yl, yh and the row weights depend on (roi, ph)
// only) and the lanes' source columns xl grow with pw, so the texels the wave needs from rows yl / yh are one contiguous run
// starting at lane 0's xl.  The wave loads that run 64 texels at a time -- ONE coalesced 16-byte load per lane and row (NHWC, C
// padded to 4) -- and every lane takes its four corners out of its neighbours' registers with ds_bpermute (__shfl): no per-lane
// scattered loads, each texel of the run fetched once per wave instead of up to four times per lane.  Same roi_sample(), same
// left-to-right unfused weighted sum of the same four values: bit-identical to roi_align_kernel by construction.
__global__ __launch_bounds__(256) void roi_align_wave_kernel(const float* tex, int tcs, int H, int W, int C, const float* rois, int B,
                                                             int R, float* out, int ocs, int PH, int PW) {
  const int lane = threadIdx.x & 63;
  const int nblk = (PW + 63) >> 6;
  const size_t nwork = (size_t)B * PH * nblk;
  for (size_t wv = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); wv < nwork; wv += (size_t)gridDim.x * 4) {
    const int blk = (int)(wv % nblk); size_t q = wv / nblk;
    const int ph = (int)(q % PH); const int b = (int)(q / PH);
    const int pw = blk * 64 + lane;
    const bool active = pw < PW;
    const int pwc = active ? pw : PW - 1;                  // idle lanes shadow the last column: every lane stays in the shuffles
    const float* img = tex + (size_t)b * H * W * tcs;
    float* o = out + (((size_t)b * PH + ph) * PW + pwc) * ocs;
    for (int r = 0; r < R; ++r) {
      const RoiSample sm = roi_sample(rois + ((size_t)b * R + r) * 4, ph, pwc, PH, PW, H, W);
      const float* rowl = img + (size_t)sm.yl * W * tcs;
      const float* rowh = img + (size_t)sm.yh * W * tcs;
      const int x0 = __shfl(sm.xl, 0), x1 = __shfl(sm.xh, 63);        // the run [x0, x1]: xl / xh are monotonic in pw
      float a1[3] = {0.f, 0.f, 0.f}, a2[3] = {0.f, 0.f, 0.f}, a3[3] = {0.f, 0.f, 0.f}, a4[3] = {0.f, 0.f, 0.f};
      for (int seg = x0; seg <= x1; seg += 64) {
        const int xs = min(seg + lane, W - 1);
        const float4 tl = *reinterpret_cast<const float4*>(rowl + (size_t)xs * tcs);
        const float4 th = *reinterpret_cast<const float4*>(rowh + (size_t)xs * tcs);
        const int il = sm.xl - seg, ih = sm.xh - seg;
        const bool inl = il >= 0 && il < 64, inh = ih >= 0 && ih < 64;
        const float l0 = __shfl(tl.x, il & 63), l1 = __shfl(tl.y, il & 63), l2 = __shfl(tl.z, il & 63);      // (yl, xl)
        const float m0 = __shfl(tl.x, ih & 63), m1 = __shfl(tl.y, ih & 63), m2 = __shfl(tl.z, ih & 63);      // (yl, xh)
        const float n0 = __shfl(th.x, il & 63), n1 = __shfl(th.y, il & 63), n2 = __shfl(th.z, il & 63);      // (yh, xl)
        const float p0 = __shfl(th.x, ih & 63), p1 = __shfl(th.y, ih & 63), p2 = __shfl(th.z, ih & 63);      // (yh, xh)
        if (inl) { a1[0] = l0; a1[1] = l1; a1[2] = l2; a3[0] = n0; a3[1] = n1; a3[2] = n2; }
        if (inh) { a2[0] = m0; a2[1] = m1; a2[2] = m2; a4[0] = p0; a4[1] = p1; a4[2] = p2; }
      }
      if (active) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
          if (c < C) {
            float v = 0.f;
            if (sm.valid)
              v = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(sm.w1, a1[c]), __fmul_rn(sm.w2, a2[c])), __fmul_rn(sm.w3, a3[c])),
                            __fmul_rn(sm.w4, a4[c]));
            o[r * C + c] = v;
          }
      }
    }
  }
}

__global__ void roi_indices_kernel(const float* rois, int K, int H, int W, int PH, int PW, int32_t* idx,
                                   uint8_t* valid) {
  const size_t total = (size_t)K * PH * PW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int pw = (int)(i % PW); size_t q = i / PW;
    const int ph = (int)(q % PH); const int k = (int)(q / PH);
    const RoiSample sm = roi_sample(rois + (size_t)k * 4, ph, pw, PH, PW, H, W);
    idx[i * 4 + 0] = sm.yl; idx[i * 4 + 1] = sm.yh; idx[i * 4 + 2] = sm.xl; idx[i * 4 + 3] = sm.xh;
    valid[i] = sm.valid ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* x, int xcs, float* y, int ycs, int N, int Ho,
                                                           int Wo, int C, int f, float* amax_out) {
  const int C4 = C >> 2, Hi = Ho / f, Wi = Wo / f;
  const size_t total = (size_t)N * Ho * Wo * C4;
  float amx = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4; size_t q = i / C4;
    const int ox = (int)(q % Wo); q /= Wo;
    const int oy = (int)(q % Ho); const int n = (int)(q / Ho);
    const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)n * Hi + oy / f) * Wi + ox / f) * xcs + c);
    *reinterpret_cast<float4*>(y + (((size_t)n * Ho + oy) * Wo + ox) * ycs + c) = v;
    amx = fmaxf(amx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  amax_fold(amx, amax_out);
}
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* dy, int dycs, float* dx, int dxcs, int N,
                                                           int Hi, int Wi, int C, int f, int accumulate) {
  const int C4 = C >> 2, Ho = Hi * f, Wo = Wi * f;
  const size_t total = (size_t)N * Hi * Wi * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4; size_t q = i / C4;
    const int ix = (int)(q % Wi); q /= Wi;
    const int iy = (int)(q % Hi); const int n = (int)(q / Hi);
    float4 a = make_float4(0, 0, 0, 0);
    for (int dyy = 0; dyy < f; ++dyy)
      for (int dxx = 0; dxx < f; ++dxx) {
        const float4 v = *reinterpret_cast<const float4*>(
            dy + (((size_t)n * Ho + iy * f + dyy) * Wo + ix * f + dxx) * dycs + c);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    float* d = dx + (((size_t)n * Hi + iy) * Wi + ix) * dxcs + c;
    if (accumulate) { const float4 o = *reinterpret_cast<const float4*>(d); a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
    *reinterpret_cast<float4*>(d) = a;
  }
}

// MaxPool2d(2,2).  bwd recomputes the arg-max with the forward's scan order (kh,kw) and
// strict '>' so ties go to the first element, like ATen's max_pool2d.
__global__ __launch_bounds__(256) void maxpool_kernel(const float* x, int xcs, float* y, int ycs, const float* dy,
                                                      int dycs, float* dx, int dxcs, int N, int Ho, int Wo, int C,
                                                      int bwd, int accumulate, float* amax_out) {
  const int C4 = C >> 2, Hi = Ho * 2, Wi = Wo * 2;
  const size_t total = (size_t)N * Ho * Wo * C4;
  float amx = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4; size_t q = i / C4;
    const int ox = (int)(q % Wo); q /= Wo;
    const int oy = (int)(q % Ho); const int n = (int)(q / Ho);
    float4 v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
      v[t] = *reinterpret_cast<const float4*>(x + (((size_t)n * Hi + oy * 2 + (t >> 1)) * Wi + ox * 2 + (t & 1)) * xcs + c);
    float m[4]; int am[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* f0 = &v[0].x;
      m[j] = f0[j]; am[j] = 0;
#pragma unroll
      for (int t = 1; t < 4; ++t) {
        const float val = (&v[t].x)[j];
        if (val > m[j] || val != val) { m[j] = val; am[j] = t; }
      }
    }
    if (!bwd) {
      *reinterpret_cast<float4*>(y + (((size_t)n * Ho + oy) * Wo + ox) * ycs + c) = make_float4(m[0], m[1], m[2], m[3]);
      amx = fmaxf(amx, fmaxf(fmaxf(fabsf(m[0]), fabsf(m[1])), fmaxf(fabsf(m[2]), fabsf(m[3]))));
    } else {
      const float4 g = *reinterpret_cast<const float4*>(dy + (((size_t)n * Ho + oy) * Wo + ox) * dycs + c);
      const float ga[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float4 o;
        o.x = am[0] == t ? ga[0] : 0.f; o.y = am[1] == t ? ga[1] : 0.f;
        o.z = am[2] == t ? ga[2] : 0.f; o.w = am[3] == t ? ga[3] : 0.f;
        float* d = dx + (((size_t)n * Hi + oy * 2 + (t >> 1)) * Wi + ox * 2 + (t & 1)) * dxcs + c;
        if (accumulate) { const float4 e = *reinterpret_cast<const float4*>(d); o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w; }
        *reinterpret_cast<float4*>(d) = o;
      }
    }
  }
  amax_fold(amx, amax_out);            // (forward: the pooled map's amax for the GEMM that reads it; NULL in backward)
}

__global__ __launch_bounds__(256) void act_pattern_kernel(const float* y, int ycs, int N, int HW, int C, uint8_t* out) {
  const size_t total = (size_t)N * C * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int pix = (int)(i % HW); size_t q = i / HW;
    const int c = (int)(q % C); const int n = (int)(q / C);
    out[i] = y[((size_t)n * HW + pix) * ycs + c] > 0.f ? 1 : 0;
  }
}
// same scan as maxpool_kernel: (kh, kw) order, strict '>' (NaN wins)
__global__ __launch_bounds__(256) void pool_pattern_kernel(const float* x, int xcs, int N, int Ho, int Wo, int C, uint8_t* out) {
  const int Hi = Ho * 2, Wi = Wo * 2;
  const size_t total = (size_t)N * C * Ho * Wo;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ox = (int)(i % Wo); size_t q = i / Wo;
    const int oy = (int)(q % Ho); q /= Ho;
    const int c = (int)(q % C); const int n = (int)(q / C);
    float m = 0.f; int am = 0;
    for (int t = 0; t < 4; ++t) {
      const float v = x[(((size_t)n * Hi + oy * 2 + (t >> 1)) * Wi + ox * 2 + (t & 1)) * xcs + c];
      if (t == 0 || v > m || v != v) { m = v; am = t; }
    }
    out[i] = (uint8_t)am;
  }
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* src, int N, int C, int HW, float* dst, int dcs) {
  const size_t total = (size_t)N * C * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int pix = (int)(i % HW); size_t q = i / HW;
    const int c = (int)(q % C); const int n = (int)(q / C);
    dst[((size_t)n * HW + pix) * dcs + c] = src[i];
  }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* src, int scs, int N, int C, int HW, float* dst) {
  const size_t total = (size_t)N * C * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int pix = (int)(i % HW); size_t q = i / HW;
    const int c = (int)(q % C); const int n = (int)(q / C);
    dst[i] = src[((size_t)n * HW + pix) * scs + c];
  }
}

__constant__ uint8_t kPalette[19][3] = {
    {0, 0, 0}, {128, 0, 0}, {255, 0, 0}, {0, 85, 0}, {255, 85, 0}, {0, 0, 85}, {0, 119, 221}, {85, 85, 0},
    {0, 85, 85}, {85, 51, 0}, {52, 86, 128}, {0, 128, 0}, {0, 0, 255}, {51, 170, 221}, {0, 255, 255},
    {85, 255, 170}, {170, 255, 85}, {255, 255, 0}, {255, 170, 0}};

// MODE 0: palette decode -> uint8 NCHW rgb ; MODE 1: int32 labels
template <int MODE>
__global__ __launch_bounds__(256) void argmax_kernel(const float* x, int xcs, int N, int HW, int C, uint8_t* rgb,
                                                     int32_t* labels) {
  const size_t total = (size_t)N * HW;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const float* p = x + e * xcs;
    int best = 0; float m = p[0];
    for (int c = 1; c < C; ++c) { const float v = p[c]; if (v > m) { m = v; best = c; } }
    if (MODE == 1) {
      labels[e] = best;
    } else {
      const int n = (int)(e / HW), pix = (int)(e - (size_t)n * HW);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)
        rgb[((size_t)n * 3 + ch) * HW + pix] = best < 19 ? kPalette[best][ch] : 0;
    }
  }
}

__global__ __launch_bounds__(256) void onehot_kernel(const int32_t* labels, size_t pixels, int C, float* y, int ycs) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < pixels; e += (size_t)gridDim.x * 256) {
    const int l = labels[e];
    float* p = y + e * ycs;
    for (int c = 0; c < C; ++c) p[c] = (c == l && l != 0) ? 1.f : 0.f;   // label 0 -> all-zero vector
  }
}

}  // namespace

void roi_align_fwd(Stream& s, const TView& tex, int C, const float* rois, int R, const TView& out) {
  if (out.C < R * C || out.N != tex.N) throw Error(1, "roi_align_fwd: output view too small");
  const size_t total = (size_t)tex.N * out.H * out.W * R;
  // the wavefront-primitive form (roi_align_wave_kernel) wherever its layout conditions hold; SWN_ROI_WAVE=0 (read per launch): scalar
  const char* e = getenv("SWN_ROI_WAVE");
  if (!(e && atoi(e) == 0) && C <= 3 && tex.cs % 4 == 0 && tex.cs >= 4) {
    const size_t nwork = (size_t)tex.N * out.H * ((out.W + 63) / 64);
    hipLaunchKernelGGL(roi_align_wave_kernel, dim3((unsigned)std::min<size_t>((nwork + 3) / 4, 256 * 32)), dim3(256), 0, hs(s), tex.p, tex.cs,
                       tex.H, tex.W, C, rois, tex.N, R, out.p, out.cs, out.H, out.W);
    check_launch("roi_align_fwd (wave)");
    return;
  }
  hipLaunchKernelGGL(roi_align_kernel, dim3(egrid(total)), dim3(256), 0, hs(s), tex.p, tex.cs, tex.H, tex.W, C, rois,
                     tex.N, R, out.p, out.cs, out.H, out.W);
  check_launch("roi_align_fwd");
}
void roi_align_indices(Stream& s, const float* rois, int K, int H, int W, int PH, int PW, int32_t* idx,
                       uint8_t* valid) {
  hipLaunchKernelGGL(roi_indices_kernel, dim3(egrid((size_t)K * PH * PW)), dim3(256), 0, hs(s), rois, K, H, W, PH, PW,
                     idx, valid);
  check_launch("roi_align_indices");
}

void upsample_nearest_fwd(Stream& s, const TView& x, const TView& y, int f, float* amax_out) {
  if (y.H != x.H * f || y.W != x.W * f || x.C % 4 || y.C != x.C) throw Error(1, "upsample_nearest_fwd: shape mismatch");
  hipLaunchKernelGGL(upsample_fwd_kernel, dim3(egrid(y.pixels() * (x.C / 4))), dim3(256), 0, hs(s), x.p, x.cs, y.p, y.cs,
                     x.N, y.H, y.W, x.C, f, amax_out);
  check_launch("upsample_nearest_fwd");
}
void upsample_nearest_bwd(Stream& s, const TView& dy, const TView& dx, int f, int accumulate) {
  if (dy.H != dx.H * f || dy.W != dx.W * f || dx.C % 4 || dy.C != dx.C) throw Error(1, "upsample_nearest_bwd: shape mismatch");
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(egrid(dx.pixels() * (dx.C / 4))), dim3(256), 0, hs(s), dy.p, dy.cs, dx.p,
                     dx.cs, dx.N, dx.H, dx.W, dx.C, f, accumulate);
  check_launch("upsample_nearest_bwd");
}
void maxpool2_fwd(Stream& s, const TView& x, const TView& y, float* amax_out) {
  if (x.H != y.H * 2 || x.W != y.W * 2 || x.C % 4 || x.C != y.C) throw Error(1, "maxpool2_fwd: shape mismatch");
  hipLaunchKernelGGL(maxpool_kernel, dim3(egrid(y.pixels() * (y.C / 4))), dim3(256), 0, hs(s), x.p, x.cs, y.p, y.cs,
                     (const float*)nullptr, 0, (float*)nullptr, 0, y.N, y.H, y.W, y.C, 0, 0, amax_out);
  check_launch("maxpool2_fwd");
}
void act_pattern(Stream& s, const TView& y, uint8_t* out_nchw) {
  hipLaunchKernelGGL(act_pattern_kernel, dim3(egrid(y.pixels() * y.C)), dim3(256), 0, hs(s), y.p, y.cs, y.N, y.H * y.W, y.C, out_nchw);
  check_launch("act_pattern");
}
void pool_pattern(Stream& s, const TView& x, const TView& y, uint8_t* out_nchw) {
  if (x.H != y.H * 2 || x.W != y.W * 2 || x.C != y.C) throw Error(1, "pool_pattern: shape mismatch");
  hipLaunchKernelGGL(pool_pattern_kernel, dim3(egrid(y.pixels() * y.C)), dim3(256), 0, hs(s), x.p, x.cs, y.N, y.H, y.W, y.C, out_nchw);
  check_launch("pool_pattern");
}
void maxpool2_bwd(Stream& s, const TView& dy, const TView& x, const TView& y, const TView& dx, int accumulate) {
  hipLaunchKernelGGL(maxpool_kernel, dim3(egrid(y.pixels() * (y.C / 4))), dim3(256), 0, hs(s), x.p, x.cs, y.p, y.cs, dy.p,
                     dy.cs, dx.p, dx.cs, y.N, y.H, y.W, y.C, 1, accumulate, (float*)nullptr);
  check_launch("maxpool2_bwd");
}

__global__ __launch_bounds__(256) void affine_gather_kernel(const float* src, float* dst, int BC, int H, int W,
                                                            const double* maps, int nmaps) {
  const size_t total = (size_t)BC * H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x0 = (int)(i % W); const size_t q = i / W;
    const int y0 = (int)(q % H); const int bc = (int)(q / H);
    long long x = x0, y = y0;
    bool inside = true;
    // out = T_n(...T_1(in)): the value at p comes from in[m_1(m_2(...m_n(p)))] -- walk the chain backwards
    for (int k = nmaps - 1; k >= 0 && inside; --k) {
      const double* m = maps + ((size_t)bc * nmaps + k) * 9;
      const int kind = (int)m[0];
      long long xin = x, yin = y;
      if (kind == 1) {
        xin = ((long long)m[3] + (long long)m[2] * y + (long long)m[1] * x) >> 16;
        yin = ((long long)m[6] + (long long)m[5] * y + (long long)m[4] * x) >> 16;
      } else if (kind == 2) {
        const double xc = (double)x + 0.5, yc = (double)y + 0.5;
        const double den = m[7] * xc + m[8] * yc + 1.0;
        const double fx = (m[1] * xc + m[2] * yc + m[3]) / den, fy = (m[4] * xc + m[5] * yc + m[6]) / den;
        xin = fx < 0.0 ? -1 : (long long)(int)fx;
        yin = fy < 0.0 ? -1 : (long long)(int)fy;
      }
      inside = xin >= 0 && xin < W && yin >= 0 && yin < H;
      x = xin; y = yin;
    }
    dst[i] = inside ? src[((size_t)bc * H + y) * W + x] : 0.f;
  }
}
void affine_gather(Stream& s, const float* src, float* dst, int B, int C, int H, int W, const double* maps, int nmaps) {
  if (nmaps < 1) throw Error(1, "affine_gather: need at least one map per channel");
  hipLaunchKernelGGL(affine_gather_kernel, dim3(egrid((size_t)B * C * H * W)), dim3(256), 0, hs(s), src, dst, B * C, H, W, maps, nmaps);
  check_launch("affine_gather");
}

void nchw_to_nhwc(Stream& s, const float* src, int N, int C, int H, int W, const TView& dst) {
  if (dst.N != N || dst.H != H || dst.W != W || dst.C < C) throw Error(1, "nchw_to_nhwc: shape mismatch");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(egrid((size_t)N * C * H * W)), dim3(256), 0, hs(s), src, N, C, H * W, dst.p, dst.cs);
  check_launch("nchw_to_nhwc");
}
void nhwc_to_nchw(Stream& s, const TView& src, float* dst, int C) {
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(egrid(src.pixels() * C)), dim3(256), 0, hs(s), src.p, src.cs, src.N, C,
                     src.H * src.W, dst);
  check_launch("nhwc_to_nchw");
}
void decode_labels(Stream& s, const TView& x, int C, uint8_t* rgb) {
  hipLaunchKernelGGL(argmax_kernel<0>, dim3(egrid(x.pixels())), dim3(256), 0, hs(s), x.p, x.cs, x.N, x.H * x.W, C, rgb,
                     (int32_t*)nullptr);
  check_launch("decode_labels");
}
void argmax_labels(Stream& s, const TView& x, int C, int32_t* labels) {
  hipLaunchKernelGGL(argmax_kernel<1>, dim3(egrid(x.pixels())), dim3(256), 0, hs(s), x.p, x.cs, x.N, x.H * x.W, C,
                     (uint8_t*)nullptr, labels);
  check_launch("argmax_labels");
}
// ---- 1-channel k4 s1 p1 head conv as taps-on-N (ops.h) ----------------------------------
__global__ void head_pack_kernel(int Cip, int Npad, const float* packed, float* wt, float* wt2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 16 * Cip) return;
  const int t = i / Cip, c = i - t * Cip;
  const float v = packed[(size_t)i * Npad];            // row (t*Cip + c), column 0
  wt[c * 16 + t] = v;
  wt2[(size_t)t * Cip + c] = v;
}
__global__ void head_unpack_kernel(int Cip, int Npad, const float* dwt, float* dpacked) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 16 * Cip) return;
  const int t = i / Cip, c = i - t * Cip;
  float* d = dpacked + (size_t)i * Npad;
  d[0] = dwt[c * 16 + t];
  for (int j = 1; j < Npad; ++j) d[j] = 0.f;
}
// one thread per output pixel: 16 shifted reads of one float each
__global__ void head_gather_kernel(const float* z, int zcs, int N, int H, int W, const float* bias, float* y, int ycs) {
  const int Ho = H - 1, Wo = W - 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * Ho * Wo) return;
  const int n = i / (Ho * Wo), rem = i - n * Ho * Wo, oy = rem / Wo, ox = rem - oy * Wo;
  float acc = 0.f;
#pragma unroll
  for (int kh = 0; kh < 4; ++kh) {
    const int sy = oy - 1 + kh;
    if (sy < 0 || sy >= H) continue;
#pragma unroll
    for (int kw = 0; kw < 4; ++kw) {
      const int sx = ox - 1 + kw;
      if (sx < 0 || sx >= W) continue;
      acc += z[((size_t)(n * H + sy) * W + sx) * zcs + kh * 4 + kw];
    }
  }
  if (bias) acc += bias[0];
  *reinterpret_cast<float4*>(y + (size_t)i * ycs) = make_float4(acc, 0.f, 0.f, 0.f);
}
// one thread per (input pixel, 4 taps): dZ[q][4 kh + kw] = dY[iy+1-kh][ix+1-kw]
__global__ void head_scatter_kernel(const float* dy, int dcs, int N, int H, int W, float* dz, int zcs) {
  const int Ho = H - 1, Wo = W - 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * H * W * 4) return;
  const int kh = i & 3, q = i >> 2;
  const int n = q / (H * W), rem = q - n * H * W, iy = rem / W, ix = rem - iy * W;
  const int oy = iy + 1 - kh;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (oy >= 0 && oy < Ho) {
#pragma unroll
    for (int kw = 0; kw < 4; ++kw) {
      const int ox = ix + 1 - kw;
      if (ox >= 0 && ox < Wo) v[kw] = dy[((size_t)(n * Ho + oy) * Wo + ox) * dcs];
    }
  }
  *reinterpret_cast<float4*>(dz + (size_t)q * zcs + kh * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

void labels_to_onehot(Stream& s, const int32_t* labels, const TView& y, int C) {
  hipLaunchKernelGGL(onehot_kernel, dim3(egrid(y.pixels())), dim3(256), 0, hs(s), labels, y.pixels(), C, y.p, y.cs);
  check_launch("labels_to_onehot");
}

static void check_head(const WShape& w) {
  if (w.KH != 4 || w.KW != 4 || w.Co != 1) throw Error(1, "head conv: needs a 4x4 kernel with one output channel");
}
void head_pack(Stream& s, const WShape& w, const float* packed, float* wt, float* wt2) {
  check_head(w);
  hipLaunchKernelGGL(head_pack_kernel, dim3((16 * w.Cip + 255) / 256), dim3(256), 0, hs(s), w.Cip, w.Npad, packed, wt, wt2);
  check_launch("head_pack");
}
void head_unpack_grad(Stream& s, const WShape& w, const float* dwt, float* dpacked) {
  check_head(w);
  hipLaunchKernelGGL(head_unpack_kernel, dim3((16 * w.Cip + 255) / 256), dim3(256), 0, hs(s), w.Cip, w.Npad, dwt, dpacked);
  check_launch("head_unpack_grad");
}
void head_gather(Stream& s, const TView& z, const float* bias, const TView& y) {
  if (z.C != 16 || y.H != z.H - 1 || y.W != z.W - 1 || y.N != z.N || y.cs % 4 || z.cs % 4)
    throw Error(1, "head_gather: shape mismatch");
  const int total = y.N * y.H * y.W;
  hipLaunchKernelGGL(head_gather_kernel, dim3((total + 255) / 256), dim3(256), 0, hs(s), z.p, z.cs, z.N, z.H, z.W, bias, y.p,
                     y.cs);
  check_launch("head_gather");
}
void head_scatter(Stream& s, const TView& dy, const TView& dz) {
  if (dz.C != 16 || dy.H != dz.H - 1 || dy.W != dz.W - 1 || dy.N != dz.N || dz.cs % 4)
    throw Error(1, "head_scatter: shape mismatch");
  const int total = dz.N * dz.H * dz.W * 4;
  hipLaunchKernelGGL(head_scatter_kernel, dim3((total + 255) / 256), dim3(256), 0, hs(s), dy.p, dy.cs, dz.N, dz.H, dz.W, dz.p,
                     dz.cs);
  check_launch("head_scatter");
}

}  // namespace swn
