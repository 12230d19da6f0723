// swapnet_amd -- Winograd F(m x m, r x r) transforms for the stride-1 convolutions: the ResidualBlock
// convs (modules/layers.py:131-138 = 59 % of WarpModule's FLOPs), the VGG16 convs of PerceptualLoss
// (modules/losses/perceptual.py:26-42) and PatchGAN's k4 s1 conv (modules/discriminators.py:124-128).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A
//
//   F(4x4,3x3)  P = 36 planes per 4x4 outputs (4x fewer multiplies), points 0, +-1, +-2, inf   [H, W % 4 == 0]
//   F(2x2,3x3)  P = 16 planes per 2x2 outputs (2.25x fewer)                                     [otherwise]
//   F(3x3,4x4)  P = 36 planes per 3x3 outputs of a 4x4 filter (4x fewer), same points / same B^T
//
// The element-wise products summed over input channels are P independent GEMMs
// M[t] = V[t] (T x C) * U[t] (C x Co), executed by the MFMA implicit-GEMM kernel in batched mode
// (conv_gemm.hip); this file holds the HBM-bound transforms around them:
//   wino_input_transform   d (input patches, reflect / zero padding) -> V[P][T][C]
//   wino_filter_transform  packed W -> U[P][K][N]       (forward, or flipped+transposed for dgrad)
//   wino_output_transform  M[P][T][Co] -> y (m x m per tile, ragged edge) + bias + activation
//   wino_dy_transform      dY (m x m per tile) -> dM[P][T][Co] = A dY A^T        (weight gradient)
//   wino_filter_grad       dU[P][K][N] -> dW packed = G^T dU G
// All 16-byte vectorised along the channel axis; fp32 throughout.  F(2x2,3x3) is hand-written, the two
// 6-point forms instantiate the generic kernels at the end of the file from constant matrices.
#include "hip_util.h"

namespace swn {
namespace {

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

__device__ __forceinline__ int wsrc(int e, int ext, int pad_mode) {
  if (pad_mode == PAD_REFLECT) {
    if (e < 0) e = -e;
    else if (e >= ext) e = 2 * ext - 2 - e;
    return e;
  }
  return (e < 0 || e >= ext) ? -1 : e;
}

// one thread per (tile, 4 channels)
__global__ __launch_bounds__(256) void wino_input_kernel(const float* x, int xcs, int N, int H, int W, int C, int pad,
                                                         int pad_mode, int Th, int Tw, float* V) {
  const int C4 = C >> 2;
  const size_t T = (size_t)N * Th * Tw;
  const size_t total = T * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t tile = i / C4;
    const int c = (int)(i - tile * C4) * 4;
    const int tx = (int)(tile % Tw); size_t q = tile / Tw;
    const int ty = (int)(q % Th); const int n = (int)(q / Th);
    float4 d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int sy = wsrc(2 * ty - pad + a, H, pad_mode);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int sx = wsrc(2 * tx - pad + b, W, pad_mode);
        d[a][b] = (sy >= 0 && sx >= 0)
                      ? *reinterpret_cast<const float4*>(x + ((size_t)(n * H + sy) * W + sx) * xcs + c)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float4 t[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {          // B^T d  (rows)
      t[0][b] = f4sub(d[0][b], d[2][b]);
      t[1][b] = f4add(d[1][b], d[2][b]);
      t[2][b] = f4sub(d[2][b], d[1][b]);
      t[3][b] = f4sub(d[1][b], d[3][b]);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {          // (.) B  (columns)
      const float4 v0 = f4sub(t[a][0], t[a][2]), v1 = f4add(t[a][1], t[a][2]);
      const float4 v2 = f4sub(t[a][2], t[a][1]), v3 = f4sub(t[a][1], t[a][3]);
      float* o = V + ((size_t)(a * 4) * T + tile) * C + c;
      *reinterpret_cast<float4*>(o) = v0;
      *reinterpret_cast<float4*>(o + T * C) = v1;
      *reinterpret_cast<float4*>(o + 2 * T * C) = v2;
      *reinterpret_cast<float4*>(o + 3 * T * C) = v3;
    }
  }
}

// U[t][k][n] = (G g G^T)[t];  mode 0: g[ky][kx] = W[(ky,kx,k=ci)][n=co]
//                              mode 1: g[ky][kx] = W[(2-ky,2-kx,ci=n)][co=k]   (transposed-conv form of the input gradient)
//                              mode 2: g[ky][kx] = W[(ky,kx,ci=n)][co=k]       (= mode 0 with the channel axes swapped: the
//                                      operand of the ADJOINT form dV = dM U^T, see wino_input_adjoint)
__global__ __launch_bounds__(256) void wino_filter_kernel(WShape w, int mode, int K, int Nn, const float* packed, float* U) {
  const size_t total = (size_t)K * Nn;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i / Nn), n = (int)(i - (size_t)k * Nn);
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      float v = 0.f;
      if (mode == 0) {
        if (k < w.Cip && n < w.Npad) v = packed[((size_t)(a * 3 + b) * w.Cip + k) * w.Npad + n];
      } else if (mode == 1) {
        if (n < w.Cip && k < w.Npad) v = packed[((size_t)((2 - a) * 3 + (2 - b)) * w.Cip + n) * w.Npad + k];
      } else {
        if (n < w.Cip && k < w.Npad) v = packed[((size_t)(a * 3 + b) * w.Cip + n) * w.Npad + k];
      }
      g[a][b] = v;
    }
  float t[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {            // G g
    t[0][b] = g[0][b];
    t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
    t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
    t[3][b] = g[2][b];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {            // (.) G^T
    const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
    const float u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
    U[(size_t)(a * 4 + 0) * total + i] = u0; U[(size_t)(a * 4 + 1) * total + i] = u1;
    U[(size_t)(a * 4 + 2) * total + i] = u2; U[(size_t)(a * 4 + 3) * total + i] = u3;
  }
}

// y[2ty+a'][2tx+b'] (+)= act( (A^T m A)[a'][b'] + bias ),  m = M[.][tile][c]
__global__ __launch_bounds__(256) void wino_output_kernel(const float* M, int Cm, int N, int Th, int Tw, const float* bias,
                                                          int act, float* y, int ycs, int yH, int yW, int Cout,
                                                          int accumulate) {
  const int C4 = (Cout + 3) >> 2;
  const size_t T = (size_t)N * Th * Tw;
  const size_t total = T * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t tile = i / C4;
    const int c = (int)(i - tile * C4) * 4;
    const int tx = (int)(tile % Tw); size_t q = tile / Tw;
    const int ty = (int)(q % Th); const int n = (int)(q / Th);
    float4 m[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) m[a][b] = *reinterpret_cast<const float4*>(M + ((size_t)(a * 4 + b) * T + tile) * Cm + c);
    float4 s[2][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {          // A^T m
      s[0][b] = f4add(f4add(m[0][b], m[1][b]), m[2][b]);
      s[1][b] = f4sub(f4sub(m[1][b], m[2][b]), m[3][b]);
    }
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = make_float4(c < Cout ? bias[c] : 0.f, c + 1 < Cout ? bias[c + 1] : 0.f,
                               c + 2 < Cout ? bias[c + 2] : 0.f, c + 3 < Cout ? bias[c + 3] : 0.f);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      float4 o[2];
      o[0] = f4add(f4add(s[a][0], s[a][1]), s[a][2]);
      o[1] = f4sub(f4sub(s[a][1], s[a][2]), s[a][3]);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int oy = 2 * ty + a, ox = 2 * tx + b;
        if (oy >= yH || ox >= yW) continue;
        float4 v = f4add(o[b], bv);
        v.x = act_apply(v.x, act); v.y = act_apply(v.y, act); v.z = act_apply(v.z, act); v.w = act_apply(v.w, act);
        float* dst = y + ((size_t)(n * yH + oy) * yW + ox) * ycs + c;
        if (accumulate) v = f4add(v, *reinterpret_cast<const float4*>(dst));
        if (c + 3 < Cout) {
          *reinterpret_cast<float4*>(dst) = v;
        } else {                              // ragged channel tail (Cout not a multiple of 4)
          const float vv[4] = {v.x, v.y, v.z, v.w};
          for (int j = 0; j < 4 && c + j < Cout; ++j) dst[j] = vv[j];
        }
      }
    }
  }
}

// dM = A dY A^T : 2x2 -> 4x4   (A = [[1,0],[1,1],[1,-1],[0,-1]])
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* dy, int dcs, int N, int H, int W, int C, int Th, int Tw,
                                                      float* dM) {
  const int C4 = C >> 2;
  const size_t T = (size_t)N * Th * Tw;
  const size_t total = T * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t tile = i / C4;
    const int c = (int)(i - tile * C4) * 4;
    const int tx = (int)(tile % Tw); size_t q = tile / Tw;
    const int ty = (int)(q % Th); const int n = (int)(q / Th);
    float4 g[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int oy = 2 * ty + a, ox = 2 * tx + b;
        g[a][b] = (oy < H && ox < W) ? *reinterpret_cast<const float4*>(dy + ((size_t)(n * H + oy) * W + ox) * dcs + c)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    float4 r[4][2];                         // A g
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      r[0][b] = g[0][b];
      r[1][b] = f4add(g[0][b], g[1][b]);
      r[2][b] = f4sub(g[0][b], g[1][b]);
      r[3][b] = make_float4(-g[1][b].x, -g[1][b].y, -g[1][b].z, -g[1][b].w);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {           // (.) A^T
      const float4 v0 = r[a][0], v1 = f4add(r[a][0], r[a][1]), v2 = f4sub(r[a][0], r[a][1]);
      const float4 v3 = make_float4(-r[a][1].x, -r[a][1].y, -r[a][1].z, -r[a][1].w);
      float* o = dM + ((size_t)(a * 4) * T + tile) * C + c;
      *reinterpret_cast<float4*>(o) = v0;
      *reinterpret_cast<float4*>(o + T * C) = v1;
      *reinterpret_cast<float4*>(o + 2 * T * C) = v2;
      *reinterpret_cast<float4*>(o + 3 * T * C) = v3;
    }
  }
}

// dW[(ky,kx,ci)][co] = (G^T dU G)[ky][kx]
__global__ __launch_bounds__(256) void wino_filter_grad_kernel(WShape w, const float* dU, float* dpacked) {
  const size_t total = (size_t)w.Cip * w.Npad;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float u[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) u[a][b] = dU[(size_t)(a * 4 + b) * total + i];
  float t[3][4];                            // G^T u
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    t[0][b] = u[0][b] + 0.5f * (u[1][b] + u[2][b]);
    t[1][b] = 0.5f * (u[1][b] - u[2][b]);
    t[2][b] = 0.5f * (u[1][b] + u[2][b]) + u[3][b];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {             // (.) G
    dpacked[(size_t)(a * 3 + 0) * total + i] = t[a][0] + 0.5f * (t[a][1] + t[a][2]);
    dpacked[(size_t)(a * 3 + 1) * total + i] = 0.5f * (t[a][1] - t[a][2]);
    dpacked[(size_t)(a * 3 + 2) * total + i] = 0.5f * (t[a][1] + t[a][2]) + t[a][3];
  }
}


// ---------------------------------------------------------------------------------------
// generic F(m x m, 3x3) transforms driven by constant matrices (instantiated for m = 4)
// ---------------------------------------------------------------------------------------
struct F43 {
  static constexpr int M = 4, R = 3, A = 6;
  static constexpr float BT[6][6] = {{4, 0, -5, 0, 1, 0},  {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                     {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
  static constexpr float G[6][3] = {{1.f / 4, 0, 0},
                                    {-1.f / 6, -1.f / 6, -1.f / 6},
                                    {-1.f / 6, 1.f / 6, -1.f / 6},
                                    {1.f / 24, 1.f / 12, 1.f / 6},
                                    {1.f / 24, -1.f / 12, 1.f / 6},
                                    {0, 0, 1}};
  static constexpr float AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
};

// F(3x3, 4x4): same interpolation points (same B^T), 36 multiplies per 3x3 outputs of a 4x4 filter
// (4x fewer than direct) -- PatchGAN's k4 s1 conv (modules/discriminators.py:124-128)
struct F34 {
  static constexpr int M = 3, R = 4, A = 6;
  static constexpr float BT[6][6] = {{4, 0, -5, 0, 1, 0},  {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                     {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
  static constexpr float G[6][4] = {{1.f / 4, 0, 0, 0},
                                    {-1.f / 6, -1.f / 6, -1.f / 6, -1.f / 6},
                                    {-1.f / 6, 1.f / 6, -1.f / 6, 1.f / 6},
                                    {1.f / 24, 1.f / 12, 1.f / 6, 1.f / 3},
                                    {1.f / 24, -1.f / 12, 1.f / 6, -1.f / 3},
                                    {0, 0, 0, 1}};
  static constexpr float AT[3][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 1}};
};

__device__ __forceinline__ void f4mac(float4& s, float c, const float4& x) {
  if (c == 0.f) return;                     // folded at compile time (c is a literal after unrolling)
  s.x += c * x.x; s.y += c * x.y; s.z += c * x.z; s.w += c * x.w;
}
__device__ __forceinline__ void f1mac(float& s, float c, float x) {
  if (c == 0.f) return;
  s += c * x;
}
#define F4ZERO make_float4(0.f, 0.f, 0.f, 0.f)

// ---- planes handed to the two-plane GEMMs PRE-CUT (round 4): element-wise "pair" words --------------------------------------------
// A transform that feeds a two-plane GEMM may write each element of its planes as the 32-bit word {h | l << 16}, h = fp16(x 2^k)
// rounded to nearest, l = fp16(x 2^k - h): the fp32 slot's footprint and addressing, but the GEMM loops then assemble their MFMA
// operands with byte permutes (8 VALU per 8 elements) instead of cutting (32).  The scale must be known BEFORE the plane is written,
// so it comes from a bound: |plane| <= gain * amax(input), gain = the squared largest absolute row sum of the transform matrix,
// amax(input) from the input buffer's amax slot (engine.h buf_slots); k puts the bound in [2^14, 2^15) -- fp16 holds it with a
// factor 2 to spare, and the bound is typically 2-3 binades above the plane's true amax, which lands where the exact-amax scale
// of the cut-in-loop form (top 2^12) puts it.  Every block derives the same k; block 0 publishes it for the GEMM (kscale_out).
typedef _Float16 pf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int pair_scale_exp(const float* in_amax, float gain, int* kscale_out) {
  const int lane = threadIdx.x & 63;
  float m = fmaxf(fmaxf(in_amax[lane], in_amax[lane + 64]), fmaxf(in_amax[lane + 128], in_amax[lane + 192]));
#pragma unroll
  for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  m *= gain;
  int k = 0;
  if (m > 0.f && m <= 3.0e38f) {
    const int e = (int)((__float_as_uint(m) >> 23) & 255u) - 127;
    k = 14 - e;
    k = k < -100 ? -100 : (k > 100 ? 100 : k);
  }
  if (kscale_out && blockIdx.x == 0 && threadIdx.x == 0) *kscale_out = k;
  return k;
}
// four elements at a time with the packed conversions: 4 v_mul, 2 v_cvt_pk_f16_f32 (h), 4 v_cvt_f32_f16 + 4 v_sub (exact residuals),
// 2 v_cvt_pk_f16_f32 (l), 4 v_perm to interleave {h | l << 16} -- 20 VALU per float4
__device__ __forceinline__ float4 pair4(const float4& v, float sc) {
  const float x0 = v.x * sc, x1 = v.y * sc, x2 = v.z * sc, x3 = v.w * sc;
  const pf16x2 h01 = pf16x2{(_Float16)x0, (_Float16)x1}, h23 = pf16x2{(_Float16)x2, (_Float16)x3};
  const pf16x2 l01 = pf16x2{(_Float16)(x0 - (float)h01[0]), (_Float16)(x1 - (float)h01[1])};
  const pf16x2 l23 = pf16x2{(_Float16)(x2 - (float)h23[0]), (_Float16)(x3 - (float)h23[1])};
  const unsigned H01 = __builtin_bit_cast(unsigned, h01), H23 = __builtin_bit_cast(unsigned, h23);
  const unsigned L01 = __builtin_bit_cast(unsigned, l01), L23 = __builtin_bit_cast(unsigned, l23);
  return make_float4(__uint_as_float(__builtin_amdgcn_perm(L01, H01, 0x05040100u)), __uint_as_float(__builtin_amdgcn_perm(L01, H01, 0x07060302u)),
                     __uint_as_float(__builtin_amdgcn_perm(L23, H23, 0x05040100u)), __uint_as_float(__builtin_amdgcn_perm(L23, H23, 0x07060302u)));
}
template <int VW> using fvec = float __attribute__((ext_vector_type(VW)));
template <int VW> __device__ __forceinline__ float fvamax(const fvec<VW>& v) {
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < VW; ++i) m = fmaxf(m, fabsf(v[i]));
  return m;
}
// {h | l << 16} words of VW (2 or 4) elements, two at a time with the packed conversions
template <int VW> __device__ __forceinline__ fvec<VW> pairv(const fvec<VW>& v, float sc) {
  fvec<VW> o;
#pragma unroll
  for (int i = 0; i < VW; i += 2) {
    const float x0 = v[i] * sc, x1 = v[i + 1] * sc;
    const pf16x2 h = pf16x2{(_Float16)x0, (_Float16)x1};
    const pf16x2 l = pf16x2{(_Float16)(x0 - (float)h[0]), (_Float16)(x1 - (float)h[1])};
    const unsigned Hh = __builtin_bit_cast(unsigned, h), Ll = __builtin_bit_cast(unsigned, l);
    o[i] = __uint_as_float(__builtin_amdgcn_perm(Ll, Hh, 0x05040100u));
    o[i + 1] = __uint_as_float(__builtin_amdgcn_perm(Ll, Hh, 0x07060302u));
  }
  return o;
}
// One thread per (tile, VW channels).  The A x A intermediate lives in registers (A*A*VW floats), so VW decides the occupancy:
// 4 channels per thread is 256 VGPRs and ONE wave per SIMD for the 6-point forms, 2 channels is half of that -- and this kernel
// is pure HBM streaming, where waves in flight are what hides the latency.
template <class F, int VW>
__global__ __launch_bounds__(256) void winog_input_kernel(const float* x, int xcs, int N, int H, int W, int C, int pad,
                                                          int pad_mode, int Th, int Tw, float* V, float* amax_out,
                                                          const float* in_amax, float gain, int* kscale_out) {
  constexpr int A = F::A, M = F::M;
  typedef fvec<VW> vt;
  const int Cv = C / VW;
  const size_t T = (size_t)N * Th * Tw;
  const size_t total = T * Cv;
  float am = 0.f;
  const bool pair = in_amax != nullptr;
  const float psc = pair ? __uint_as_float((unsigned)(127 + pair_scale_exp(in_amax, gain, kscale_out)) << 23) : 1.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t tile = i / Cv;
    const int c = (int)(i - tile * Cv) * VW;
    const int tx = (int)(tile % Tw); size_t q = tile / Tw;
    const int ty = (int)(q % Th); const int n = (int)(q / Th);
    int sy[A];
#pragma unroll
    for (int a = 0; a < A; ++a) sy[a] = wsrc(M * ty - pad + a, H, pad_mode);
    vt t[A][A];
#pragma unroll
    for (int b = 0; b < A; ++b) {            // B^T d, one input column at a time
      const int sx = wsrc(M * tx - pad + b, W, pad_mode);
      vt d[A];
#pragma unroll
      for (int a = 0; a < A; ++a)
        d[a] = (sy[a] >= 0 && sx >= 0) ? *reinterpret_cast<const vt*>(x + ((size_t)(n * H + sy[a]) * W + sx) * xcs + c) : vt(0.f);
#pragma unroll
      for (int r = 0; r < A; ++r) {
        vt s = vt(0.f);
#pragma unroll
        for (int k = 0; k < A; ++k)
          if (F::BT[r][k] != 0.f) s += F::BT[r][k] * d[k];
        t[r][b] = s;
      }
    }
#pragma unroll
    for (int a = 0; a < A; ++a)              // (.) B
#pragma unroll
      for (int j = 0; j < A; ++j) {
        vt s = vt(0.f);
#pragma unroll
        for (int k = 0; k < A; ++k)
          if (F::BT[j][k] != 0.f) s += F::BT[j][k] * t[a][k];
        *reinterpret_cast<vt*>(V + ((size_t)(a * A + j) * T + tile) * C + c) = pair ? pairv<VW>(s, psc) : s;
        am = fmaxf(am, fvamax<VW>(s));
      }
  }
  if (!pair) amax_fold(am, amax_out);
}

template <class F>
__global__ __launch_bounds__(256) void winog_filter_kernel(WShape w, int mode, int K, int Nn, const float* packed, float* U) {
  constexpr int A = F::A, R = F::R;
  const size_t total = (size_t)K * Nn;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i / Nn), n = (int)(i - (size_t)k * Nn);
  float g[R][R];
#pragma unroll
  for (int a = 0; a < R; ++a)
#pragma unroll
    for (int b = 0; b < R; ++b) {
      float v = 0.f;
      if (mode == 0) {
        if (k < w.Cip && n < w.Npad) v = packed[((size_t)(a * R + b) * w.Cip + k) * w.Npad + n];
      } else if (mode == 1) {
        if (n < w.Cip && k < w.Npad) v = packed[((size_t)((R - 1 - a) * R + (R - 1 - b)) * w.Cip + n) * w.Npad + k];
      } else {
        if (n < w.Cip && k < w.Npad) v = packed[((size_t)(a * R + b) * w.Cip + n) * w.Npad + k];
      }
      g[a][b] = v;
    }
  float t[A][R];
#pragma unroll
  for (int r = 0; r < A; ++r)
#pragma unroll
    for (int b = 0; b < R; ++b) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < R; ++q) f1mac(s, F::G[r][q], g[q][b]);
      t[r][b] = s;
    }
#pragma unroll
  for (int a = 0; a < A; ++a)
#pragma unroll
    for (int j = 0; j < A; ++j) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < R; ++q) f1mac(s, F::G[j][q], t[a][q]);
      U[(size_t)(a * A + j) * total + i] = s;
    }
}

template <class F>
__global__ __launch_bounds__(256) void winog_output_kernel(const float* Mx, int Cm, int N, int Th, int Tw, const float* bias,
                                                           int act, float* y, int ycs, int yH, int yW, int Cout,
                                                           int accumulate, float* amax_out) {
  constexpr int A = F::A, M = F::M;
  const int C4 = (Cout + 3) >> 2;
  const size_t T = (size_t)N * Th * Tw;
  const size_t total = T * C4;
  float amx = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t tile = i / C4;
    const int c = (int)(i - tile * C4) * 4;
    const int tx = (int)(tile % Tw); size_t q = tile / Tw;
    const int ty = (int)(q % Th); const int n = (int)(q / Th);
    float4 s1[M][A];
#pragma unroll
    for (int b = 0; b < A; ++b) {            // A^T m, one column of planes at a time
      float4 m[A];
#pragma unroll
      for (int a = 0; a < A; ++a) m[a] = *reinterpret_cast<const float4*>(Mx + ((size_t)(a * A + b) * T + tile) * Cm + c);
#pragma unroll
      for (int r = 0; r < M; ++r) {
        float4 s = F4ZERO;
#pragma unroll
        for (int k = 0; k < A; ++k) f4mac(s, F::AT[r][k], m[k]);
        s1[r][b] = s;
      }
    }
    float4 bv = F4ZERO;
    if (bias) bv = make_float4(c < Cout ? bias[c] : 0.f, c + 1 < Cout ? bias[c + 1] : 0.f,
                               c + 2 < Cout ? bias[c + 2] : 0.f, c + 3 < Cout ? bias[c + 3] : 0.f);
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
      for (int b = 0; b < M; ++b) {
        const int oy = M * ty + a, ox = M * tx + b;
        if (oy >= yH || ox >= yW) continue;
        float4 v = F4ZERO;
#pragma unroll
        for (int k = 0; k < A; ++k) f4mac(v, F::AT[b][k], s1[a][k]);
        v = f4add(v, bv);
        v.x = act_apply(v.x, act); v.y = act_apply(v.y, act); v.z = act_apply(v.z, act); v.w = act_apply(v.w, act);
        float* dst = y + ((size_t)(n * yH + oy) * yW + ox) * ycs + c;
        if (accumulate) v = f4add(v, *reinterpret_cast<const float4*>(dst));
        if (c + 3 < Cout) {
          *reinterpret_cast<float4*>(dst) = v;
          amx = fmaxf(amx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
          for (int j = 0; j < 4 && c + j < Cout; ++j) { dst[j] = vv[j]; amx = fmaxf(amx, fabsf(vv[j])); }
        }
      }
  }
  amax_fold(amx, amax_out);            // (the layer's output with its fused activation feeds the next GEMM: round 5)
}

// dM = A dY A^T : m x m -> (m+2) x (m+2)   (A = AT^T)
template <class F, int VW>
__global__ __launch_bounds__(256) void winog_dy_kernel(const float* dy, int dcs, int N, int H, int W, int C, int Th, int Tw,
                                                       float* dM, float* amax_out, const float* in_amax, float gain, int* kscale_out) {
  constexpr int A = F::A, M = F::M;
  typedef fvec<VW> vt;
  const int Cv = C / VW;
  const size_t T = (size_t)N * Th * Tw;
  const size_t total = T * Cv;
  float am = 0.f;
  const bool pair = in_amax != nullptr;
  const float psc = pair ? __uint_as_float((unsigned)(127 + pair_scale_exp(in_amax, gain, kscale_out)) << 23) : 1.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t tile = i / Cv;
    const int c = (int)(i - tile * Cv) * VW;
    const int tx = (int)(tile % Tw); size_t q = tile / Tw;
    const int ty = (int)(q % Th); const int n = (int)(q / Th);
    vt r[A][M];
#pragma unroll
    for (int b = 0; b < M; ++b) {
      vt g[M];
#pragma unroll
      for (int a = 0; a < M; ++a) {
        const int oy = M * ty + a, ox = M * tx + b;
        g[a] = (oy < H && ox < W) ? *reinterpret_cast<const vt*>(dy + ((size_t)(n * H + oy) * W + ox) * dcs + c) : vt(0.f);
      }
#pragma unroll
      for (int p = 0; p < A; ++p) {
        vt s = vt(0.f);
#pragma unroll
        for (int a = 0; a < M; ++a)
          if (F::AT[a][p] != 0.f) s += F::AT[a][p] * g[a];
        r[p][b] = s;
      }
    }
#pragma unroll
    for (int p = 0; p < A; ++p)
#pragma unroll
      for (int j = 0; j < A; ++j) {
        vt s = vt(0.f);
#pragma unroll
        for (int b = 0; b < M; ++b)
          if (F::AT[b][j] != 0.f) s += F::AT[b][j] * r[p][b];
        *reinterpret_cast<vt*>(dM + ((size_t)(p * A + j) * T + tile) * C + c) = pair ? pairv<VW>(s, psc) : s;
        am = fmaxf(am, fvamax<VW>(s));
      }
  }
  if (!pair) amax_fold(am, amax_out);
}

// dW[(ky,kx,ci)][co] = (G^T dU G)[ky][kx]
template <class F>
__global__ __launch_bounds__(256) void winog_filter_grad_kernel(WShape w, const float* dU, float* dpacked) {
  constexpr int A = F::A, R = F::R;
  const size_t total = (size_t)w.Cip * w.Npad;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float t[R][A];
#pragma unroll
  for (int b = 0; b < A; ++b) {
    float u[A];
#pragma unroll
    for (int a = 0; a < A; ++a) u[a] = dU[(size_t)(a * A + b) * total + i];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float s = 0.f;
#pragma unroll
      for (int a = 0; a < A; ++a) f1mac(s, F::G[a][r], u[a]);
      t[r][b] = s;
    }
  }
#pragma unroll
  for (int a = 0; a < R; ++a)
#pragma unroll
    for (int j = 0; j < R; ++j) {
      float s = 0.f;
#pragma unroll
      for (int b = 0; b < A; ++b) f1mac(s, F::G[b][j], t[a][b]);
      dpacked[(size_t)(a * R + j) * total + i] = s;
    }
}


// ---------------------------------------------------------------------------------------
// Adjoint of the input transform: the input gradient of a Winograd convolution in its own tiling.
//   forward   V_t = BT d_t BT^T ,  d_t[a][b] = x[src(m ty - pad + a)][src(m tx - pad + b)]      (zero / reflect padding folded in)
//   backward  dx[y][x] = sum over (t, a, b) with src(.) = (y, x) of (BT^T dV_t BT)[a][b]
// The transposed-convolution form used until round 2 ran the same GEMMs over the (for reflect padding: padded) INPUT grid --
// 25 tiles per 16x16 map where the forward pass has 16 -- and needed its own input transform of dY; here dV = dM U^T reuses
// dM = A dY A^T (the weight gradient's operand) on the forward tiling: 36 % fewer GEMM rows on the resblock convs.
// Two kernels: the per-tile patch BT^T dV BT in place, then a gather per input pixel (fixed summation order, no atomics).
// ---------------------------------------------------------------------------------------
struct F23 {
  static constexpr int M = 2, R = 3, A = 4;
  static constexpr float BT[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
};

template <class F>
__global__ __launch_bounds__(256) void winog_patch_kernel(float* V, int C, size_t T) {
  constexpr int A = F::A;
  const int C4 = C >> 2;
  const size_t total = T * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t tile = i / C4;
    const int c = (int)(i - tile * C4) * 4;
    float4 t[A][A];
#pragma unroll
    for (int b = 0; b < A; ++b) {            // BT^T v, one column of planes at a time
      float4 v[A];
#pragma unroll
      for (int a = 0; a < A; ++a) v[a] = *reinterpret_cast<const float4*>(V + ((size_t)(a * A + b) * T + tile) * C + c);
#pragma unroll
      for (int r = 0; r < A; ++r) {
        float4 s = F4ZERO;
#pragma unroll
        for (int k = 0; k < A; ++k) f4mac(s, F::BT[k][r], v[k]);
        t[r][b] = s;
      }
    }
#pragma unroll
    for (int a = 0; a < A; ++a)              // (.) BT
#pragma unroll
      for (int j = 0; j < A; ++j) {
        float4 s = F4ZERO;
#pragma unroll
        for (int k = 0; k < A; ++k) f4mac(s, F::BT[k][j], t[a][k]);
        *reinterpret_cast<float4*>(V + ((size_t)(a * A + j) * T + tile) * C + c) = s;
      }
  }
}

// candidates e (tile coordinates, -pad <= e <= emax) with src(e) == y: y itself and its single-bounce reflections
__device__ __forceinline__ int adj_sources(int y, int ext, int pad, int pad_mode, int emax, int (&e)[3]) {
  int n = 0;
  if (y <= emax) e[n++] = y;
  if (pad_mode == PAD_REFLECT) {
    if (y >= 1 && -y >= -pad) e[n++] = -y;
    const int r = 2 * ext - 2 - y;
    if (r >= ext && r <= emax) e[n++] = r;
  }
  return n;
}

__global__ __launch_bounds__(256) void wino_fold_kernel(const float* P, int C, int N, int H, int W, int m, int A, int pad,
                                                        int pad_mode, int Th, int Tw, float* dx, int dcs, int accumulate) {
  const int C4 = C >> 2;
  const size_t T = (size_t)N * Th * Tw;
  const size_t total = (size_t)N * H * W * C4;
  const int emax_y = m * (Th - 1) - pad + A - 1, emax_x = m * (Tw - 1) - pad + A - 1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t px = i / C4;
    const int c = (int)(i - px * C4) * 4;
    const int n = (int)(px / ((size_t)H * W));
    const int rem = (int)(px - (size_t)n * H * W);
    const int y = rem / W, x = rem - y * W;
    int ey[3], ex[3];
    const int ny = adj_sources(y, H, pad, pad_mode, emax_y, ey), nx = adj_sources(x, W, pad, pad_mode, emax_x, ex);
    float4 acc = F4ZERO;
    for (int iy = 0; iy < ny; ++iy) {
      const int qy = ey[iy] + pad;                                   // = m ty + a
      const int ty1 = min(qy / m, Th - 1);
      for (int ty = ty1; ty >= 0 && qy - m * ty < A; --ty) {
        const int a = qy - m * ty;
        for (int ix = 0; ix < nx; ++ix) {
          const int qx = ex[ix] + pad;
          const int tx1 = min(qx / m, Tw - 1);
          for (int tx = tx1; tx >= 0 && qx - m * tx < A; --tx) {
            const int b = qx - m * tx;
            const size_t tile = ((size_t)n * Th + ty) * Tw + tx;
            const float4 v = *reinterpret_cast<const float4*>(P + ((size_t)(a * A + b) * T + tile) * C + c);
            acc = f4add(acc, v);
          }
        }
      }
    }
    float* d = dx + px * dcs + c;
    if (accumulate) acc = f4add(acc, *reinterpret_cast<const float4*>(d));
    *reinterpret_cast<float4*>(d) = acc;
  }
}


// ---------------------------------------------------------------------------------------
// Strided Winograd F(4x4, 2x2) for the k4 s2 p1 convolutions (UNetDown, PatchGAN) and their transposes (UNetUp)
// (modules/layers.py:15,31; modules/discriminators.py:110-120; modules/pix2pix_modules.py:216-246), round 3.
// A 4x4 stride-2 convolution is the sum of four 2x2 STRIDE-1 convolutions over the polyphase components of its input
// (rows 2q - 1 + s, columns 2q' - 1 + t, s, t in {0, 1}) with the filters g_st[a][b] = w[2a + s][2b + t].  Each of them takes
// the 5-point Winograd form F(4x4, 2x2) (points 0, +-1, 1/2, inf: 25 multiplies per 16 outputs instead of 64), and in the
// transformed domain the four phases are simply four times the input channels of ONE batched GEMM:
//   fine -> coarse  (conv forward, convT input gradient):  V[25][T][4 Cf] (this file) -> M = V U -> coarse = A^T M A
//   coarse -> fine  (convT forward, conv input gradient):  dM = A c A^T -> dV = dM U^T [25][T][4 Cf] -> adjoint of the
//                                                          polyphase input transform (patch + gather, no atomics)
//   weight gradient:                                       dU = V^T dM -> dW = G^T dU G per phase
// 2.56x fewer multiplies than the direct implicit GEMM; fp32 error against float64 ~1.7e-6 (between the direct kernels and
// F(4x4,3x3)).  Used where the activation side dominates (engine.cpp: coarse channels >= 256, weights <= 4 M).
// ---------------------------------------------------------------------------------------
struct F42 {
  static constexpr int M = 4, R = 2, A = 5;
  static constexpr float BT[5][5] = {{0.5f, -1, -0.5f, 1, 0}, {0, -0.5f, 0.5f, 1, 0}, {0, 0.5f, -1.5f, 1, 0}, {0, -1, 0, 1, 0},
                                     {0, 0.5f, -1, -0.5f, 1}};
  static constexpr float G[5][2] = {{2, 0}, {1, 1}, {-1.f / 3, 1.f / 3}, {-8.f / 3, -4.f / 3}, {0, 1}};
  static constexpr float AT[4][5] = {{1, 1, 1, 1, 0}, {0, 1, -1, 0.5f, 0}, {0, 1, 1, 0.25f, 0}, {0, 1, -1, 0.125f, 1}};
};

// V[(a*5+j)][tile][q*C + c] = (BT d_q BT^T)[a][j],  d_q[i][j] = x[2 (4 ty + i) - 1 + s][2 (4 tx + j) - 1 + t],  q = 2 s + t
template <int VW>
__global__ __launch_bounds__(256) void wino_s2_input_kernel(const float* x, int xcs, int N, int H, int W, int C, int Th, int Tw,
                                                            float* V, float* amax_out, const float* in_amax, float gain, int* kscale_out) {
  constexpr int A = 5;
  typedef fvec<VW> vt;
  const int C4 = C / VW;
  const size_t T = (size_t)N * Th * Tw;
  const size_t total = T * 4 * C4;
  const int CV = 4 * C;
  float am = 0.f;
  const bool pair = in_amax != nullptr;
  const float psc = pair ? __uint_as_float((unsigned)(127 + pair_scale_exp(in_amax, gain, kscale_out)) << 23) : 1.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C4) * VW; size_t r = i / C4;
    const int q = (int)(r & 3); const size_t tile = r >> 2;
    const int s = q >> 1, t = q & 1;
    const int tx = (int)(tile % Tw); size_t u = tile / Tw;
    const int ty = (int)(u % Th); const int n = (int)(u / Th);
    int sy[A];
#pragma unroll
    for (int a = 0; a < A; ++a) { const int e = 2 * (4 * ty + a) - 1 + s; sy[a] = (e >= 0 && e < H) ? e : -1; }
    vt tt[A][A];
#pragma unroll
    for (int b = 0; b < A; ++b) {
      const int e = 2 * (4 * tx + b) - 1 + t;
      const int sx = (e >= 0 && e < W) ? e : -1;
      vt d[A];
#pragma unroll
      for (int a = 0; a < A; ++a)
        d[a] = (sy[a] >= 0 && sx >= 0) ? *reinterpret_cast<const vt*>(x + ((size_t)(n * H + sy[a]) * W + sx) * xcs + c) : vt(0.f);
#pragma unroll
      for (int rr = 0; rr < A; ++rr) {
        vt acc = vt(0.f);
#pragma unroll
        for (int k = 0; k < A; ++k)
          if (F42::BT[rr][k] != 0.f) acc += F42::BT[rr][k] * d[k];
        tt[rr][b] = acc;
      }
    }
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
      for (int j = 0; j < A; ++j) {
        vt acc = vt(0.f);
#pragma unroll
        for (int k = 0; k < A; ++k)
          if (F42::BT[j][k] != 0.f) acc += F42::BT[j][k] * tt[a][k];
        *reinterpret_cast<vt*>(V + ((size_t)(a * A + j) * T + tile) * CV + q * C + c) = pair ? pairv<VW>(acc, psc) : acc;
        am = fmaxf(am, fvamax<VW>(acc));
      }
  }
  if (!pair) amax_fold(am, amax_out);
}

// gather side of the adjoint: P[25][T][4 C] holds the patches BT^T dV BT; fine pixel (r, cc) of phase (s, t) sits at patch
// entry (i, j) = (qy - 4 ty, qx - 4 tx) of the tiles that cover phase coordinates qy = (r + 1 - s) / 2, qx likewise
__global__ __launch_bounds__(256) void wino_s2_fold_kernel(const float* P, int C, int N, int H, int W, int Th, int Tw, const float* bias,
                                                           float* dx, int dcs, int accumulate) {
  constexpr int A = 5, M = 4;
  const int C4 = C >> 2, CV = 4 * C;
  const size_t T = (size_t)N * Th * Tw;
  const size_t total = (size_t)N * H * W * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t px = i / C4;
    const int c = (int)(i - px * C4) * 4;
    const int n = (int)(px / ((size_t)H * W));
    const int rem = (int)(px - (size_t)n * H * W);
    const int r = rem / W, cc = rem - r * W;
    const int s = (r + 1) & 1, t = (cc + 1) & 1;
    const int qy = (r + 1 - s) >> 1, qx = (cc + 1 - t) >> 1;
    float4 acc = F4ZERO;
    for (int ty = min(qy / M, Th - 1); ty >= 0 && qy - M * ty < A; --ty) {
      const int a = qy - M * ty;
      for (int tx = min(qx / M, Tw - 1); tx >= 0 && qx - M * tx < A; --tx) {
        const int b = qx - M * tx;
        const size_t tile = ((size_t)n * Th + ty) * Tw + tx;
        acc = f4add(acc, *reinterpret_cast<const float4*>(P + ((size_t)(a * A + b) * T + tile) * CV + (2 * s + t) * C + c));
      }
    }
    if (bias) { acc.x += bias[c]; acc.y += bias[c + 1]; acc.z += bias[c + 2]; acc.w += bias[c + 3]; }
    float* d = dx + px * dcs + c;
    if (accumulate) acc = f4add(acc, *reinterpret_cast<const float4*>(d));
    *reinterpret_cast<float4*>(d) = acc;
  }
}

// the weight of the k4 s2 p1 convolution fine -> coarse that the layer is (conv) or is the adjoint of (convT), from the packed
// arena layouts of ops.h: conv [(kh*4+kw)*Cip + ci][co]; convT phase blocks p = a*2+b of [(dy*2+dx)*Cip + ci][co],
// tap (ky, kx) = (3 - a - 2 dy, 3 - b - 2 dx), with W_convT[ci][co][ky][kx] = Wc[ky][kx][fine = co][coarse = ci]
__device__ __forceinline__ size_t s2_widx(const WShape& w, int kh, int kw, int cf, int cc) {
  if (w.kind == WK_CONV) return ((size_t)(kh * 4 + kw) * w.Cip + cf) * w.Npad + cc;
  const int a = (3 - kh) & 1, dy = (3 - kh - a) >> 1, b = (3 - kw) & 1, dx = (3 - kw - b) >> 1;
  return (size_t)(a * 2 + b) * 4 * w.Cip * w.Npad + ((size_t)(dy * 2 + dx) * w.Cip + cc) * w.Npad + cf;
}

// mode 0: U[p][q*Cf + cf][cc] (threads: cc fastest); mode 1: U[p][cc][q*Cf + cf] (threads: cf fastest)
__global__ __launch_bounds__(256) void wino_s2_filter_kernel(WShape w, int mode, int Cf, int Cc, const float* packed, float* U) {
  constexpr int A = 5;
  const size_t total = (size_t)Cf * Cc;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cf = mode == 0 ? (int)(i / Cc) : (int)(i % Cf), cc = mode == 0 ? (int)(i % Cc) : (int)(i / Cf);
  const size_t plane = (size_t)4 * Cf * Cc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int s = q >> 1, t = q & 1;
    float g[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) g[a][b] = packed[s2_widx(w, 2 * a + s, 2 * b + t, cf, cc)];
    float tt[A][2];
#pragma unroll
    for (int r = 0; r < A; ++r)
#pragma unroll
      for (int b = 0; b < 2; ++b) tt[r][b] = F42::G[r][0] * g[0][b] + F42::G[r][1] * g[1][b];
    const size_t o = mode == 0 ? ((size_t)(q * Cf + cf) * Cc + cc) : ((size_t)cc * 4 * Cf + q * Cf + cf);
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
      for (int j = 0; j < A; ++j) U[(size_t)(a * A + j) * plane + o] = tt[a][0] * F42::G[j][0] + tt[a][1] * F42::G[j][1];
  }
}

// dW[2a+s][2b+t][cf][cc] = (G^T dU_q G)[a][b],  dU[p][q*Cf + cf][cc]
__global__ __launch_bounds__(256) void wino_s2_filter_grad_kernel(WShape w, int Cf, int Cc, const float* dU, float* dpacked) {
  constexpr int A = 5;
  const size_t total = (size_t)Cf * Cc;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cf = (int)(i / Cc), cc = (int)(i % Cc);
  const size_t plane = (size_t)4 * Cf * Cc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int s = q >> 1, t = q & 1;
    const size_t o = (size_t)(q * Cf + cf) * Cc + cc;
    float tt[2][A];
#pragma unroll
    for (int b = 0; b < A; ++b) {
      float u[A];
#pragma unroll
      for (int a = 0; a < A; ++a) u[a] = dU[(size_t)(a * A + b) * plane + o];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < A; ++a) f1mac(acc, F42::G[a][r], u[a]);
        tt[r][b] = acc;
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < A; ++k) f1mac(acc, F42::G[k][b], tt[a][k]);
        dpacked[s2_widx(w, 2 * a + s, 2 * b + t, cf, cc)] = acc;
      }
  }
}


// ---- filter transform straight into the PRE-CUT operand layout of the ring kernel (conv_gemm.hip conv_fwd_pc_kernel) ----------
// U[p][k][n] as above (mode 0: k = ci, n = co; mode 2: k = co, n = ci), but each thread produces the 8 consecutive k of one
// (k / 8, column) entry, cuts them into the three bf16 planes and writes the 16-byte entries of
//   out[p][stage = k / 16][tile_n][kq 2][plane 3][pos BN][8 k],  pos = (n % NB) * 32 + n / NB  (NB = BN / 32).
// The fp32 U is never materialised: 6 B instead of 4 B written per element, nothing re-read by a separate conv_precut pass.
typedef unsigned wu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void cut8_store(const float (&u)[8], wu32x4* o, size_t e0, size_t plane_stride) {
  unsigned hi[4], mid[4], lo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = u[2 * j], x1 = u[2 * j + 1];
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    hi[j] = (u0 >> 16) | (u1 & 0xffff0000u);
    mid[j] = (v0 >> 16) | (v1 & 0xffff0000u);
    lo[j] = (__float_as_uint(s0) >> 16) | (__float_as_uint(s1) & 0xffff0000u);
  }
  o[e0] = wu32x4{hi[0], hi[1], hi[2], hi[3]};
  o[e0 + plane_stride] = wu32x4{mid[0], mid[1], mid[2], mid[3]};
  o[e0 + 2 * plane_stride] = wu32x4{lo[0], lo[1], lo[2], lo[3]};
}

// two-plane fp16 form (conv_gemm.hip): u * 2^kB as h + l
typedef _Float16 wf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cut8_store_h(const float (&u)[8], float sb, wu32x4* o, size_t e0, size_t plane_stride, bool low_plane) {
  unsigned hi[4], lo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = u[2 * j] * sb, x1 = u[2 * j + 1] * sb;
    const wf16x2 hh = wf16x2{(_Float16)x0, (_Float16)x1};
    hi[j] = __builtin_bit_cast(unsigned, hh);
    lo[j] = __builtin_bit_cast(unsigned, wf16x2{(_Float16)(x0 - (float)hh[0]), (_Float16)(x1 - (float)hh[1])});
  }
  o[e0] = wu32x4{hi[0], hi[1], hi[2], hi[3]};
  if (low_plane) o[e0 + plane_stride] = wu32x4{lo[0], lo[1], lo[2], lo[3]};
}
__device__ __forceinline__ int wino_scale_exp(const float* part, int lane, int top) {      // as conv_gemm.hip scale_exp(amax256())
  float m = fmaxf(fmaxf(part[lane], part[lane + 64]), fmaxf(part[lane + 128], part[lane + 192]));
#pragma unroll
  for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (!(m > 0.f) || m > 3.0e38f) return 0;
  const int e = (int)((__float_as_uint(m) >> 23) & 255u) - 127;
  const int k = top - 1 - e;
  return k < -100 ? -100 : (k > 100 ? 100 : k);
}

template <class F>
__global__ __launch_bounds__(256) void winog_filter_pc_kernel(WShape w, int mode, int K, int Nn, int BN, const float* packed,
                                                              unsigned short* out, size_t panel_elems, const float* wamax, int planes) {
  constexpr int A = F::A, R = F::R;
  const int NBc = BN / 32, tiles_n = (Nn + BN - 1) / BN;
  const size_t total = (size_t)(K / 8) * tiles_n * BN;
  const size_t i = ((size_t)(blockIdx.x / (8 * A)) * 8 + blockIdx.x % 8) * 256 + threadIdx.x;   // slab: see the row index below
  const int PL = planes;               // 3 bf16 planes (wamax == NULL), 2 or 1 fp16 planes of the scaled filter
  int kB = 0;
  if (wamax) kB = wino_scale_exp(wamax, threadIdx.x & 63, 10);       // PC_TOP_B; |G g G^T| <= |g|max for both 6-point forms
  if (i >= total) return;
  // consecutive threads take consecutive COLUMNS (coalesced reads of the packed weights: a thread per operand position read
  // every fourth column, a quarter of each sector); the operand position of column nl is pos = (nl % NB) * 32 + nl / NB.
  // The plane ROW a = (blockIdx.x % 8A) / 8: a thread builds (G g)[a][.] for its 8 k and writes the A planes (a, 0..A-1).
  // The A rows of one 256-thread slab of taps are blocks b, b + 8, ..., b + 8(A-1): dispatched together and, workgroups going
  // round-robin over the 8 XCDs, onto the SAME XCD, so the re-reads (9 floats per (k, n), A times) meet in that L2 -- with the
  // row as the slow grid index each of the A passes streamed the 38 MB of a 1024 x 1024 layer again.
  const int nl = (int)(i % BN); const size_t q = i / BN;
  const int tn = (int)(q % tiles_n), kq = (int)(q / tiles_n);
  const int n = tn * BN + nl;
  const int pos = (nl % NBc) * 32 + nl / NBc;
  const int a = (int)(blockIdx.x % (8 * A)) / 8;
  float Ga[R];
#pragma unroll
  for (int qq = 0; qq < R; ++qq) Ga[qq] = F::G[a][qq];
  // t[kk][b] = (G g)[a][b] for the 8 k of this entry
  float t[8][R];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk)
#pragma unroll
    for (int b = 0; b < R; ++b) t[kk][b] = 0.f;
  if (mode == 0) {
    if (n < w.Npad) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int k = kq * 8 + kk;
        if (k < w.Cip) {
#pragma unroll
          for (int qq = 0; qq < R; ++qq)
#pragma unroll
            for (int b = 0; b < R; ++b) t[kk][b] += Ga[qq] * packed[((size_t)(qq * R + b) * w.Cip + k) * w.Npad + n];
        }
      }
    }
  } else if (n < w.Cip && kq * 8 < w.Npad) {       // transposed forms: the 8 k are consecutive floats of one row (Npad % 8 == 0)
#pragma unroll
    for (int qq = 0; qq < R; ++qq)
#pragma unroll
      for (int b = 0; b < R; ++b) {
        const int tap = mode == 1 ? (R - 1 - qq) * R + (R - 1 - b) : qq * R + b;
        const float4* src = reinterpret_cast<const float4*>(packed + ((size_t)tap * w.Cip + n) * w.Npad + kq * 8);
        const float4 g0 = src[0], g1 = src[1];
        t[0][b] += Ga[qq] * g0.x; t[1][b] += Ga[qq] * g0.y; t[2][b] += Ga[qq] * g0.z; t[3][b] += Ga[qq] * g0.w;
        t[4][b] += Ga[qq] * g1.x; t[5][b] += Ga[qq] * g1.y; t[6][b] += Ga[qq] * g1.z; t[7][b] += Ga[qq] * g1.w;
      }
  }
  const size_t e0 = ((((size_t)(kq >> 1) * tiles_n + tn) * 2 + (kq & 1)) * PL) * BN + pos;     // in 16-byte entries
  const float sb = __uint_as_float((unsigned)(127 + kB) << 23);
#pragma unroll
  for (int j = 0; j < A; ++j) {
    float u[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float s = 0.f;
#pragma unroll
      for (int qq = 0; qq < R; ++qq) f1mac(s, F::G[j][qq], t[kk][qq]);
      u[kk] = s;
    }
    unsigned short* panel = out + (size_t)(a * A + j) * panel_elems;
    if (wamax) {
      cut8_store_h(u, sb, reinterpret_cast<wu32x4*>(panel), e0, (size_t)BN, PL == 2);
      if (i == 0) *reinterpret_cast<int*>(panel + (size_t)(K / 16) * tiles_n * 2 * PL * BN * 8) = kB;  // trailer of every panel
    } else {
      cut8_store(u, reinterpret_cast<wu32x4*>(panel), e0, (size_t)BN);
    }
  }
}


// ---------------------------------------------------------------------------------------
// The folded tail conv (Upsample x2 + ZeroPad(1,0,1,0) + Conv k4 p1, swapnet_modules.py:85-90; ops.h tail_fold_weights) in
// Winograd form.  Its four sub-pixel phases are (2+a) x (2+b)-tap stride-1 convolutions over the SAME un-upsampled input with the
// same top-left offset, i.e. four F(4x4,3x3) convolutions (the 2-tap axes zero-extended) that share ONE input transform; their
// transformed filters sit side by side on the N axis of one batched GEMM (N = 4 Npad = 80 for the 19-channel output).  36
// multiplies per 4x4 input positions and output channel pair instead of the 25 x 16 of the folded direct form (2.8x fewer), on
// the 32-wide bf16-split MFMA tiles instead of the 4x4x1 f32 blocks the N = 19 direct kernels are confined to.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ size_t tailw_fold_off(int Cip, int Npad, int ph) {
  const int pre = ph == 0 ? 0 : (ph == 1 ? 4 : (ph == 2 ? 10 : 16));
  return (size_t)pre * Cip * Npad;
}
// U[p][ci][ph * Npad + co] = (G g_ph G^T)[p],  g_ph[r][c] = folded_ph[(r * (2 + b) + c) * Cip + ci][co]  (0 beyond the phase's taps)
__global__ __launch_bounds__(256) void tailw_filter_kernel(int Cip, int Npad, const float* folded, float* U) {
  constexpr int A = 6, R = 3;
  const int N4 = 4 * Npad;
  const size_t total = (size_t)Cip * N4;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int ci = (int)(i / N4), n = (int)(i % N4);
  const int ph = n / Npad, co = n - ph * Npad, a = ph >> 1, b = ph & 1;
  const float* f = folded + tailw_fold_off(Cip, Npad, ph);
  float g[R][R];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < R; ++c) g[r][c] = (r < 2 + a && c < 2 + b) ? f[((size_t)(r * (2 + b) + c) * Cip + ci) * Npad + co] : 0.f;
  float t[A][R];
#pragma unroll
  for (int r = 0; r < A; ++r)
#pragma unroll
    for (int c = 0; c < R; ++c) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < R; ++q) f1mac(s, F43::G[r][q], g[q][c]);
      t[r][c] = s;
    }
#pragma unroll
  for (int r = 0; r < A; ++r)
#pragma unroll
    for (int j = 0; j < A; ++j) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < R; ++q) f1mac(s, F43::G[j][q], t[r][q]);
      U[(size_t)(r * A + j) * total + i] = s;
    }
}
// dfolded_ph[(r * (2 + b) + c) * Cip + ci][co] = (G^T dU_ph G)[r][c] for the phase's taps
__global__ __launch_bounds__(256) void tailw_filter_grad_kernel(int Cip, int Npad, const float* dU, float* dfolded) {
  constexpr int A = 6, R = 3;
  const int N4 = 4 * Npad;
  const size_t total = (size_t)Cip * N4;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int ci = (int)(i / N4), n = (int)(i % N4);
  const int ph = n / Npad, co = n - ph * Npad, a = ph >> 1, b = ph & 1;
  float t[R][A];
#pragma unroll
  for (int j = 0; j < A; ++j) {
    float u[A];
#pragma unroll
    for (int r = 0; r < A; ++r) u[r] = dU[(size_t)(r * A + j) * total + i];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < A; ++r) f1mac(s, F43::G[r][q], u[r]);
      t[q][j] = s;
    }
  }
  float* f = dfolded + tailw_fold_off(Cip, Npad, ph);
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < R; ++c) {
      if (r >= 2 + a || c >= 2 + b) continue;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < A; ++j) f1mac(s, F43::G[j][c], t[r][j]);
      f[((size_t)(r * (2 + b) + c) * Cip + ci) * Npad + co] = s;
    }
}
// y[2 (4 ty + i) + a][2 (4 tx + j) + b][co] = act((A^T M_ph A)[i][j] + bias),  M[p][tile][ph * Npad + co]
__global__ __launch_bounds__(256) void tailw_output_kernel(const float* Mx, int N, int Th, int Tw, int Npad, const float* bias, int act,
                                                           float* y, int ycs, int yH, int yW, int Cout) {
  constexpr int A = 6, M = 4;
  const int C4 = Npad >> 2, CM = 4 * Npad;
  const size_t T = (size_t)N * Th * Tw;
  const size_t total = T * 4 * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4; size_t r = i / C4;
    const int ph = (int)(r & 3); const size_t tile = r >> 2;
    const int a = ph >> 1, b = ph & 1;
    const int tx = (int)(tile % Tw); size_t u = tile / Tw;
    const int ty = (int)(u % Th); const int n = (int)(u / Th);
    float4 s1[M][A];
#pragma unroll
    for (int j = 0; j < A; ++j) {
      float4 m[A];
#pragma unroll
      for (int k = 0; k < A; ++k) m[k] = *reinterpret_cast<const float4*>(Mx + ((size_t)(k * A + j) * T + tile) * CM + ph * Npad + c);
#pragma unroll
      for (int q = 0; q < M; ++q) {
        float4 s = F4ZERO;
#pragma unroll
        for (int k = 0; k < A; ++k) f4mac(s, F43::AT[q][k], m[k]);
        s1[q][j] = s;
      }
    }
    float4 bv = F4ZERO;
    if (bias) bv = make_float4(c < Cout ? bias[c] : 0.f, c + 1 < Cout ? bias[c + 1] : 0.f, c + 2 < Cout ? bias[c + 2] : 0.f,
                               c + 3 < Cout ? bias[c + 3] : 0.f);
#pragma unroll
    for (int q = 0; q < M; ++q)
#pragma unroll
      for (int jj = 0; jj < M; ++jj) {
        const int oy = 2 * (M * ty + q) + a, ox = 2 * (M * tx + jj) + b;
        if (oy >= yH || ox >= yW) continue;
        float4 v = F4ZERO;
#pragma unroll
        for (int k = 0; k < A; ++k) f4mac(v, F43::AT[jj][k], s1[q][k]);
        v = f4add(v, bv);
        v.x = act_apply(v.x, act); v.y = act_apply(v.y, act); v.z = act_apply(v.z, act); v.w = act_apply(v.w, act);
        float* dst = y + ((size_t)(n * yH + oy) * yW + ox) * ycs + c;
        if (c + 3 < Cout) {
          *reinterpret_cast<float4*>(dst) = v;
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
          for (int e = 0; e < 4 && c + e < Cout; ++e) dst[e] = vv[e];
        }
      }
  }
}
// dM[p][tile][ph * Npad + c] = (A g_ph A^T)[p],  g_ph[i][j] = dy[2 (4 ty + i) + a][2 (4 tx + j) + b][c]
__global__ __launch_bounds__(256) void tailw_dy_kernel(const float* dy, int dcs, int N, int yH, int yW, int Th, int Tw, int Npad,
                                                       float* dM, float* amax_out, const float* in_amax, float gain, int* kscale_out) {
  constexpr int A = 6, M = 4;
  const int C4 = Npad >> 2, CM = 4 * Npad;
  const size_t T = (size_t)N * Th * Tw;
  const size_t total = T * 4 * C4;
  float am = 0.f;
  const bool pair = in_amax != nullptr;
  const float psc = pair ? __uint_as_float((unsigned)(127 + pair_scale_exp(in_amax, gain, kscale_out)) << 23) : 1.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4; size_t r = i / C4;
    const int ph = (int)(r & 3); const size_t tile = r >> 2;
    const int a = ph >> 1, b = ph & 1;
    const int tx = (int)(tile % Tw); size_t u = tile / Tw;
    const int ty = (int)(u % Th); const int n = (int)(u / Th);
    float4 rr[A][M];
#pragma unroll
    for (int jj = 0; jj < M; ++jj) {
      float4 g[M];
#pragma unroll
      for (int q = 0; q < M; ++q) {
        const int oy = 2 * (M * ty + q) + a, ox = 2 * (M * tx + jj) + b;
        g[q] = (oy < yH && ox < yW) ? *reinterpret_cast<const float4*>(dy + ((size_t)(n * yH + oy) * yW + ox) * dcs + c) : F4ZERO;
      }
#pragma unroll
      for (int p = 0; p < A; ++p) {
        float4 s = F4ZERO;
#pragma unroll
        for (int q = 0; q < M; ++q) f4mac(s, F43::AT[q][p], g[q]);
        rr[p][jj] = s;
      }
    }
#pragma unroll
    for (int p = 0; p < A; ++p)
#pragma unroll
      for (int j = 0; j < A; ++j) {
        float4 s = F4ZERO;
#pragma unroll
        for (int jj = 0; jj < M; ++jj) f4mac(s, F43::AT[jj][j], rr[p][jj]);
        *reinterpret_cast<float4*>(dM + ((size_t)(p * A + j) * T + tile) * CM + ph * Npad + c) = pair ? pair4(s, psc) : s;
        am = fmaxf(am, f4amax(s));
      }
  }
  if (!pair) amax_fold(am, amax_out);
}

inline unsigned wgrid(size_t total) { return (unsigned)std::min<size_t>(std::max<size_t>((total + 255) / 256, 1), 256 * 32); }

}  // namespace

// variant id: 0 = F(2,3) hand-written, 1 = F(4,3), 2 = F(3,4)
static int variant(int m, int r) {
  if (m == 2 && r == 3) return 0;
  if (m == 4 && r == 3) return 1;
  if (m == 3 && r == 4) return 2;
  if (m == 4 && r == 2) return 3;            // strided form: output / dy / patch transforms only
  throw Error(1, "winograd: supported forms are F(2,3), F(4,3), F(3,4) and the strided F(4,2)");
}
// channels per thread of the 6-point input / dY transforms: two (round 4: the register-friendly form; the four-channel
// instantiations of round 3 and their switch are gone)
static constexpr int wino_vec_width() { return 2; }
// gains of the pair-form bound (squared largest absolute row sum of the transform matrix)
static float input_gain(int v) { return v == 3 ? 9.f : 100.f; }                         // B^T of F(4,2): 3; of the 6-point forms: 10
static float dy_gain(int v) { return v == 1 ? 225.f : (v == 2 ? 49.f : 16.f); }        // A of F(4,3): 15; F(3,4): 7; F(4,2): 4
void wino_input_transform(Stream& s, int m, int r, const TView& x, int pad, int pad_mode, int Th, int Tw, float* V, float* amax_out,
                          const float* in_amax, int* kscale_out) {
  const int v = variant(m, r);
  if (v == 3) throw Error(1, "wino_input_transform: F(4,2) is the strided form (wino_s2_input_transform)");
  if (x.C % 4 || x.cs % 4) throw Error(1, "wino_input_transform: C must be a multiple of 4");
  const int vw = wino_vec_width();
  const size_t total = (size_t)x.N * Th * Tw * (x.C / (v == 0 ? 4 : vw));
  const dim3 grid(wgrid(total));
  if (v == 0 && in_amax) throw Error(1, "wino_input_transform: F(2,3) planes have no pair form");
#define SWN_WIN_LAUNCH(F, VW)                                                                                                  \
  hipLaunchKernelGGL((winog_input_kernel<F, VW>), grid, dim3(256), 0, hs(s), x.p, x.cs, x.N, x.H, x.W, x.C, pad, pad_mode, Th, \
                     Tw, V, amax_out, in_amax, input_gain(v), kscale_out)
  if (v == 0)            // (F(2,3) planes feed the fp32-operand kernels only: no slot to fill)
    hipLaunchKernelGGL(wino_input_kernel, grid, dim3(256), 0, hs(s), x.p, x.cs, x.N, x.H, x.W, x.C, pad, pad_mode, Th, Tw, V);
  else if (v == 1) SWN_WIN_LAUNCH(F43, 2);
  else SWN_WIN_LAUNCH(F34, 2);
#undef SWN_WIN_LAUNCH
  check_launch("wino_input_transform");
}
void wino_filter_transform(Stream& s, int m, int r, const WShape& w, int mode, const float* packed, float* U) {
  const int v = variant(m, r);
  if (v == 3) throw Error(1, "wino_filter_transform: F(4,2) is the strided form (wino_s2_filter_transform)");
  const int K = mode == 0 ? w.Cip : w.Npad, Nn = mode == 0 ? w.Npad : w.Cip;      // modes 1, 2: [Npad][Cip]
  const dim3 grid((unsigned)(((size_t)K * Nn + 255) / 256));
  if (v == 0) hipLaunchKernelGGL(wino_filter_kernel, grid, dim3(256), 0, hs(s), w, mode, K, Nn, packed, U);
  else if (v == 1) hipLaunchKernelGGL(winog_filter_kernel<F43>, grid, dim3(256), 0, hs(s), w, mode, K, Nn, packed, U);
  else hipLaunchKernelGGL(winog_filter_kernel<F34>, grid, dim3(256), 0, hs(s), w, mode, K, Nn, packed, U);
  check_launch("wino_filter_transform");
}
void wino_filter_transform_pc(Stream& s, int m, int r, const WShape& w, int mode, const float* packed, int bn, uint16_t* out,
                              size_t panel_elems, const float** amax_io) {
  const int v = variant(m, r);
  if (v != 1 && v != 2) throw Error(1, "wino_filter_transform_pc: the 6-point forms F(4,3) / F(3,4) only");
  if (mode < 0 || mode > 2) throw Error(1, "wino_filter_transform_pc: mode 0, 1 or 2");
  const int K = mode == 0 ? w.Cip : w.Npad, Nn = mode == 0 ? w.Npad : w.Cip;
  if (K % 16 || (bn != 64 && bn != 128)) throw Error(1, "wino_filter_transform_pc: K must be a multiple of 16, tile 64 or 128");
  const size_t total = (size_t)(K / 8) * ((Nn + bn - 1) / bn) * bn;
  const dim3 grid((unsigned)((((total + 255) / 256 + 7) / 8) * 8) * (unsigned)(m + r - 1));      // slabs in groups of 8, A rows each
  // two-plane form: the scale comes from the amax of the layer's packed weights ([r * r][Cip][Npad])
  const float* wamax = (amax_io && *amax_io) ? *amax_io : conv_precut_amax(s, packed, (size_t)r * r * w.Cip, w.Npad, 1, 0);
  if (amax_io) *amax_io = wamax;
  if (v == 1)
    hipLaunchKernelGGL(winog_filter_pc_kernel<F43>, grid, dim3(256), 0, hs(s), w, mode, K, Nn, bn, packed, out, panel_elems, wamax,
                       conv_precut_planes());
  else
    hipLaunchKernelGGL(winog_filter_pc_kernel<F34>, grid, dim3(256), 0, hs(s), w, mode, K, Nn, bn, packed, out, panel_elems, wamax,
                       conv_precut_planes());
  check_launch("wino_filter_transform_pc");
}
void wino_output_transform(Stream& s, int m, int r, const float* M, int Cm, int Th, int Tw, const float* bias, int act,
                           const TView& y, int Cout, int accumulate, float* amax_out) {
  const int v = variant(m, r);
  if (v == 0 && amax_out) throw Error(1, "wino_output_transform: the F(2,3) kernel has no amax fold");
  const size_t total = (size_t)y.N * Th * Tw * ((Cout + 3) / 4);
  const dim3 grid(wgrid(total));
  if (v == 0)
    hipLaunchKernelGGL(wino_output_kernel, grid, dim3(256), 0, hs(s), M, Cm, y.N, Th, Tw, bias, act, y.p, y.cs, y.H, y.W,
                       Cout, accumulate);
  else if (v == 1)
    hipLaunchKernelGGL(winog_output_kernel<F43>, grid, dim3(256), 0, hs(s), M, Cm, y.N, Th, Tw, bias, act, y.p, y.cs, y.H,
                       y.W, Cout, accumulate, amax_out);
  else if (v == 2)
    hipLaunchKernelGGL(winog_output_kernel<F34>, grid, dim3(256), 0, hs(s), M, Cm, y.N, Th, Tw, bias, act, y.p, y.cs, y.H,
                       y.W, Cout, accumulate, amax_out);
  else
    hipLaunchKernelGGL(winog_output_kernel<F42>, grid, dim3(256), 0, hs(s), M, Cm, y.N, Th, Tw, bias, act, y.p, y.cs, y.H,
                       y.W, Cout, accumulate, amax_out);
  check_launch("wino_output_transform");
}
void wino_dy_transform(Stream& s, int m, int r, const TView& dy, int Th, int Tw, float* dM, float* amax_out, const float* in_amax,
                       int* kscale_out) {
  const int v = variant(m, r);
  if (dy.C % 4 || dy.cs % 4) throw Error(1, "wino_dy_transform: C must be a multiple of 4");
  const int vw = v == 0 ? 4 : wino_vec_width();
  const size_t total = (size_t)dy.N * Th * Tw * (dy.C / vw);
  const dim3 grid(wgrid(total));
  if (v == 0 && in_amax) throw Error(1, "wino_dy_transform: F(2,3) planes have no pair form");
#define SWN_WDY_LAUNCH(F, VW)                                                                                                       \
  hipLaunchKernelGGL((winog_dy_kernel<F, VW>), grid, dim3(256), 0, hs(s), dy.p, dy.cs, dy.N, dy.H, dy.W, dy.C, Th, Tw, dM, amax_out, \
                     in_amax, dy_gain(v), kscale_out)
  if (v == 0)
    hipLaunchKernelGGL(wino_dy_kernel, grid, dim3(256), 0, hs(s), dy.p, dy.cs, dy.N, dy.H, dy.W, dy.C, Th, Tw, dM);
  else if (v == 1) SWN_WDY_LAUNCH(F43, 2);
  else if (v == 2) SWN_WDY_LAUNCH(F34, 2);
  else SWN_WDY_LAUNCH(F42, 2);
#undef SWN_WDY_LAUNCH
  check_launch("wino_dy_transform");
}
void tailw_filter_transform(Stream& s, const WShape& w, const float* folded, float* U) {
  const size_t total = (size_t)w.Cip * 4 * w.Npad;
  hipLaunchKernelGGL(tailw_filter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, hs(s), w.Cip, w.Npad, folded, U);
  check_launch("tailw_filter_transform");
}
void tailw_filter_grad(Stream& s, const WShape& w, const float* dU, float* dfolded) {
  const size_t total = (size_t)w.Cip * 4 * w.Npad;
  hipLaunchKernelGGL(tailw_filter_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, hs(s), w.Cip, w.Npad, dU, dfolded);
  check_launch("tailw_filter_grad");
}
void tailw_output_transform(Stream& s, const float* M, int Th, int Tw, int Npad, const float* bias, int act, const TView& y, int Cout) {
  if (Npad % 4 || y.cs % 4 || y.C < Npad) throw Error(1, "tailw_output_transform: bad channel counts");
  const size_t total = (size_t)y.N * Th * Tw * 4 * (Npad / 4);
  hipLaunchKernelGGL(tailw_output_kernel, dim3(wgrid(total)), dim3(256), 0, hs(s), M, y.N, Th, Tw, Npad, bias, act, y.p, y.cs, y.H, y.W, Cout);
  check_launch("tailw_output_transform");
}
void tailw_dy_transform(Stream& s, const TView& dy, int Th, int Tw, int Npad, float* dM, float* amax_out, const float* in_amax,
                        int* kscale_out) {
  if (Npad % 4 || dy.cs % 4 || dy.C < Npad) throw Error(1, "tailw_dy_transform: bad channel counts");
  const size_t total = (size_t)dy.N * Th * Tw * 4 * (Npad / 4);
  hipLaunchKernelGGL(tailw_dy_kernel, dim3(wgrid(total)), dim3(256), 0, hs(s), dy.p, dy.cs, dy.N, dy.H, dy.W, Th, Tw, Npad, dM, amax_out,
                     in_amax, 225.f, kscale_out);
  check_launch("tailw_dy_transform");
}
void wino_s2_input_transform(Stream& s, const TView& x, int Th, int Tw, float* V, float* amax_out, const float* in_amax, int* kscale_out) {
  if (x.C % 4 || x.cs % 4) throw Error(1, "wino_s2_input_transform: C must be a multiple of 4");
  const int vw = wino_vec_width();
  const size_t total = (size_t)x.N * Th * Tw * 4 * (x.C / vw);
  hipLaunchKernelGGL(wino_s2_input_kernel<2>, dim3(wgrid(total)), dim3(256), 0, hs(s), x.p, x.cs, x.N, x.H, x.W, x.C, Th, Tw, V,
                       amax_out, in_amax, 9.f, kscale_out);
  check_launch("wino_s2_input_transform");
}
void wino_s2_input_adjoint(Stream& s, float* dV, int Cf, int Th, int Tw, const TView& dx, const float* bias, int accumulate) {
  if (Cf % 4 || dx.C != Cf || dx.cs % 4) throw Error(1, "wino_s2_input_adjoint: bad channel count");
  const size_t T = (size_t)dx.N * Th * Tw;
  hipLaunchKernelGGL(winog_patch_kernel<F42>, dim3(wgrid(T * Cf)), dim3(256), 0, hs(s), dV, 4 * Cf, T);
  hipLaunchKernelGGL(wino_s2_fold_kernel, dim3(wgrid(dx.pixels() * (Cf / 4))), dim3(256), 0, hs(s), dV, Cf, dx.N, dx.H, dx.W, Th, Tw,
                     bias, dx.p, dx.cs, accumulate);
  check_launch("wino_s2_input_adjoint");
}
static void s2_dims(const WShape& w, int& Cf, int& Cc) {
  if (w.KH != 4 || w.KW != 4) throw Error(1, "strided Winograd: k4 s2 layers only");
  Cf = w.kind == WK_CONV ? w.Cip : w.Npad;
  Cc = w.kind == WK_CONV ? w.Npad : w.Cip;
}
void wino_s2_filter_transform(Stream& s, const WShape& w, int mode, const float* packed, float* U) {
  int Cf, Cc;
  s2_dims(w, Cf, Cc);
  hipLaunchKernelGGL(wino_s2_filter_kernel, dim3((unsigned)(((size_t)Cf * Cc + 255) / 256)), dim3(256), 0, hs(s), w, mode, Cf, Cc, packed, U);
  check_launch("wino_s2_filter_transform");
}
void wino_s2_filter_grad(Stream& s, const WShape& w, const float* dU, float* dpacked) {
  int Cf, Cc;
  s2_dims(w, Cf, Cc);
  hipLaunchKernelGGL(wino_s2_filter_grad_kernel, dim3((unsigned)(((size_t)Cf * Cc + 255) / 256)), dim3(256), 0, hs(s), w, Cf, Cc, dU, dpacked);
  check_launch("wino_s2_filter_grad");
}
void wino_input_adjoint(Stream& s, int m, int r, float* dV, int C, int pad, int pad_mode, int Th, int Tw, const TView& dx,
                        int accumulate) {
  const int v = variant(m, r);
  if (v == 3) throw Error(1, "wino_input_adjoint: F(4,2) is the strided form (wino_s2_input_adjoint)");
  if (C % 4 || dx.C != C || dx.cs % 4) throw Error(1, "wino_input_adjoint: bad channel count");
  const size_t T = (size_t)dx.N * Th * Tw;
  const dim3 grid(wgrid(T * (C / 4)));
  if (v == 0) hipLaunchKernelGGL(winog_patch_kernel<F23>, grid, dim3(256), 0, hs(s), dV, C, T);
  else if (v == 1) hipLaunchKernelGGL(winog_patch_kernel<F43>, grid, dim3(256), 0, hs(s), dV, C, T);
  else hipLaunchKernelGGL(winog_patch_kernel<F34>, grid, dim3(256), 0, hs(s), dV, C, T);
  const int A = m + r - 1;
  hipLaunchKernelGGL(wino_fold_kernel, dim3(wgrid(dx.pixels() * (C / 4))), dim3(256), 0, hs(s), dV, C, dx.N, dx.H, dx.W, m, A, pad,
                     pad_mode, Th, Tw, dx.p, dx.cs, accumulate);
  check_launch("wino_input_adjoint");
}
void wino_filter_grad(Stream& s, int m, int r, const WShape& w, const float* dU, float* dpacked) {
  const int v = variant(m, r);
  if (v == 3) throw Error(1, "wino_filter_grad: F(4,2) is the strided form (wino_s2_filter_grad)");
  const size_t total = (size_t)w.Cip * w.Npad;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (v == 0) hipLaunchKernelGGL(wino_filter_grad_kernel, grid, dim3(256), 0, hs(s), w, dU, dpacked);
  else if (v == 1) hipLaunchKernelGGL(winog_filter_grad_kernel<F43>, grid, dim3(256), 0, hs(s), w, dU, dpacked);
  else hipLaunchKernelGGL(winog_filter_grad_kernel<F34>, grid, dim3(256), 0, hs(s), w, dU, dpacked);
  check_launch("wino_filter_grad");
}

}  // namespace swn
