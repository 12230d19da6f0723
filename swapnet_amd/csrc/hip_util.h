// swapnet_amd -- HIP-side helpers shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "common.h"
#include "ops.h"

namespace swn {

#define SWN_HIP_CHECK(expr)                                                                 \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      throw ::swn::Error(2, std::string("HIP error ") + hipGetErrorString(_e) + " at " +   \
                                __FILE__ + ":" + std::to_string(__LINE__) + " in " + #expr); \
  } while (0)

inline hipStream_t hs(const Stream& s) { return (hipStream_t)s.handle; }

inline void check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw Error(2, std::string("kernel launch failed (") + what + "): " + hipGetErrorString(e));
}

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case ACT_LRELU: return v > 0.f ? v : 0.2f * v;
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// derivative of act expressed through the activation's INPUT x (lrelu/relu) or OUTPUT y (tanh)
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
  switch (act) {
    case ACT_LRELU: return y > 0.f ? 1.f : 0.2f;
    case ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}

// ---- amax slots (ops.h ConvFwdArgs::x_amax): a producer folds max |v| over what it writes into one entry of the slot --
// an atomic max on the bit pattern (non-negative floats order like unsigned integers: exact and order-independent, so the
// result is deterministic), skipped when the entry already holds as much.  NaN (sign bit cleared by fabsf) orders above
// every finite value: a NaN anywhere makes the slot NaN, which the consumers treat as "no scale".
__device__ __forceinline__ float f4amax(const float4& v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
__device__ __forceinline__ void amax_store(float m, float* amax_out, unsigned entry) {
  unsigned* dst = reinterpret_cast<unsigned*>(amax_out) + (entry % AMAX_SLOT);
  const unsigned bits = __float_as_uint(m);
  if (bits > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, bits);
}
// one atomic per wave (no barrier: usable where lanes have left the kernel, as long as the calling wave is converged)
__device__ __forceinline__ void amax_fold_wave(float am, float* amax_out, unsigned entry) {
#pragma unroll
  for (int o = 32; o; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
  if ((threadIdx.x & 63) == 0) amax_store(am, amax_out, entry);
}
// one atomic per block: ALL threads of a 1-D block of at most 1024 threads must call it
__device__ __forceinline__ void amax_fold(float am, float* amax_out) {
  if (!amax_out) return;                       // (uniform over the grid)
  __shared__ float amax_red[16];
#pragma unroll
  for (int o = 32; o; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
  if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = am;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = amax_red[0];
    for (int w = 1; w < (int)((blockDim.x + 63) >> 6); ++w) m = fmaxf(m, amax_red[w]);
    amax_store(m, amax_out, blockIdx.x + blockIdx.y * gridDim.x);
  }
}

// counter-based dropout stream: keep decision for element `idx` of the call with `seed`
__device__ __forceinline__ uint32_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 11);
}
__device__ __forceinline__ float drop_scale(uint64_t seed, uint64_t idx, float p) {
  // uniform in [0,1) from 24 random bits; keep with probability 1-p, scale 1/(1-p)
  float u = (float)(mix64(seed * 0xD1342543DE82EF95ull + idx) & 0xFFFFFFu) * (1.0f / 16777216.0f);
  return u >= p ? 1.0f / (1.0f - p) : 0.0f;
}

}  // namespace swn
