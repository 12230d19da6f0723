// mfma_dep_probe -- issue rate of v_mfma_f32_32x32x16_bf16 as a function of how many independent accumulators a wave
// rotates through (dependent-accumulator latency) and of the waves per SIMD.  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void probe(const float* in, float* out, int iters) {
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)in[(threadIdx.x * 8 + i) & 4095]; b[i] = (__bf16)in[(threadIdx.x * 8 + i + 2048) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / (NACC > 8 ? 8 : NACC) * (NACC > 8 ? 1 : 1); ++r)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  }
  float r = 0.f;
  for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) r += acc[j][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
// the 16x16x32 form: 4 accumulator registers per block, so a 64x64 wave tile is 16 independent accumulators
template <int NACC>
__global__ __launch_bounds__(256) void probe16(const float* in, float* out, int iters) {
  f32x4 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)in[(threadIdx.x * 8 + i) & 4095]; b[i] = (__bf16)in[(threadIdx.x * 8 + i + 2048) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
  }
  float r = 0.f;
  for (int j = 0; j < NACC; ++j) for (int e = 0; e < 4; ++e) r += acc[j][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// the K = 16 form of the 16x16 block (4 k per lane): fits 16-k LDS stages
template <int NACC>
__global__ __launch_bounds__(256) void probe16k16(const float* in, float* out, int iters) {
  f32x4 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
  bf16x4 a, b;
  for (int i = 0; i < 4; ++i) { a[i] = (__bf16)in[(threadIdx.x * 4 + i) & 4095]; b[i] = (__bf16)in[(threadIdx.x * 4 + i + 2048) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(__attribute__((ext_vector_type(4))) short, a), __builtin_bit_cast(__attribute__((ext_vector_type(4))) short, b), acc[j], 0, 0, 0);
  }
  float r = 0.f;
  for (int j = 0; j < NACC; ++j) for (int e = 0; e < 4; ++e) r += acc[j][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int NACC>
static void run16k16(hipStream_t st, const float* din, float* dout, hipEvent_t e0, hipEvent_t e1) {
  for (int wps : {1, 2, 3, 4}) {
    const int iters = 2000, blocks = 256 * wps, per_it = 16;
    hipLaunchKernelGGL(probe16k16<NACC>, dim3(blocks), dim3(256), 0, st, din, dout, 10);
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(probe16k16<NACC>, dim3(blocks), dim3(256), 0, st, din, dout, iters);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    printf("16x16x16 NACC %2d, %d wave(s)/SIMD: %.1f bf16 TFLOP/s, %.1f ns per MFMA per SIMD\n", NACC, wps,
           (double)blocks * 4 * iters * per_it * 8192.0 / t * 1e-9, t * 1e6 / ((double)wps * iters * per_it));
  }
}
template <int NACC>
static void run16(hipStream_t st, const float* din, float* dout, hipEvent_t e0, hipEvent_t e1) {
  for (int wps : {1, 2, 3, 4}) {
    const int iters = 2000, blocks = 256 * wps, per_it = 16;
    hipLaunchKernelGGL(probe16<NACC>, dim3(blocks), dim3(256), 0, st, din, dout, 10);
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(probe16<NACC>, dim3(blocks), dim3(256), 0, st, din, dout, iters);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    printf("16x16x32 NACC %2d, %d wave(s)/SIMD: %.1f bf16 TFLOP/s, %.1f ns per MFMA per SIMD\n", NACC, wps,
           (double)blocks * 4 * iters * per_it * 16384.0 / t * 1e-9, t * 1e6 / ((double)wps * iters * per_it));
  }
}
template <int NACC>
static void run(hipStream_t st, const float* din, float* dout, hipEvent_t e0, hipEvent_t e1) {
  for (int wps : {1, 2, 3, 4}) {
    const int iters = 2000, blocks = 256 * wps;          // 256-thread blocks: 1 wave per SIMD each
    const int per_it = (8 / NACC) * NACC;
    hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(256), 0, st, din, dout, 10);
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(256), 0, st, din, dout, iters);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    const double mfma_per_simd = (double)wps * iters * per_it;
    printf("NACC %d, %d wave(s)/SIMD: %.1f bf16 TFLOP/s, %.1f ns per MFMA per SIMD (32 cycles @2.4 GHz = 13.3 ns)\n", NACC, wps,
           (double)blocks * 4 * iters * per_it * 32768.0 / t * 1e-9, t * 1e6 / mfma_per_simd);
  }
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float *din, *dout;
  CK(hipMalloc((void**)&din, 4096 * 4)); CK(hipMalloc((void**)&dout, 256 * 4 * 256 * 4));
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
  CK(hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice));
  run16k16<4>(st, din, dout, e0, e1); run16k16<16>(st, din, dout, e0, e1);
  return 0;
}
