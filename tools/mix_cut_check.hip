// mix_cut_check (round 6): the fused low-plane cut of conv_gemm.hip (resid_pack: v_fma_mixlo / mixhi_f16 of x * 1 - h) against the
// cvt / sub / cvt sequence it replaces, bit for bit, for truncated and nearest h over normal, subnormal and scaled values.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/mix_cut_check.hip -o tools/_bin/mixtest
#include <hip/hip_runtime.h>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned resid_pack(float x0, float x1, unsigned h) {
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(h));
  return l;
}
__global__ void k(const float* v, float sa, unsigned* out, int mode) {
  const int i = threadIdx.x;
  f32x2 x = f32x2{v[2*i], v[2*i+1]} * f32x2{sa, sa};
  unsigned h, l;
  if (mode == 0) h = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x[0], x[1]));
  else h = __builtin_bit_cast(unsigned, f16x2{(_Float16)x[0], (_Float16)x[1]});
  l = resid_pack(x[0], x[1], h);
  out[2*i] = h; out[2*i+1] = l;
}
__global__ void kref(const float* v, float sa, unsigned* out, int mode) {
  const int i = threadIdx.x;
  const float x0 = v[2*i]*sa, x1 = v[2*i+1]*sa;
  f16x2 h;
  if (mode == 0) h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0, x1)); else h = f16x2{(_Float16)x0, (_Float16)x1};
  const float r0 = x0 - (float)h[0], r1 = x1 - (float)h[1];
  out[2*i] = __builtin_bit_cast(unsigned, h); out[2*i+1] = __builtin_bit_cast(unsigned, f16x2{(_Float16)r0, (_Float16)r1});
}
int main() {
  const int N = 256; float hv[2*N]; unsigned s = 12345;
  for (int i = 0; i < 2*N; ++i) { s = s*1664525u+1013904223u; int e = (int)((s>>8)%40) - 30; s = s*1664525u+1013904223u; hv[i] = ldexpf(((int)(s>>8)-(1<<23))/8388608.0f, e); }
  hv[0]=0.f; hv[1]=-0.f; hv[2]=1e-30f; hv[3]=65504.f/4096.f;
  float* dv; unsigned *da, *db; hipMalloc((void**)&dv, sizeof hv); hipMalloc((void**)&da, 8*N); hipMalloc((void**)&db, 8*N);
  hipMemcpy(dv, hv, sizeof hv, hipMemcpyHostToDevice);
  int bad = 0;
  for (int mode = 0; mode < 2; ++mode) for (float sa : {1.0f, 4096.f, 1.f/1024}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(N), 0, 0, dv, sa, da, mode); hipLaunchKernelGGL(kref, dim3(1), dim3(N), 0, 0, dv, sa, db, mode);
    unsigned ha[2*N], hb[2*N]; hipMemcpy(ha, da, 8*N, hipMemcpyDeviceToHost); hipMemcpy(hb, db, 8*N, hipMemcpyDeviceToHost);
    for (int i = 0; i < 2*N; ++i) if (ha[i] != hb[i]) { if (bad < 8) printf("mode %d sa %g i %d: %08x vs %08x (v %g %g)\n", mode, sa, i, ha[i], hb[i], hv[i&~1], hv[i|1]); ++bad; }
  }
  printf("mismatches: %d\n", bad);
  return bad != 0;
}
