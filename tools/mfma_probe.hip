// Layout probe for v_mfma_f32_4x4x1_16b_f32 (used by the narrow-N conv kernels): prints, for
// each (lane, register) of the result, which lane's A value and which lane's B value it used,
// without broadcast and with the A block broadcast (cbsz = 4, abid = 2).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
  const int l = threadIdx.x;
  const float id = (float)l, one = 1.f;
  const f32x4 c = {0, 0, 0, 0};
  const f32x4 a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(id, one, c, 0, 0, 0);   // A lane
  const f32x4 b0 = __builtin_amdgcn_mfma_f32_4x4x1f32(one, id, c, 0, 0, 0);   // B lane
  const f32x4 a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(id, one, c, 4, 2, 0);
  const f32x4 b1 = __builtin_amdgcn_mfma_f32_4x4x1f32(one, id, c, 4, 2, 0);
  for (int r = 0; r < 4; ++r) {
    out[l * 16 + r] = a0[r]; out[l * 16 + 4 + r] = b0[r]; out[l * 16 + 8 + r] = a1[r]; out[l * 16 + 12 + r] = b1[r];
  }
}
int main() {
  float* d;
  if (hipMalloc(&d, 64 * 16 * 4) != hipSuccess) return 1;
  probe<<<1, 64>>>(d);
  float h[1024];
  if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d  plain:", l);
    for (int r = 0; r < 4; ++r) printf(" (A%2.0f,B%2.0f)", h[l * 16 + r], h[l * 16 + 4 + r]);
    printf("   cbsz4/abid2:");
    for (int r = 0; r < 4; ++r) printf(" (A%2.0f,B%2.0f)", h[l * 16 + 8 + r], h[l * 16 + 12 + r]);
    printf("\n");
    if (l == 9) l = 57;
  }
  return 0;
}
