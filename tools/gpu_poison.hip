// gpu_poison (round 6): fills (almost) all free device memory with a bit pattern and frees it again, so that the next process on the box
// sees that pattern -- not a previous run's leftovers -- wherever it reads memory it did not write.  A library whose results depend on
// uninitialised memory changes its weight-arena hashes (tools/native_ab ... hash) behind this.
// usage: gpu_poison [pattern hex, default 7fc12345 = a NaN]      build: hipcc --offload-arch=gfx950 -O2 tools/gpu_poison.hip -o tools/_bin/gpu_poison
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void fill(unsigned* p, size_t n, unsigned v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v ^ (unsigned)(i * 2654435761u >> 27);
}
int main(int argc, char** argv) {
  const unsigned pat = argc > 1 ? (unsigned)strtoul(argv[1], nullptr, 16) : 0x7fc12345u;
  size_t fr = 0, tot = 0; hipMemGetInfo(&fr, &tot);
  std::vector<void*> bufs; size_t got = 0; const size_t chunk = (size_t)4 << 30;
  while (got + chunk + ((size_t)8 << 30) < fr) { void* p = nullptr; if (hipMalloc(&p, chunk) != hipSuccess) break; bufs.push_back(p); got += chunk; }
  for (void* p : bufs) hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (unsigned*)p, chunk / 4, pat);
  hipDeviceSynchronize();
  for (void* p : bufs) hipFree(p);
  printf("poisoned %.1f GB of %.1f GB free with %08x\n", got / 1e9, fr / 1e9, pat);
  return 0;
}
