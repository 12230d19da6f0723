// swapnet_amd -- device memory / stream primitives of ops.h (HIP implementation).
#include "hip_util.h"
#include <algorithm>
#include <cstdint>
#include <cstring>

namespace swn {

void* dev_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0) bytes = 16;
  SWN_HIP_CHECK(hipMalloc(&p, bytes));
  SWN_HIP_CHECK(hipMemset(p, 0, bytes));
  // hipMemset on device memory returns before the fill has run (it is a launch on the null stream), and the library's streams are
  // hipStreamNonBlocking: they do NOT order behind the null stream.  Without this wait a stream operation issued right behind the
  // allocation could land BEFORE the fill and be zeroed by it -- observed in round 6 (tools/native_ab ... trace, first process of a
  // call, where the first fill also pays the runtime's lazy start-up): the conditional-input channel map of PatchGAN's first layer
  // (ParamArena::allocate: alloc + upload on the context's stream) arrived zeroed, pack_weight dropped ten input channels of
  // model.0.weight and the whole run computed with another discriminator (profiles/alloc_fill_race_r06.txt).
  SWN_HIP_CHECK(hipStreamSynchronize(nullptr));
  return p;
}
void dev_free(void* p) {
  if (p) (void)hipFree(p);
}
// Fill as an ordinary kernel launch of ours, not hipMemsetAsync: the fills of this library sit inside sequences that are recorded into
// hipGraphs (the amax slots at the top of every forward pass, loss accumulators), and a recorded MEMSET node was observed to
// misbehave on replay once another model of the same context had issued eager hipMemsetAsync calls in between (round 4:
// tools/r04_pipe_probe3.py -- the replayed texture stage read garbage scales; the kernel node is immune).
__global__ __launch_bounds__(256) void fill32_kernel(uint32_t* p, uint32_t v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
void dev_memset(Stream& s, void* p, int v, size_t bytes) {
  if (bytes == 0) return;
  if (bytes % 4 == 0 && ((uintptr_t)p & 3) == 0) {
    const uint32_t b = (uint32_t)(v & 0xff), pat = b | (b << 8) | (b << 16) | (b << 24);
    const size_t n = bytes / 4;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(fill32_kernel, dim3(grid), dim3(256), 0, hs(s), static_cast<uint32_t*>(p), pat, n);
    SWN_HIP_CHECK(hipGetLastError());
    return;
  }
  SWN_HIP_CHECK(hipMemsetAsync(p, v, bytes, hs(s)));
}
struct Small64 { uint32_t w[16]; };
__global__ void store_small_kernel(uint32_t* dst, Small64 v, int n) {
  if (threadIdx.x < (unsigned)n) dst[threadIdx.x] = v.w[threadIdx.x];
}
void dev_store_small(Stream& s, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return;
  if (bytes > 64 || bytes % 4 || ((uintptr_t)dst & 3)) throw Error(1, "dev_store_small: at most 64 bytes, 4-byte granular");
  Small64 v{};
  memcpy(v.w, src, bytes);
  hipLaunchKernelGGL(store_small_kernel, dim3(1), dim3(64), 0, hs(s), static_cast<uint32_t*>(dst), v, (int)(bytes / 4));
  SWN_HIP_CHECK(hipGetLastError());
}
void dev_copy(Stream& s, void* dst, const void* src, size_t bytes) {
  SWN_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, hs(s)));
}
void dev_upload(Stream& s, void* dst, const void* src, size_t bytes) {
  SWN_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, hs(s)));
  SWN_HIP_CHECK(hipStreamSynchronize(hs(s)));
}
void dev_download(Stream& s, void* dst, const void* src, size_t bytes) {
  SWN_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, hs(s)));
  SWN_HIP_CHECK(hipStreamSynchronize(hs(s)));
}
void stream_sync(Stream& s) { SWN_HIP_CHECK(hipStreamSynchronize(hs(s))); }
void* stream_create(int device) {
  int count = 0;
  SWN_HIP_CHECK(hipGetDeviceCount(&count));
  if (count <= 0) throw Error(3, "swapnet_hip: no HIP device visible (this library has no CPU path)");
  if (device < 0 || device >= count) throw Error(1, "swapnet_hip: bad device index");
  SWN_HIP_CHECK(hipSetDevice(device));
  hipStream_t st;
  SWN_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  return (void*)st;
}
void device_check(int device) {
  int count = 0;
  SWN_HIP_CHECK(hipGetDeviceCount(&count));
  if (count <= 0) throw Error(3, "swapnet_hip: no HIP device visible (this library has no CPU path)");
  if (device < 0 || device >= count) throw Error(1, "swapnet_hip: bad device index");
  SWN_HIP_CHECK(hipSetDevice(device));
}
void stream_destroy(void* h) {
  if (h) (void)hipStreamDestroy((hipStream_t)h);
}
void* event_create() {
  hipEvent_t e;
  SWN_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return (void*)e;
}
void event_destroy(void* ev) {
  if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}
void event_record(void* ev, Stream& s) { SWN_HIP_CHECK(hipEventRecord((hipEvent_t)ev, hs(s))); }
void stream_wait_event(Stream& s, void* ev) { SWN_HIP_CHECK(hipStreamWaitEvent(hs(s), (hipEvent_t)ev, 0)); }
int is_device_build() { return 1; }

void* stream_create_current() {
  hipStream_t st;
  SWN_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  return (void*)st;
}
void graph_begin(Stream& s) { SWN_HIP_CHECK(hipStreamBeginCapture(hs(s), hipStreamCaptureModeThreadLocal)); }
void* graph_end(Stream& s) {
  hipGraph_t g = nullptr;
  SWN_HIP_CHECK(hipStreamEndCapture(hs(s), &g));
  hipGraphExec_t exec = nullptr;
  const hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) throw Error(2, std::string("hipGraphInstantiate failed: ") + hipGetErrorString(e));
  return (void*)exec;
}
void graph_abort(Stream& s) {
  hipGraph_t g = nullptr;
  (void)hipStreamEndCapture(hs(s), &g);
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
}
void graph_launch(void* exec, Stream& s) { SWN_HIP_CHECK(hipGraphLaunch((hipGraphExec_t)exec, hs(s))); }
void graph_destroy(void* exec) {
  if (exec) (void)hipGraphExecDestroy((hipGraphExec_t)exec);
}

}  // namespace swn
