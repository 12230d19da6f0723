// fork_probe (round 6): when does a kernel on the MAIN stream start, relative to a batch of kernels on a SIDE stream, for the event
// patterns the engine uses (Ctx::fork_side, Net::prefetch_dgrad, Net::need)?  Every kernel spins for a given time and leaves its
// start / end wall-clock ticks (s_memrealtime, 100 MHz); the host prints the timeline.
//   build: hipcc --offload-arch=gfx950 -O2 tools/fork_probe.hip -o tools/_bin/fork_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(3); } } while (0)
__global__ void spin(unsigned long long* slot, int us) {
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) slot[0] = t0;
  while (wall_clock64() - t0 < (unsigned long long)us * 100ull) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) slot[1] = wall_clock64();
}
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 100, side_us = argc > 2 ? atoi(argv[2]) : 20, flags = argc > 3 ? atoi(argv[3]) : (int)hipEventDisableTiming, a_us = argc > 4 ? atoi(argv[4]) : 5000;
  hipStream_t mainS, side;
  CK(hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  unsigned long long* d; CK(hipMalloc(&d, (N + 8) * 16)); std::vector<unsigned long long> h((N + 8) * 2);
  std::vector<hipEvent_t> ev(N + 4); for (auto& e : ev) CK(hipEventCreateWithFlags(&e, flags));
  auto launch = [&](hipStream_t s, int slot, int us) { hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, d + 2 * slot, us); };
  // slot 0: A on main, slots 1..N: side kernels, slot N+1: B on main, slot N+2: C on main
  for (int c = 0; c <= 5; c++) {
    for (int warm = 0; warm < 2; warm++) {
      CK(hipMemset(d, 0, (N + 8) * 16)); CK(hipDeviceSynchronize());
      launch(mainS, 0, a_us);                                                 // A (long: the host finishes enqueuing everything while it runs)
      if (c >= 1) { CK(hipEventRecord(ev[N], mainS)); CK(hipStreamWaitEvent(side, ev[N], 0)); }      // fork: side behind A
      if (c == 4 || c == 5) {                                                     // the interleaved order: the wait is issued while only a few side kernels are enqueued
        const int first = c == 4 ? 1 : 5;
        for (int i = 1; i <= first; i++) { launch(side, i, side_us); CK(hipEventRecord(ev[i], side)); }
        CK(hipStreamWaitEvent(mainS, ev[1], 0));
        launch(mainS, N + 1, 100);                                                // B
        for (int i = first + 1; i <= N; i++) { launch(side, i, side_us); CK(hipEventRecord(ev[i], side)); }
      } else {
        for (int i = 1; i <= N; i++) { launch(side, i, side_us); if (c >= 2) CK(hipEventRecord(ev[i], side)); }
        if (c == 3) CK(hipStreamWaitEvent(mainS, ev[1], 0));                      // main needs only the FIRST side kernel
        launch(mainS, N + 1, 100);                                                // B
      }
      launch(mainS, N + 2, 10);                                                   // C
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), d, (N + 8) * 16, hipMemcpyDeviceToHost));
    }
    static const char* what[] = {"no events at all (two independent streams)", "fork only (side waits for A); B launched behind the side batch, no wait", "fork + an event recorded behind every side kernel; B not waiting",
                                 "fork + per-kernel events; B waits for the FIRST side kernel's event (wait issued after the whole batch was enqueued)",
                                 "as above, the wait and B issued when only side kernel 1 was enqueued", "as above, wait for kernel 1 issued when 5 side kernels were enqueued"};
    const double t0 = (double)h[1];      // end of A
    int done_before_B = 0; for (int i = 1; i <= N; i++) done_before_B += h[2 * i + 1] <= h[2 * (N + 1)];
    printf("case %d: %s\n   side: first start %+8.1f us, last end %+8.1f us | B start %+8.1f us (side kernels finished before B started: %d of %d) | C end %+8.1f us\n", c, what[c],
           (h[2] - t0) / 100.0, (h[2 * N + 1] - t0) / 100.0, (h[2 * (N + 1)] - t0) / 100.0, done_before_B, N, (h[2 * (N + 2) + 1] - t0) / 100.0);
  }
  return 0;
}
