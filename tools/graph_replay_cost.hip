// graph_replay_cost (round 6) -- what a recorded hipGraph costs per node on this runtime, next to the same launches issued eagerly.
// The captured training step (swn_model_step_captured, BASELINE.json C5's "hipGraph-captured step") replays ~500 kernel nodes recorded
// from two streams and measured 0.6 ms SLOWER than eager launches in rounds 4 and 5 (1284 vs 1318 img/s).  This tool separates the
// runtime's share from the library's: chains of N kernels of a fixed device duration (a clock-spin of `us` microseconds, or empty),
// (a) launched eagerly on one stream, (b) recorded on one stream and replayed, (c) in groups of four with one kernel forked onto a second stream
// BESIDE the other three and joined behind them (the shape of the step: weight gradients beside the input-gradient chain) -- a kernel
// occupies one workgroup per CU, so two can run side by side: concurrent branches take 3 kernel times per group, serialised ones 4.  Host enqueue time and end-to-end
// time per chain, per node.  If replay's end-to-end time per node exceeds eager's while the kernels themselves are identical, the
// difference is the runtime's inter-node scheduling (a barrier packet / signal wait per graph node) and no re-arrangement of the
// recorded step removes it.
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/graph_replay_cost.hip -o tools/_bin/graph_replay_cost
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void spin_kernel(long long ticks, int* sink) {
  // one workgroup per CU would measure dispatch of a wide grid; the step's kernels ARE wide, so: 256 workgroups of 256 threads
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink && threadIdx.x == 0 && blockIdx.x == 0 && ticks < 0) *sink = 1;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  int wall_khz = 0; CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t fork_ev, join_ev; CK(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join_ev, hipEventDisableTiming));
  int* sink; CK(hipMalloc((void**)&sink, 4));
  printf("wall clock %d kHz; %d repetitions per figure; grid 256 x 256 threads per kernel\n", wall_khz, reps);
  printf("%-6s %-7s %-34s %12s %12s %12s\n", "nodes", "us/node", "form", "enqueue us", "total us", "per node us");
  for (int us : {0, 10, 40}) {
    const long long ticks = (long long)us * wall_khz / 1000;
    for (int N : {100, 400, 500}) {
      auto chain = [&](bool two) {
        // groups of four: one kernel on the second stream BESIDE three on the first (fork in front of the group, join behind it): if the
        // two branches really run concurrently a group takes 3 kernel times, serialised it takes 4
        for (int i = 0; i < N; i += 4) {
          if (two) {
            CK(hipEventRecord(fork_ev, s0)); CK(hipStreamWaitEvent(s1, fork_ev, 0));
            hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s1, ticks, sink);
            CK(hipEventRecord(join_ev, s1));
          } else {
            hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s0, ticks, sink);
          }
          for (int j = 1; j < 4 && i + j < N; ++j) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, s0, ticks, sink);
          if (two) CK(hipStreamWaitEvent(s0, join_ev, 0));
        }
      };
      for (int form = 0; form < 4; ++form) {
        const bool two = form & 1, graph = form >= 2;
        hipGraphExec_t exec = nullptr;
        if (graph) {
          hipGraph_t g;
          CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
          chain(two);
          CK(hipStreamEndCapture(s0, &g));
          CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
          CK(hipGraphDestroy(g));
        }
        auto run = [&] { if (graph) CK(hipGraphLaunch(exec, s0)); else chain(two); };
        run(); CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
        double enq = 0, tot = 0;
        for (int r = 0; r < reps; ++r) {
          const double a = now();
          run();
          const double b = now();
          CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
          const double c = now();
          enq += b - a; tot += c - a;
        }
        printf("%-6d %-7d %-34s %12.1f %12.1f %12.3f\n", N, us,
               form == 0 ? "eager, one stream" : form == 1 ? "eager, 1 of 4 on a 2nd stream" : form == 2 ? "graph replay, one stream" : "graph replay, 1 of 4 on a 2nd stream",
               enq / reps * 1e6, tot / reps * 1e6, tot / reps * 1e6 / N);
        fflush(stdout);
        if (exec) CK(hipGraphExecDestroy(exec));
      }
    }
  }
  return 0;
}
