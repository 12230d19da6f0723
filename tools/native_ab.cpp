// Torch-free driver of the C-ABI (include/swapnet_hip.h): the G+D training step of the warp stage (BASELINE.json C2 by default) run
// from plain C++ against libswapnet_hip.so -- hipMalloc'd buffers, no Python, no torch.  A HIP process starts in a second where
// `import torch` on a fresh box takes a minute or two, so this is the instrument for short GPU calls:
//   1. conv_fwd_pcm_kernel (SWN_PC_MI=2, 256 x 128 tiles of 64-row wave tiles) against the shipped 128 x 128 pre-cut kernel on the
//      step's two largest launch shapes: bit-equality of forward and input gradient, route lines of both;
//   2. same-process A/B of the whole step, alternating blocks: default | SWN_PC_MI=2 (ms/step, losses);
//   3. the library-owned exchange with real RCCL at world size 1 (swn_ctx_attach_comm + swn_model_step_dp) against swn_model_step
//      from the same state: hashes of both weight arenas, losses, ms/step.
// Every line is flushed as it is produced: a call cut off by its time limit still leaves what it measured.
//   hipcc -O2 -std=c++17 tools/native_ab.cpp -Iinclude -Lswapnet_amd/csrc -lswapnet_hip -ldl -o tools/_bin/native_ab
//   (CPU check of the harness itself: g++ -DHOSTSIM ... -Ltests/hostsim/build -lswapnet_hostsim, then `native_ab 2 64 2`)
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <dlfcn.h>

#include "swapnet_hip.h"

#ifndef HOSTSIM
#include <hip/hip_runtime.h>
#define HCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); fflush(stdout); exit(3); } } while (0)
static void* dalloc(size_t bytes) { void* p = nullptr; HCHECK(hipMalloc(&p, bytes)); return p; }
static void dfree(void* p) { HCHECK(hipFree(p)); }
static void h2d(void* d, const void* h, size_t b) { HCHECK(hipMemcpy(d, h, b, hipMemcpyHostToDevice)); }
static void d2h(void* h, const void* d, size_t b) { HCHECK(hipMemcpy(h, d, b, hipMemcpyDeviceToHost)); }
static void dzero(void* d, size_t b) { HCHECK(hipMemset(d, 0, b)); HCHECK(hipDeviceSynchronize()); }
#else
static void* dalloc(size_t bytes) { return malloc(bytes); }
static void dfree(void* p) { free(p); }
static void h2d(void* d, const void* h, size_t b) { memcpy(d, h, b); }
static void d2h(void* h, const void* d, size_t b) { memcpy(h, d, b); }
static void dzero(void* d, size_t b) { memset(d, 0, b); }
#endif

#define SW(x) do { if ((x) != 0) { printf("FAILED %s: %s\n", #x, swn_last_error()); fflush(stdout); exit(2); } } while (0)
#define SAY(...) do { printf(__VA_ARGS__); fflush(stdout); } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// xorshift64* + a 12-uniform sum: close enough to N(0,1) for operand statistics, cheap enough for 140 M parameters
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint32_t rnd32() { rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27; return (uint32_t)((rng_state * 0x2545F4914F6CDD1Dull) >> 32); }
static inline float rndn() { uint32_t a = rnd32(), b = rnd32(), c = rnd32(); float s = 0.f; s += (a & 0xFFFF) + (a >> 16) + (b & 0xFFFF) + (b >> 16) + (c & 0xFFFF) + (c >> 16); return (s - 3.f * 65535.f) * (1.f / (65536.f * 0.70710678f)); }

static std::vector<float> NOISE;          // one block of normals, re-used (at shifted offsets) for every tensor
static void fill_normal(std::vector<float>& v, float std_, size_t salt) {
  if (NOISE.empty()) { NOISE.resize((size_t)1 << 24); for (auto& x : NOISE) x = rndn(); }
  size_t m = NOISE.size() - 1, o = (salt * 7919u) & m;
  for (size_t i = 0; i < v.size(); i++) v[i] = NOISE[(o + i) & m] * std_;
}

static uint64_t fnv(const void* p, size_t bytes) {
  const uint64_t* q = (const uint64_t*)p; uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < bytes / 8; i++) { h ^= q[i]; h *= 1099511628211ull; }
  return h;
}

static std::string route() { std::vector<char> b(1 << 20); swn_route_report(b.data(), (int)b.size()); return std::string(b.data()); }

struct Param { std::string name; size_t n; float* dev; };
struct State { std::vector<Param> p[2]; float* zeros = nullptr; size_t zeros_n = 0; };

static void init_params(swn_model* m, State& st) {
  for (int net = 0; net < 2; net++) {
    int cnt = 0; SW(swn_model_param_count(m, net, &cnt));
    for (int i = 0; i < cnt; i++) {
      char name[256]; int shape[4] = {1, 1, 1, 1}, nd = 0;
      SW(swn_model_param_info(m, net, i, name, 256, shape, &nd));
      size_t n = 1; for (int d = 0; d < nd; d++) n *= (size_t)shape[d];
      std::vector<float> h(n, 0.f);
      if (nd == 4) fill_normal(h, std::sqrt(2.f / (float)((size_t)shape[1] * shape[2] * shape[3])), (size_t)net * 1000 + i);   // kaiming, fan_in
      float* d = (float*)dalloc(n * 4); h2d(d, h.data(), n * 4);
      st.p[net].push_back({name, n, d});
      if (n > st.zeros_n) st.zeros_n = n;
    }
  }
  st.zeros = (float*)dalloc(st.zeros_n * 4); dzero(st.zeros, st.zeros_n * 4);
}

// weights from the kept copies, both Adam moments zero, step counters 0: what tests/backends.py reset_state does
static void reset_state(swn_model* m, State& st) {
  for (int net = 0; net < 2; net++) {
    for (auto& q : st.p[net]) {
      SW(swn_model_param_set(m, net, 0, q.name.c_str(), q.dev));
      SW(swn_model_param_set(m, net, 2, q.name.c_str(), st.zeros));
      SW(swn_model_param_set(m, net, 3, q.name.c_str(), st.zeros));
    }
    SW(swn_model_optim_step_set(m, net, 0));
  }
}

static void arena_hashes(swn_ctx* ctx, swn_model* m, uint64_t out[2]) {
  SW(swn_ctx_sync(ctx));
  for (int net = 0; net < 2; net++) {
    float* p = nullptr; size_t n = 0; SW(swn_model_weight_arena(m, net, &p, &n));
    std::vector<float> h(n); d2h(h.data(), p, n * 4); out[net] = fnv(h.data(), n * 4);
  }
}

static void say_losses(swn_model* m, const char* what) {
  float L[10] = {0}; SW(swn_model_get_losses(m, L, 10));
  bool fin = true; for (int i = 0; i < 6; i++) fin = fin && std::isfinite(L[i]);
  SAY("%s losses D %.5f D_real %.5f D_fake %.5f G %.5f G_gan %.5f G_ce %.5f finite %d\n", what, L[0], L[1], L[2], L[3], L[4], L[5], (int)fin);
}

// ---- 1. operator level: the same launch with SWN_PC_MI=1 and =2 -------------------------------------------------------------------
static void op_case(swn_ctx* ctx, const char* tag, int kind, int n, int ci, int h, int co, int ho) {
  size_t nx = (size_t)n * ci * h * h, nw = (size_t)co * ci * (kind == 1 ? 9 : 16), ny = (size_t)n * co * ho * ho;
  std::vector<float> hx(nx), hw(nw), hb(co), hdy(ny);
  fill_normal(hx, 1.f, 1); fill_normal(hw, std::sqrt(2.f / (float)(nw / co)), 2); fill_normal(hb, 0.1f, 3); fill_normal(hdy, 1.f, 4);
  float *x = (float*)dalloc(nx * 4), *w = (float*)dalloc(nw * 4), *b = (float*)dalloc(co * 4), *y = (float*)dalloc(ny * 4), *dx = (float*)dalloc(nx * 4);
  h2d(w, hw.data(), nw * 4); h2d(b, hb.data(), co * 4);
  std::vector<float> out[2], gin[2];
  for (int mi = 1; mi <= 2; mi++) {
    setenv("SWN_PC_MI", mi == 2 ? "2" : "1", 1); setenv("SWN_PC_MI_MIN_TILES", "1", 1);
    h2d(x, hx.data(), nx * 4); dzero(y, ny * 4);
    swn_route_trace(1);
    SW(swn_op_conv(ctx, kind, 0, 0, 0, x, n, ci, h, h, w, co, b, 0, y));                  // forward
    SW(swn_ctx_sync(ctx));
    out[mi - 1].resize(ny); d2h(out[mi - 1].data(), y, ny * 4);
    h2d(y, hdy.data(), ny * 4); dzero(dx, nx * 4);
    SW(swn_op_conv(ctx, kind, 0, 2, 0, dx, n, ci, h, h, w, co, nullptr, 0, y));           // input gradient
    SW(swn_ctx_sync(ctx));
    swn_route_trace(0);
    gin[mi - 1].resize(nx); d2h(gin[mi - 1].data(), dx, nx * 4);
    std::string r = route(); int pcm = 0, pc = 0;
    for (size_t p = 0; (p = r.find("conv_fwd_pc", p)) != std::string::npos; p++) { if (r.compare(p, 12, "conv_fwd_pcm") == 0) pcm++; else pc++; }
    SAY("op %s SWN_PC_MI=%d: route lines with conv_fwd_pcm %d, conv_fwd_pc %d\n", tag, mi, pcm, pc);
    if (mi == 2 && getenv("NATIVE_AB_VERBOSE")) SAY("%s\n", r.c_str());
  }
  for (int k = 0; k < 2; k++) {
    auto& a = k ? gin[0] : out[0]; auto& c = k ? gin[1] : out[1];
    double num = 0, den = 0; size_t nan = 0;
    for (size_t i = 0; i < a.size(); i++) { double d = (double)a[i] - c[i]; num += d * d; den += (double)a[i] * a[i]; nan += !std::isfinite(c[i]); }
    SAY("op %s %s: bit-equal %d  rel-L2(128x128, 256x128) %.3e  non-finite %zu  |a| %.4e\n", tag, k ? "dgrad" : "fwd",
        (int)(memcmp(a.data(), c.data(), a.size() * 4) == 0), std::sqrt(num / (den + 1e-300)), nan, std::sqrt(den / a.size()));
  }
  dfree(x); dfree(w); dfree(b); dfree(y); dfree(dx);
  setenv("SWN_PC_MI", "1", 1);
}

struct NcclId { char internal[128]; };

int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 32, H = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 10;
  int rounds = argc > 4 ? atoi(argv[4]) : 3;
  double t00 = now();
  SAY("native_ab: abi %d device build %d  B %d H %d K %d\n", swn_abi_version(), swn_is_device_build(), B, H, K);
  swn_ctx* ctx = nullptr;
  SW(swn_ctx_create(0, nullptr, 1, (size_t)1024 << 20, &ctx));
  SAY("ctx up at %.1f s\n", now() - t00);

  const bool generic = argc > 5 && (!strcmp(argv[5], "bench") || !strcmp(argv[5], "ab") || !strcmp(argv[5], "prof") || !strcmp(argv[5], "host") || !strcmp(argv[5], "phases") || !strcmp(argv[5], "hash") || !strcmp(argv[5], "trace") || !strcmp(argv[5], "truth"));
  if (!getenv("NATIVE_AB_SKIP_OPS") && !generic) {
    int big = H >= 256;
    op_case(ctx, "k4s2 64->128 (body_down2 / PatchGAN model.2 shape)", 0, big ? 8 : 2, 64, big ? 128 : 16, 128, big ? 64 : 8);
    op_case(ctx, "k3 reflect 1024->1024 @16x16 (resblock: 36 Winograd planes, pair-form operand)", 1, big ? 32 : 1, big ? 1024 : 256, 16, big ? 1024 : 256, 16);
    SAY("ops done at %.1f s\n", now() - t00);
  }

  swn_model* m = nullptr;
  SW(swn_warp_model_create(ctx, B, H, H, 1, 0.5f, &m));
  swn_hyper hy; memset(&hy, 0, sizeof hy);
  hy.lr = 1e-4f; hy.d_lr = 4e-4f; hy.weight_decay = 0.f; hy.d_weight_decay = 0.01f; hy.b1 = 0.9f; hy.b2 = 0.999f;
  hy.lambda_gan = 1.f; hy.lambda_ce = 100.f; hy.lambda_l1 = 10.f; hy.lambda_content = 20.f; hy.lambda_style = 1e-8f;
  hy.grad_scale = 1.f; hy.d_b1 = -1.f; hy.d_b2 = -1.f; hy.lambda_gp = 10.f;
  SW(swn_model_set_hyper(m, &hy));
  State st; init_params(m, st);
  reset_state(m, st);
  {   // synthetic batch: bodys ~ N(0,1); cloths = one-hot of blocky label maps, expanded on the device
    size_t nb = (size_t)B * 3 * H * H; std::vector<float> hb(nb); fill_normal(hb, 1.f, 77);
    float* db = (float*)dalloc(nb * 4); h2d(db, hb.data(), nb * 4);
    SW(swn_model_set_input(m, 0, db, B, 3, H, H));
    for (int slot = 1; slot <= 2; slot++) {
      std::vector<int32_t> lab((size_t)B * H * H);
      for (int n = 0; n < B; n++) for (int by = 0; by < H / 8; by++) for (int bx = 0; bx < H / 8; bx++) {
        int32_t v = (int32_t)(rnd32() % 19);
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) lab[((size_t)n * H + by * 8 + y) * H + bx * 8 + x] = v;
      }
      int32_t* dl = (int32_t*)dalloc(lab.size() * 4); h2d(dl, lab.data(), lab.size() * 4);
      SW(swn_model_set_input_labels(m, slot, dl, B, H, H));
      SW(swn_ctx_sync(ctx)); dfree(dl);
    }
    SW(swn_ctx_sync(ctx)); dfree(db);
  }
  size_t bytes = 0; swn_ctx_bytes_allocated(ctx, &bytes);
  SAY("model up at %.1f s, %.1f GB allocated by the library\n", now() - t00, bytes / 1e9);

  const float labels[3] = {0.9f, 0.8f, 1.0f};
  uint64_t seed = 1000;
  auto block = [&](const char* what, int mi, int steps, const char* min_tiles = nullptr) {
    setenv("SWN_PC_MI", mi == 2 ? "2" : "1", 1);
    if (min_tiles) setenv("SWN_PC_MI_MIN_TILES", min_tiles, 1); else unsetenv("SWN_PC_MI_MIN_TILES");
    SW(swn_ctx_sync(ctx));
    double t0 = now();
    for (int i = 0; i < steps; i++) SW(swn_model_step(m, labels, 1, ++seed));
    SW(swn_ctx_sync(ctx));
    double ms = (now() - t0) * 1e3 / steps;
    SAY("step %-22s %8.3f ms/step  %8.1f img/s   (at %.1f s)\n", what, ms, B / ms * 1e3, now() - t00);
    return ms;
  };

  // ---- generic modes for the next rounds' short GPU calls ----------------------------------------------------------------------
  //   native_ab B H K rounds bench                      one line: ms/step of K steps after 5 warm-up steps, under the caller's environment
  //                                                     (for switches a process reads once: `env SWN_X=1 native_ab 32 256 40 0 bench`)
  //   native_ab B H K rounds prof                       per GEMM kernel family: launches, ms per step, TFLOP/s (HIP events, one stream)
  //   native_ab B H K rounds ab "A=1 B=2" "C=3" ...     alternating blocks of K steps in ONE process: no switch | each configuration,
  //                                                     `rounds` times (switches the library reads per launch / per step)
  auto plain_steps = [&](int steps) {
    SW(swn_ctx_sync(ctx));
    double t0 = now();
    for (int i = 0; i < steps; i++) SW(swn_model_step(m, labels, 1, ++seed));
    SW(swn_ctx_sync(ctx));
    return (now() - t0) * 1e3 / steps;
  };
  if (argc > 5 && !strcmp(argv[5], "truth")) {
    // K steps three ways from the same state, weight-arena hashes of each: (a) phase by phase with the device drained after every phase
    // and the second stream off (nothing can overtake anything: the reference outcome), (b) phase by phase, two streams, no draining,
    // (c) the fused swn_model_step.  All three must print the same hashes.
    // who owns an arena element: every parameter's LOGICAL elements marked through swn_model_param_set on the gradient arena; what stays
    // unmarked is layout padding (columns Co .. Npad, channels Ci .. Cip of the packed panels), which no state_dict ever sees
    std::vector<float> first[2][4]; std::vector<int> owner[2];
    for (int net = 0; net < 2; net++) {
      float* g = nullptr; size_t n = 0; SW(swn_model_arena(m, net, 1, &g, &n));
      dzero(g, n * 4);
      float* marks = (float*)dalloc(st.zeros_n * 4);
      for (size_t i = 0; i < st.p[net].size(); i++) {
        std::vector<float> h(st.p[net][i].n, (float)(i + 1)); h2d(marks, h.data(), h.size() * 4);
        SW(swn_model_param_set(m, net, 1, st.p[net][i].name.c_str(), marks));
      }
      SW(swn_ctx_sync(ctx));
      std::vector<float> h(n); d2h(h.data(), g, n * 4);
      owner[net].resize(n); size_t pad = 0;
      for (size_t i = 0; i < n; i++) { owner[net][i] = (int)h[i] - 1; pad += h[i] == 0.f; }
      SAY("truth net %d: arena of %zu elements, %zu of them layout padding\n", net, n, pad);
      dfree(marks); dzero(g, n * 4);
    }
    int run_no = 0;
    auto compare = [&](const char* what) {
      static const char* arena_name[4] = {"weight", "grad", "exp_avg", "exp_avg_sq"};
      for (int net = 0; net < 2; net++) for (int which = 0; which < 4; which++) {
        float* p = nullptr; size_t n = 0; SW(swn_model_arena(m, net, which, &p, &n));
        std::vector<float> h(n); d2h(h.data(), p, n * 4);
        if (run_no == 0) { first[net][which] = h; continue; }
        size_t diff = 0, diff_pad = 0, shown = 0;
        for (size_t i = 0; i < n; i++) if (memcmp(&h[i], &first[net][which][i], 4)) {
          diff++; const int o = owner[net][i]; diff_pad += o < 0;
          if (o < 0 && diff_pad <= 4) {
            size_t j = i; while (j > 0 && owner[net][j] < 0) j--;
            SAY("truth   %s net %d %s[%zu] (padding, %zu behind the last element of %s): %.9g against %.9g in the first run\n", what, net, arena_name[which], i, i - j,
                owner[net][j] >= 0 ? st.p[net][owner[net][j]].name.c_str() : "?", h[i], first[net][which][i]);
          }
          if (o >= 0 && shown++ < 12) SAY("truth   %s net %d %s[%zu] of %s: %.9g against %.9g in the first run\n", what, net, arena_name[which], i, st.p[net][o].name.c_str(), h[i], first[net][which][i]);
        }
        if (diff) SAY("truth   %s net %d %s arena: %zu elements differ from the first run, %zu of them layout padding\n", what, net, arena_name[which], diff, diff_pad);
      }
      run_no++;
    };
    auto run = [&](const char* what, int mode) {
      reset_state(m, st); SW(swn_ctx_sync(ctx));
      uint64_t sd = 1000;
      if (mode == 0) SW(swn_ctx_set_overlap(ctx, 0));
      for (int k = 0; k < K; k++) {
        ++sd;
        if (mode == 2) { SW(swn_model_step(m, labels, 1, sd)); continue; }
        SW(swn_model_forward(m, 1, sd)); if (mode == 0) SW(swn_ctx_sync(ctx));
        SW(swn_model_backward_D(m, labels[0], labels[1])); if (mode == 0) SW(swn_ctx_sync(ctx));
        SW(swn_model_optimizer_step(m, 1)); if (mode == 0) SW(swn_ctx_sync(ctx));
        SW(swn_model_backward_G(m, labels[2])); if (mode == 0) SW(swn_ctx_sync(ctx));
        SW(swn_model_optimizer_step(m, 0)); if (mode == 0) SW(swn_ctx_sync(ctx));
      }
      if (mode == 0) SW(swn_ctx_set_overlap(ctx, 1));
      uint64_t hh[2]; arena_hashes(ctx, m, hh);
      SAY("truth %-58s G %016llx D %016llx\n", what, (unsigned long long)hh[0], (unsigned long long)hh[1]);
      say_losses(m, "truth");
      compare(what);
    };
    run("phases, one stream, drained after every phase", 0);
    run("phases, two streams, not drained", 1);
    run("fused swn_model_step", 2);
    run("phases, one stream, drained after every phase (again)", 0);
    SW(swn_model_destroy(m)); SW(swn_ctx_destroy(ctx));
    return 0;
  }
  if (argc > 5 && !strcmp(argv[5], "trace")) {
    // FNV hashes of all four arenas of both networks after initialisation and after each of K training-mode steps, with the losses: two
    // processes that should compute the same bits print the same lines, and the first line that differs says where they part
    auto dump = [&](const char* what) {
      SW(swn_ctx_sync(ctx));
      for (int net = 0; net < 2; net++) {
        unsigned long long hh[4];
        for (int which = 0; which < 4; which++) {
          float* p = nullptr; size_t n = 0; SW(swn_model_arena(m, net, which, &p, &n));
          std::vector<float> h(n); d2h(h.data(), p, n * 4); hh[which] = fnv(h.data(), n * 4);
        }
        SAY("trace %-8s net %d  w %016llx  g %016llx  m %016llx  v %016llx\n", what, net, hh[0], hh[1], hh[2], hh[3]);
      }
    };
    // TRACE_PRELUDE: pieces of the truth mode's ownership prelude in front of the first step (which of them changes the outcome?)
    //   1 zero both gradient arenas through hipMemset   2 write marks into the gradient arenas (param_set which = 1), not cleared
    //   3 hipMalloc + hipFree of a parameter-sized buffer   4 = 2, then 1 (the truth mode's prelude)
    const int prelude = getenv("TRACE_PRELUDE") ? atoi(getenv("TRACE_PRELUDE")) : 0;
    if (prelude == 3) { void* q = dalloc(st.zeros_n * 4); dfree(q); }
    if (prelude == 2 || prelude == 4) {
      float* marks = (float*)dalloc(st.zeros_n * 4);
      for (int net = 0; net < 2; net++) for (size_t i = 0; i < st.p[net].size(); i++) {
        std::vector<float> h(st.p[net][i].n, (float)(i + 1)); h2d(marks, h.data(), h.size() * 4);
        SW(swn_model_param_set(m, net, 1, st.p[net][i].name.c_str(), marks));
      }
      SW(swn_ctx_sync(ctx)); dfree(marks);
    }
    if (prelude == 1 || prelude == 4) for (int net = 0; net < 2; net++) { float* g = nullptr; size_t n = 0; SW(swn_model_arena(m, net, 1, &g, &n)); dzero(g, n * 4); }
    SAY("trace prelude %d\n", prelude);
    dump("init");
    for (int k = 0; k < K; k++) {
      SW(swn_model_step(m, labels, 1, ++seed));
      char tag[32]; snprintf(tag, sizeof tag, "step%d", k + 1);
      dump(tag); say_losses(m, tag);
    }
    SW(swn_model_destroy(m)); SW(swn_ctx_destroy(ctx));
    return 0;
  }
  if (argc > 5 && !strcmp(argv[5], "hash")) {
    // K training-mode steps from the seeded state, then the FNV hashes of both weight arenas: two builds of the library that claim
    // to compute the same bits (a re-scheduled kernel, a cheaper instruction sequence for the same arithmetic) print the same line
    plain_steps(K);
    uint64_t hh[2]; arena_hashes(ctx, m, hh);
    say_losses(m, "hash");
    SAY("hash after %d steps: G %016llx D %016llx\n", K, (unsigned long long)hh[0], (unsigned long long)hh[1]);
    SW(swn_model_destroy(m)); SW(swn_ctx_destroy(ctx));
    return 0;
  }
  if (argc > 5 && !strcmp(argv[5], "bench")) {
    plain_steps(5);
    double ms = plain_steps(K);
    say_losses(m, "bench");
    SAY("bench %.3f ms/step %.1f img/s (B %d, %d x %d, %d steps)\n", ms, B / ms * 1e3, B, H, H, K);
    SW(swn_model_destroy(m)); SW(swn_ctx_destroy(ctx));
    return 0;
  }
  if (argc > 5 && !strcmp(argv[5], "phases")) {
    // wall time of the four phases of the step, each synchronised (un-profiled): forward | backward_D + optimizer_D | backward_G + optimizer_G
    plain_steps(5);
    double t[3] = {0, 0, 0};
    for (int i = 0; i < K; i++) {
      SW(swn_ctx_sync(ctx)); double a = now();
      SW(swn_model_forward(m, 1, ++seed)); SW(swn_ctx_sync(ctx)); double b = now();
      SW(swn_model_backward_D(m, labels[0], labels[1])); SW(swn_model_optimizer_step(m, 1)); SW(swn_ctx_sync(ctx)); double c = now();
      SW(swn_model_backward_G(m, labels[2])); SW(swn_model_optimizer_step(m, 0)); SW(swn_ctx_sync(ctx)); double d = now();
      t[0] += b - a; t[1] += c - b; t[2] += d - c;
    }
    {   // the forward pass again with nothing to refresh (weights unchanged since the last refresh): the chain alone
      double f2 = 0;
      for (int i = 0; i < K; i++) { SW(swn_ctx_sync(ctx)); double a = now(); SW(swn_model_forward(m, 1, ++seed)); SW(swn_ctx_sync(ctx)); f2 += now() - a; }
      SAY("phases: forward with current operands (no refresh) %.3f ms\n", f2 * 1e3 / K);
    }
    SAY("phases: forward %.3f ms | backward_D + AdamW(D) %.3f ms | backward_G + AdamW(G) %.3f ms | sum %.3f ms  (fused step: %.3f ms)\n", t[0] * 1e3 / K,
        t[1] * 1e3 / K, t[2] * 1e3 / K, (t[0] + t[1] + t[2]) * 1e3 / K, plain_steps(K));
    SW(swn_model_destroy(m)); SW(swn_ctx_destroy(ctx));
    return 0;
  }
  if (argc > 5 && !strcmp(argv[5], "host")) {
    // how long the HOST takes to enqueue one step (swn_model_step returning) against the GPU's time per step: host-bound or not
    plain_steps(5);
    SW(swn_ctx_sync(ctx));
    double enq = 0, t0 = now();
    for (int i = 0; i < K; i++) { const double a = now(); SW(swn_model_step(m, labels, 1, ++seed)); enq += now() - a; }
    const double t_enq_done = now() - t0;
    SW(swn_ctx_sync(ctx));
    const double total = now() - t0;
    SAY("host: %d steps: enqueue %.3f ms/step (all enqueued after %.3f ms), GPU done after %.3f ms = %.3f ms/step\n", K, enq * 1e3 / K, t_enq_done * 1e3,
        total * 1e3, total * 1e3 / K);
    // and from an idle GPU: one step enqueued after a sync (what a per-step host read-back of the losses sees)
    double s1 = 0;
    for (int i = 0; i < K; i++) { SW(swn_ctx_sync(ctx)); const double a = now(); SW(swn_model_step(m, labels, 1, ++seed)); SW(swn_ctx_sync(ctx)); s1 += now() - a; }
    SAY("host: synchronised every step: %.3f ms/step\n", s1 * 1e3 / K);
    SW(swn_model_destroy(m)); SW(swn_ctx_destroy(ctx));
    return 0;
  }
  if (argc > 5 && !strcmp(argv[5], "prof")) {
    // per-kernel-family timing of the implicit-GEMM launches with HIP events (swn_prof_*: what bench.py's roofline leg uses), in
    // order on one stream: K steps, "<kernel> <launches> <total ms> <flops>" -> ms per step and TFLOP/s.  Two seconds instead of a
    // rocprofv3 run when only the GEMM families are in question.
    plain_steps(3);
    SW(swn_ctx_set_overlap(ctx, 0));
    plain_steps(1);
    SW(swn_prof_reset()); SW(swn_prof_enable(1));
    double ms = plain_steps(K);
    SW(swn_prof_enable(0)); SW(swn_ctx_set_overlap(ctx, 1));
    int need = swn_prof_report(nullptr, 0);
    std::vector<char> buf((size_t)need + 16); swn_prof_report(buf.data(), need + 16);
    SAY("prof: %d steps in order on one stream, %.3f ms/step (events included)\n", K, ms);
    std::string rep(buf.data()); size_t p0 = 0; double tot = 0;
    while (p0 < rep.size()) {
      size_t e = rep.find('\n', p0); if (e == std::string::npos) e = rep.size();
      std::string line = rep.substr(p0, e - p0); p0 = e + 1;
      size_t a = line.rfind(' '); if (a == std::string::npos) continue;
      size_t b = line.rfind(' ', a - 1); size_t c = line.rfind(' ', b - 1);
      if (b == std::string::npos || c == std::string::npos) continue;
      double flops = atof(line.c_str() + a + 1), tms = atof(line.c_str() + b + 1); long launches = atol(line.c_str() + c + 1);
      tot += tms;
      SAY("prof %-56s %5ld launches/step %8.3f ms/step %8.1f us/launch %8.1f TFLOP/s\n", line.substr(0, c).c_str(), launches / K, tms / K,
          launches ? tms / launches * 1e3 : 0.0, tms > 0 ? flops / (tms * 1e-3) / 1e12 : 0.0);
    }
    SAY("prof GEMM families total %.3f ms/step\n", tot / K);
    SW(swn_model_destroy(m)); SW(swn_ctx_destroy(ctx));
    return 0;
  }
  if (argc > 5 && !strcmp(argv[5], "ab")) {
    std::vector<std::vector<std::pair<std::string, std::string>>> cfg(1);       // cfg[0] = no switch
    for (int a = 6; a < argc; a++) {
      std::vector<std::pair<std::string, std::string>> kv; std::string t = argv[a]; size_t p = 0;
      while (p < t.size()) {
        size_t e = t.find(' ', p); if (e == std::string::npos) e = t.size();
        std::string tok = t.substr(p, e - p); size_t q = tok.find('=');
        if (q != std::string::npos) kv.push_back({tok.substr(0, q), tok.substr(q + 1)});
        p = e + 1;
      }
      cfg.push_back(kv);
    }
    std::vector<double> sum(cfg.size(), 0.0);
    auto apply = [&](size_t c, bool on) { for (auto& kv : cfg[c]) { if (on) setenv(kv.first.c_str(), kv.second.c_str(), 1); else unsetenv(kv.first.c_str()); } };
    for (size_t c = 0; c < cfg.size(); c++) { apply(c, true); plain_steps(3); apply(c, false); }       // every configuration warm
    for (int r = 0; r < rounds; r++)
      for (size_t c = 0; c < cfg.size(); c++) {
        apply(c, true); double ms = plain_steps(K); apply(c, false); sum[c] += ms;
        SAY("ab round %d  %-40s %8.3f ms/step\n", r, c ? argv[5 + c] : "(default)", ms);
      }
    say_losses(m, "ab");
    for (size_t c = 0; c < cfg.size() && rounds; c++)
      SAY("ab mean   %-40s %8.3f ms/step  %+.3f\n", c ? argv[5 + c] : "(default)", sum[c] / rounds, (sum[c] - sum[0]) / rounds);
    SW(swn_model_destroy(m)); SW(swn_ctx_destroy(ctx));
    return 0;
  }

  // ---- 2. whole-step A/B -------------------------------------------------------------------------------------------------
  block("warm-up default", 1, 3);
  say_losses(m, "default");
  block("warm-up SWN_PC_MI=2", 2, 3);
  say_losses(m, "SWN_PC_MI=2");
  double s1 = 0, s2 = 0, s3 = 0;
  for (int r = 0; r < rounds; r++) { s1 += block("default", 1, K); s2 += block("SWN_PC_MI=2", 2, K); s3 += block("SWN_PC_MI=2 min 256", 2, K, "256"); }
  if (rounds) SAY("A/B mean over %d alternating blocks of %d steps: default %.3f ms  SWN_PC_MI=2 %.3f ms (%+.3f)  with SWN_PC_MI_MIN_TILES=256 %.3f ms (%+.3f)\n",
                  rounds, K, s1 / rounds, s2 / rounds, (s2 - s1) / rounds, s3 / rounds, (s3 - s1) / rounds);
  say_losses(m, "after A/B");
  {   // the route of one step under SWN_PC_MI=2: how many launches took the new tile
    setenv("SWN_PC_MI", "2", 1);
    swn_route_trace(1); SW(swn_model_step(m, labels, 1, ++seed)); SW(swn_ctx_sync(ctx)); swn_route_trace(0);
    std::string r = route(); int pcm = 0, pc = 0;
    for (size_t p = 0; (p = r.find("conv_fwd_pc", p)) != std::string::npos; p++) { if (r.compare(p, 12, "conv_fwd_pcm") == 0) pcm++; else pc++; }
    SAY("one step under SWN_PC_MI=2: %d launches on conv_fwd_pcm, %d on conv_fwd_pc\n", pcm, pc);
    setenv("SWN_PC_MI", "1", 1);
  }

  // ---- 2b. diagnostic (argv[5] == "diag"): where do swn_model_step and swn_model_step_dp part? -------------------------------------
  if (argc > 5 && !strcmp(argv[5], "diag")) {
    float* gw = nullptr; size_t gn = 0; SW(swn_model_weight_arena(m, 0, &gw, &gn));
    auto run = [&](const char* what, int steps, int dp, const char* stream_adamw) {
      if (stream_adamw) setenv("SWN_STREAM_ADAMW", stream_adamw, 1); else unsetenv("SWN_STREAM_ADAMW");
      reset_state(m, st);
      for (int i = 0; i < steps; i++) { if (dp) SW(swn_model_step_dp(m, labels, 1, 5000 + i, 0)); else SW(swn_model_step(m, labels, 1, 5000 + i)); }
      SW(swn_ctx_sync(ctx));
      std::vector<float> h(gn); d2h(h.data(), gw, gn * 4);
      float* dw = nullptr; size_t dn = 0; SW(swn_model_weight_arena(m, 1, &dw, &dn));
      std::vector<float> hd(dn); d2h(hd.data(), dw, dn * 4);
      SAY("diag %-34s G %016llx D %016llx  (at %.1f s)\n", what, (unsigned long long)fnv(h.data(), gn * 4), (unsigned long long)fnv(hd.data(), dn * 4), now() - t00);
      unsetenv("SWN_STREAM_ADAMW");
      return h;
    };
    auto diff = [&](const char* what, const std::vector<float>& a, const std::vector<float>& b) {
      size_t nd = 0, first = 0, last = 0; double worst = 0;
      for (size_t i = 0; i < a.size(); i++) if (memcmp(&a[i], &b[i], 4)) {
        if (!nd) first = i; last = i; nd++;
        double r = std::fabs((double)a[i] - b[i]) / (std::fabs((double)a[i]) + 1e-30); if (r > worst) worst = r;
        if (nd <= 6) SAY("   [%zu] %.9g vs %.9g\n", i, a[i], b[i]);
      }
      SAY("diag diff %-40s %zu of %zu elements differ (first %zu last %zu, worst rel %.2e)\n", what, nd, a.size(), first, last, worst);
    };
    auto noop = +[](const void* sb, void* rb, size_t, int, int, void*, void*) -> int { return sb == rb ? 0 : 1; };   // world 1: SUM in place = identity
    auto f1 = run("fused x1", 1, 0, nullptr);
    auto f1b = run("fused x1 again", 1, 0, nullptr);
    diff("fused x1 vs fused x1 again", f1, f1b);
    auto u1 = run("SWN_STREAM_ADAMW=0 x1", 1, 0, "0");
    diff("fused x1 vs one AdamW launch x1", f1, u1);
    SW(swn_ctx_attach_comm(ctx, (swn_allreduce_fn)noop, (void*)1, 1));
    auto d1 = run("step_dp x1 (identity callback)", 1, 1, nullptr);
    diff("fused x1 vs step_dp x1", f1, d1);
    auto d1b = run("step_dp x1 again", 1, 1, nullptr);
    diff("step_dp x1 vs again", d1, d1b);
    f1b.clear(); f1b.shrink_to_fit(); u1.clear(); u1.shrink_to_fit(); d1b.clear(); d1b.shrink_to_fit();
    auto f2 = run("fused x2", 2, 0, nullptr);
    auto d2 = run("step_dp x2 (identity callback)", 2, 1, nullptr);
    diff("fused x2 vs step_dp x2", f2, d2);
    SW(swn_ctx_attach_comm(ctx, nullptr, nullptr, 1));
    {   // the exchange buckets of the generator's arena
      int np = 0; SW(swn_model_backward_G_parts(m, &np));
      SW(swn_model_forward(m, 1, 9)); SW(swn_model_backward_D(m, labels[0], labels[1]));
      for (int part = 0; part < np; part++) { size_t off = 0, cnt = 0; SW(swn_model_backward_G_part(m, labels[2], part, &off, &cnt)); SAY("diag bucket %d: off %zu count %zu\n", part, off, cnt); }
      SW(swn_ctx_sync(ctx));
    }
    if (!getenv("NATIVE_AB_SKIP_RCCL")) {   // and with RCCL's own ncclAllReduce at world size 1
      const char* lib = getenv("NATIVE_AB_RCCL") ? getenv("NATIVE_AB_RCCL") : "/opt/rocm/lib/librccl.so.1";
      void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
      auto get_id = h ? (int (*)(NcclId*))dlsym(h, "ncclGetUniqueId") : nullptr;
      auto init = h ? (int (*)(void**, int, NcclId, int))dlsym(h, "ncclCommInitRank") : nullptr;
      auto allred = h ? (swn_allreduce_fn)dlsym(h, "ncclAllReduce") : nullptr;
      NcclId id; void* comm = nullptr;
      if (!get_id || !init || !allred || get_id(&id) != 0 || init(&comm, 1, id, 0) != 0) SAY("diag rccl: not available\n");
      else {
        SW(swn_ctx_attach_comm(ctx, allred, comm, 1));
        auto r2 = run("step_dp x2 (RCCL world 1)", 2, 1, nullptr);
        diff("fused x2 vs step_dp x2 (RCCL)", f2, r2);
        diff("step_dp x2 identity vs RCCL", d2, r2);
        SW(swn_ctx_attach_comm(ctx, nullptr, nullptr, 1));
      }
    }
    SW(swn_model_destroy(m)); SW(swn_ctx_destroy(ctx));
    SAY("native_ab diag done at %.1f s\n", now() - t00);
    return 0;
  }

  // ---- 3. library-owned exchange, RCCL at world size 1 ---------------------------------------------------------------------
  if (!getenv("NATIVE_AB_SKIP_RCCL")) {
    const char* lib = getenv("NATIVE_AB_RCCL") ? getenv("NATIVE_AB_RCCL") : "/opt/rocm/lib/librccl.so.1";
    void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { SAY("rccl: dlopen(%s) failed: %s\n", lib, dlerror()); }
    else {
      auto get_id = (int (*)(NcclId*))dlsym(h, "ncclGetUniqueId");
      auto init = (int (*)(void**, int, NcclId, int))dlsym(h, "ncclCommInitRank");
      auto allred = (swn_allreduce_fn)dlsym(h, "ncclAllReduce");
      auto destroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
      NcclId id; void* comm = nullptr; int rc;
      if (!get_id || !init || !allred) SAY("rccl: symbols missing\n");
      else if ((rc = get_id(&id)) != 0 || (rc = init(&comm, 1, id, 0)) != 0) SAY("rccl: init failed rc %d\n", rc);
      else {
        SAY("rccl communicator (world 1) up at %.1f s\n", now() - t00);
        uint64_t ha[2], hb[2];
        reset_state(m, st);
        for (int i = 0; i < 2; i++) SW(swn_model_step(m, labels, 1, 5000 + i));
        arena_hashes(ctx, m, ha); say_losses(m, "swn_model_step x2   ");
        SW(swn_ctx_attach_comm(ctx, allred, comm, 1));
        reset_state(m, st);
        for (int i = 0; i < 2; i++) SW(swn_model_step_dp(m, labels, 1, 5000 + i, 0));
        arena_hashes(ctx, m, hb); say_losses(m, "swn_model_step_dp x2");
        SAY("native exchange vs fused step after 2 steps: G arena %s (%016llx / %016llx)  D arena %s (%016llx / %016llx)\n",
            ha[0] == hb[0] ? "BIT-EQUAL" : "DIFFERENT", (unsigned long long)ha[0], (unsigned long long)hb[0],
            ha[1] == hb[1] ? "BIT-EQUAL" : "DIFFERENT", (unsigned long long)ha[1], (unsigned long long)hb[1]);
        SW(swn_ctx_sync(ctx));
        double t0 = now();
        for (int i = 0; i < K; i++) SW(swn_model_step_dp(m, labels, 1, ++seed, 0));
        SW(swn_ctx_sync(ctx));
        double ms_dp = (now() - t0) * 1e3 / K;
        SW(swn_ctx_attach_comm(ctx, nullptr, nullptr, 1));
        double ms_f = block("fused (after dp block)", 1, K);
        SAY("step_dp with RCCL world 1: %.3f ms/step against fused %.3f ms/step\n", ms_dp, ms_f);
        if (destroy) destroy(comm);
      }
    }
  }
  SW(swn_model_destroy(m));
  SW(swn_ctx_destroy(ctx));
  SAY("native_ab done at %.1f s\n", now() - t00);
  return 0;
}
