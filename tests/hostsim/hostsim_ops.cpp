// TEST INFRASTRUCTURE ONLY -- never linked into libswapnet_hip.so, never loaded by the
// swapnet_amd package.
//
// Plain-loop host implementation of swapnet_amd/csrc/ops.h.  Linked with the real
// engine.cpp / nets.cpp / texture.cpp / capi.cpp it yields libswapnet_hostsim.so, which lets
// the CPU-only CI (pytest -m "not gpu") check the engine's graph wiring, accumulate planner,
// weight packing / dgrad re-packing and loss plumbing against the oracle at small sizes
// without a GPU.  The loops are written independently of the HIP kernels (they do not share
// index math), so they also serve as a second opinion on the gather geometry.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../swapnet_amd/csrc/ops.h"

namespace swn {
// SWN_SIM_TIMES=1: wall time per simulated operator, printed at exit (where does a CPU-suite step go?)
struct SimTimes {
  struct Row { const char* name; double s; long calls; };
  std::vector<Row> rows;
  bool on = getenv("SWN_SIM_TIMES") != nullptr;
  int slot(const char* n) { for (size_t i = 0; i < rows.size(); ++i) if (rows[i].name == n) return (int)i; rows.push_back({n, 0.0, 0}); return (int)rows.size() - 1; }
  ~SimTimes() {
    if (!on) return;
    double tot = 0; for (auto& r : rows) tot += r.s;
    for (auto& r : rows) fprintf(stderr, "[sim-times] %-34s %8.3f s %7ld calls %5.1f %%\n", r.name, r.s, r.calls, 100.0 * r.s / (tot + 1e-30));
  }
};
static SimTimes g_sim_times;
struct SimTimer {
  int i = -1; double t0 = 0;
  static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
  explicit SimTimer(const char* n) { if (g_sim_times.on) { i = g_sim_times.slot(n); t0 = now(); } }
  ~SimTimer() { if (i >= 0) { g_sim_times.rows[i].s += now() - t0; g_sim_times.rows[i].calls++; } }
};
#define SIM_TIMED SimTimer sim_timer_(__func__)


void* dev_alloc(size_t bytes) { if (!bytes) bytes = 16; void* p = std::calloc(1, bytes); if (!p) throw Error(2, "hostsim: out of memory"); return p; }
void dev_free(void* p) { std::free(p); }
void dev_memset(Stream&, void* p, int v, size_t bytes) { SIM_TIMED; std::memset(p, v, bytes); }
void dev_copy(Stream&, void* d, const void* s, size_t b) { std::memcpy(d, s, b); }
void dev_upload(Stream&, void* d, const void* s, size_t b) { std::memcpy(d, s, b); }
void dev_store_small(Stream&, void* d, const void* s, size_t b) { std::memcpy(d, s, b); }
void dev_download(Stream&, void* d, const void* s, size_t b) { std::memcpy(d, s, b); }
void stream_sync(Stream&) {}
void* stream_create(int) { return nullptr; }
void stream_destroy(void*) {}
void device_check(int) {}
void* event_create() { return nullptr; }
void event_destroy(void*) {}
void event_record(void*, Stream&) {}
void stream_wait_event(Stream&, void*) {}
int is_device_build() { return 0; }
void* stream_create_current() { return nullptr; }
void graph_begin(Stream&) {}
void* graph_end(Stream&) { return nullptr; }
void graph_abort(Stream&) {}
void graph_launch(void*, Stream&) {}
void graph_destroy(void*) {}
void conv_force_naive(int) {}
void prof_enable(int) {}
void prof_reset() {}
int prof_report(char* buf, int len) { if (buf && len > 0) buf[0] = 0; return 0; }
void probe_mfma(Stream&, int, int, float*) { throw Error(3, "swn_probe_mfma measures the matrix pipe of a GPU: device build only"); }

static inline float actf(float v, int a) {
  switch (a) { case ACT_LRELU: return v > 0 ? v : 0.2f * v; case ACT_RELU: return v > 0 ? v : 0.f;
               case ACT_TANH: return std::tanh(v); default: return v; }
}
static inline float actg_in(float v, int a) {
  switch (a) { case ACT_LRELU: return v > 0 ? 1.f : 0.2f; case ACT_RELU: return v > 0 ? 1.f : 0.f;
               case ACT_TANH: { float t = std::tanh(v); return 1.f - t * t; } default: return 1.f; }
}
static inline float actg_out(float y, int a) {
  switch (a) { case ACT_LRELU: return y > 0 ? 1.f : 0.2f; case ACT_RELU: return y > 0 ? 1.f : 0.f;
               case ACT_TANH: return 1.f - y * y; default: return 1.f; }
}
static inline int srcc(int e, int ext, int mode, int ups) {
  if (mode == PAD_REFLECT) { if (e < 0) e = -e; else if (e >= ext) e = 2 * ext - 2 - e; }
  else if (e < 0 || e >= ext) return -1;
  return e >> ups;
}
static inline float* at(const TView& v, int n, int y, int x) { return v.p + ((size_t)(n * v.H + y) * v.W + x) * v.cs; }

static void conv_fwd_one(const ConvFwdArgs& a) {
  const TView& X = a.x; const Gather& g = a.g;
  const int He = X.H << g.ups, We = X.W << g.ups;
  const long total = (long)X.N * g.Ho * g.Wo;
#pragma omp parallel
  {
    std::vector<double> acc(a.Cout);
#pragma omp for schedule(static)
    for (long m = 0; m < total; ++m) {
      const int n = (int)(m / ((long)g.Ho * g.Wo)); const int rem = (int)(m - (long)n * g.Ho * g.Wo);
      const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
      std::fill(acc.begin(), acc.end(), 0.0);
      for (int kh = 0; kh < g.KH; ++kh)
        for (int kw = 0; kw < g.KW; ++kw) {
          const int sy = srcc(oy * g.stride - g.pad_t + kh, He, g.pad_mode, g.ups);
          const int sx = srcc(ox * g.stride - g.pad_l + kw, We, g.pad_mode, g.ups);
          if (sy < 0 || sx < 0) continue;
          const float* xp = at(X, n, sy, sx);
          const float* wp = a.w + (size_t)((kh * g.KW + kw) * X.C) * a.Npad;
          for (int ci = 0; ci < X.C; ++ci) {
            const float xv = xp[ci];
            if (xv == 0.f) continue;
            const float* wr = wp + (size_t)ci * a.Npad;
            for (int co = 0; co < a.Cout; ++co) acc[co] += (double)xv * wr[co];
          }
        }
      float* yp = at(a.y, n, oy * a.om.ymul + a.om.yoff, ox * a.om.xmul + a.om.xoff);
      for (int co = 0; co < a.Cout; ++co) {
        float v = (float)acc[co];
        if (a.bias) v += a.bias[co];
        v = actf(v, a.act);
        if (a.accumulate) v += yp[co];
        yp[co] = v;
      }
    }
  }
}
template <class Args>
static void take_phase(Args& c, int ph) {      // sub-pixel phase mode (ops.h)
  if (!c.phases) return;
  if (c.phases != 4 || c.batch > 1 || c.om.ymul != 2 || c.om.xmul != 2) throw Error(1, "conv: bad sub-pixel phase launch");
  c.g.pad_t -= ph >> 1; c.g.pad_l -= ph & 1; c.om.yoff = ph >> 1; c.om.xoff = ph & 1;
  c.phases = 0;
}
// fused folded-tail launch (ops.h tail4) evaluated phase by phase
template <class Args>
static Args tail_phase(const Args& a, int ph) {
  if (a.g.KH != 3 || a.g.KW != 3 || a.g.stride != 1 || a.g.pad_t != 1 || a.g.pad_l != 1 || a.om.ymul != 2 || a.om.xmul != 2)
    throw Error(1, "conv: bad tail4 launch");
  Args c = a;
  c.tail4 = 0;
  c.g.KH = 2 + (ph >> 1); c.g.KW = 2 + (ph & 1);
  c.om.yoff = ph >> 1; c.om.xoff = ph & 1;
  return c;
}
static size_t tail_panel(int ph, int xC, int Npad) {
  static const int pre[4] = {0, 4, 10, 16};
  return (size_t)pre[ph] * xC * Npad;
}
// ---- pre-cut weight operands (ops.h conv_precut_*).  The simulator multiplies in plain fp32 / fp64, so it has no use for fp16 planes;
// it keeps the PLUMBING of the product honest instead: a "panel" here is the fp32 operand itself (a two-plane panel has exactly fp32's
// 4 bytes per element) in [K][tiles * bn] order plus the 16-byte trailer, written by the same producers and consumed through the same
// ConvFwdArgs::wpc / wpc_bn / wpc_bs fields, so the engine's panel sizes, offsets, batch strides and refresh order are exercised in CI.
// A consumer checks the trailer's magic (a wrong offset / stride / tile shows up as an error, not as a silently different result).
static const int kPanelMagic = 0x5EC07E00;
static int sim_tile_for(int Npad) { return Npad <= 64 ? 64 : ((Npad > 128 && Npad <= 192) ? 192 : 128); }
int conv_fwd_stat_chunk(int, int, int, int, int) { return 0; }      // (the epilogue statistics are the device kernel's: never offered here)
int conv_precut_tile(int xC, int Npad) {
  const char* e = getenv("SWN_PRECUT");
  if ((e && atoi(e) == 0) || xC % 16 || Npad <= 32) return 0;
  return sim_tile_for(Npad);
}
int conv_precut_planes() { return 2; }
// Pair-form planes (ops.h wino_input_transform): a device storage format -- by default the simulator's planes stay fp32 and no
// pair form is announced.  SWN_SIM_PAIR=1 announces it with the device's predicates and ROUNDS the planes the way the device
// stores them (x 2^k -> h = fp16 RN, l = fp16 RN of the residual, k from the gain bound x the input's amax slot), so the CPU
// suite sees the engine's pair plumbing (which slot bounds which transform) and the precision the bound leaves; the GEMMs then
// multiply the rounded fp32 values.
static bool sim_pair_on() { const char* e = getenv("SWN_SIM_PAIR"); return e && atoi(e) != 0; }
// ---- the 16-bit operand formats of the device's ring kernels, as a rounding model (SWN_SIM_PAIR=1 only) ------------------------------
// An operand element x enters the MFMAs as h + l, two fp16 values of x 2^k, k from the tensor's amax (top 2^12 for activations /
// gradients, 2^10 for weights: conv_gemm.hip PC_TOP_A / PC_TOP_B).  Pre-cut weights and pair-form planes round h to nearest; an
// activation cut in the loop truncates h (split8h), the weight gradient's dY operand rounds it (split8h_rn).  The simulator applies the
// same cut to the fp32 value and multiplies the result -- what CPU CI then sees is the 22-bit operand arithmetic with the engine's
// own choice of scales (slots, bounds, hand-overs), not the accumulation order of the MFMAs.
float sim_slot_max(const float* slot);
static float sim_f16_rn(float x);
static float sim_f16_trunc(float x) {
  if (x == 0.f || !std::isfinite(x)) return x;
  int e;
  std::frexp(std::fabs(x), &e);
  const int ue = std::max(e - 11, -24);
  return std::ldexp((float)std::trunc(std::ldexp((double)x, -ue)), ue);
}
static int sim_scale_exp(float amax, int top) {                 // as conv_gemm.hip scale_exp / wino.hip wino_scale_exp
  if (!(amax > 0.f) || amax > 3.0e38f) return 0;
  int e;
  std::frexp(amax, &e);                                          // amax = m 2^e, m in [0.5, 1): binary exponent e - 1
  return std::max(-100, std::min(100, top - 1 - (e - 1)));
}
static inline float sim_cut(float v, int k, bool trunc_h) {
  const float x = std::ldexp(v, k);
  const float h = trunc_h ? sim_f16_trunc(x) : sim_f16_rn(x);
  return std::ldexp(h + sim_f16_rn(x - h), -k);
}
static void sim_cut_dense(float* f, size_t n, int k, bool trunc_h) {
  for (size_t i = 0; i < n; ++i) f[i] = sim_cut(f[i], k, trunc_h);
}
bool wino_pair_planes() { return sim_pair_on(); }
bool conv_fwd_takes_pairs(int xC, int Npad) { return sim_pair_on() && conv_precut_tile(xC, Npad) != 0; }
bool conv_wgrad_takes_pairs(size_t T, int K, int Npad) {
  if (!sim_pair_on() || Npad <= 32 || K % 4 || Npad % 4 || T % 16 || T * (size_t)std::max(K, Npad) * 4 >= ((size_t)1 << 31)) return false;
  const int bmk = Npad > 64 ? 128 : 256;
  const double fillf = (double)K / (double)(((K + bmk - 1) / bmk) * bmk);
  return fillf > 0.8 || (bmk == 128 && fillf >= 0.75);
}
const float* conv_precut_amax(Stream&, const float*, size_t, int, int, size_t) { return nullptr; }
static size_t sim_np(int Npad, int bn) { return (size_t)((Npad + bn - 1) / bn) * bn; }
size_t conv_precut_elems(int K, int Npad, int bn) { return (size_t)(K / 16) * ((Npad + bn - 1) / bn) * 4 * bn * 8 + 8; }
static void sim_trailer(uint16_t* panel, int K, int Npad, int bn) {
  reinterpret_cast<int*>(reinterpret_cast<float*>(panel) + (size_t)K * sim_np(Npad, bn))[0] = kPanelMagic + bn;
}
static void sim_store_panel(const float* w, int K, int Npad, int bn, uint16_t* out) {
  const size_t NP = sim_np(Npad, bn);
  float* f = reinterpret_cast<float*>(out);
  if (NP == (size_t)Npad) memcpy(f, w, (size_t)K * Npad * sizeof(float));
  else
    for (int k = 0; k < K; ++k)
      for (size_t n = 0; n < NP; ++n) f[(size_t)k * NP + n] = n < (size_t)Npad ? w[(size_t)k * Npad + n] : 0.f;
  sim_trailer(out, K, Npad, bn);
}
// amax_io (ops.h conv_precut): the simulator keeps the "partial maxima" of the last pass in one static slot -- exactly the lifetime the
// device's stream scratch gives them -- and CHECKS a handed-in pointer against the source it is now claimed to bound
static float g_sim_wamax[AMAX_SLOT];
static int g_sim_wamax_owner = 0, g_sim_wamax_handed = -1;   // bumped by every pass: a stale hand-over (another pass in between) is an engine bug
static void sim_weight_amax(const float* src, size_t rows, int C, int batch, size_t bs, const float** amax_io, const char* what) { SIM_TIMED;
  float actual = 0.f;
  for (int b = 0; b < batch; ++b) {
    const float* sp = src + (size_t)b * bs;
    const long n = (long)(rows * (size_t)C);
    float m = 0.f;
#pragma omp parallel for reduction(max : m)
    for (long i = 0; i < n; ++i) m = std::max(m, std::fabs(sp[i]));
    actual = std::max(actual, m);
  }
  if (amax_io && *amax_io) {
    if (*amax_io != g_sim_wamax || g_sim_wamax_handed != g_sim_wamax_owner)
      throw Error(1, std::string("hostsim ") + what + ": handed-in weight amax is not the last pass's (another pass reused the scratch)");
    const float have = g_sim_wamax[0];
    if (!(have >= actual) || (actual > 0.f && have > 4096.f * actual))
      throw Error(1, std::string("hostsim ") + what + ": handed-in weight amax " + std::to_string(have) + " does not bound the source (amax " + std::to_string(actual) + ")");
    return;
  }
  for (int i = 0; i < AMAX_SLOT; ++i) g_sim_wamax[i] = 0.f;
  g_sim_wamax[0] = actual;
  ++g_sim_wamax_owner;
  if (amax_io) { *amax_io = g_sim_wamax; g_sim_wamax_handed = g_sim_wamax_owner; }
}
void conv_precut(Stream&, const float* w, int K, int Npad, int bn, int batch, size_t w_bs, uint16_t* out, const float** amax_io) { SIM_TIMED;
  if (K % 16 || (bn != 64 && bn != 128 && bn != 192)) throw Error(1, "conv_precut: K must be a multiple of 16, tile 64, 128 or 192");
  sim_weight_amax(w, (size_t)K, Npad, batch, w_bs, amax_io, "conv_precut");
  const size_t pe = conv_precut_elems(K, Npad, bn);
  for (int z = 0; z < batch; ++z) {
    sim_store_panel(w + (size_t)z * w_bs, K, Npad, bn, out + (size_t)z * pe);
    if (sim_pair_on())           // two fp16 planes of w 2^kB, one scale for all panels of the launch (conv_precut_kernel)
      sim_cut_dense(reinterpret_cast<float*>(out + (size_t)z * pe), (size_t)K * sim_np(Npad, bn), sim_scale_exp(g_sim_wamax[0], 10), false);
  }
}
static void wino_filter_transform_strided(int m, int r, const WShape& w, int mode, const float* packed, float* U, size_t total);
void wino_filter_transform_pc(Stream&, int m, int r, const WShape& w, int mode, const float* packed, int bn, uint16_t* out,
                              size_t panel_elems, const float** amax_io) { SIM_TIMED;
  sim_weight_amax(packed, (size_t)r * r * w.Cip, w.Npad, 1, 0, amax_io, "wino_filter_transform_pc");
  const int K = mode == 0 ? w.Cip : w.Npad, Nn = mode == 0 ? w.Npad : w.Cip;
  if (K % 16 || (bn != 64 && bn != 128)) throw Error(1, "wino_filter_transform_pc: K must be a multiple of 16, tile 64 or 128");
  if (panel_elems != conv_precut_elems(K, Nn, bn)) throw Error(1, "wino_filter_transform_pc: panel stride does not match conv_precut_elems");
  const int A = m + r - 1, P = A * A;
  if (sim_np(Nn, bn) == (size_t)Nn) {             // dense panels: transform straight into them (plane stride = the panel stride)
    wino_filter_transform_strided(m, r, w, mode, packed, reinterpret_cast<float*>(out), panel_elems / 2);
    for (int p = 0; p < P; ++p) {
      sim_trailer(out + (size_t)p * panel_elems, K, Nn, bn);
      // (the scale of the transformed planes comes from the amax of the PACKED filter: |G g G^T| <= |g|max for the 6-point forms)
      if (sim_pair_on()) sim_cut_dense(reinterpret_cast<float*>(out + (size_t)p * panel_elems), (size_t)K * Nn, sim_scale_exp(g_sim_wamax[0], 10), false);
    }
    return;
  }
  std::vector<float> U((size_t)P * K * Nn);
  wino_filter_transform_strided(m, r, w, mode, packed, U.data(), (size_t)K * Nn);
  for (int p = 0; p < P; ++p) {
    sim_store_panel(U.data() + (size_t)p * K * Nn, K, Nn, bn, out + (size_t)p * panel_elems);
    if (sim_pair_on()) sim_cut_dense(reinterpret_cast<float*>(out + (size_t)p * panel_elems), (size_t)K * sim_np(Nn, bn), sim_scale_exp(g_sim_wamax[0], 10), false);
  }
}

void sim_slot_check(const float* slot, const float* x, size_t rows, int C, size_t rs, int batch, size_t bs, const char* what);
void sim_fold_view(float* slot, const TView& v);
void conv_fwd(Stream& s, const ConvFwdArgs& a) {
  if (a.y_amax) {                   // fold the amax of the output view once the launch below has written it
    ConvFwdArgs c = a;
    c.y_amax = nullptr;
    conv_fwd(s, c);
    sim_fold_view(a.y_amax, a.y);
    return;
  }
  SIM_TIMED;
  if (a.x_amax) {
    const int nbb = a.phases ? 1 : (a.batch > 0 ? a.batch : 1);
    sim_slot_check(a.x_amax, a.x.p, (size_t)a.x.N * a.x.H * a.x.W, a.x.C, (size_t)a.x.cs, nbb, a.x_bs, "conv_fwd");
  }
  if (a.wpc) {
    // decode the panels back into the fp32 operand the loops below read
    if (a.tail4) throw Error(1, "hostsim conv_fwd: pre-cut operand on a tail4 launch");
    const int nb = a.phases ? a.phases : (a.batch > 0 ? a.batch : 1);
    const int K = a.g.KH * a.g.KW * a.x.C, bn = a.wpc_bn;
    if (bn != sim_tile_for(a.Npad)) throw Error(1, "hostsim conv_fwd: operand pre-cut for another column tile");
    const size_t NP = sim_np(a.Npad, bn), pe = conv_precut_elems(K, a.Npad, bn);
    if (nb > 1 && a.wpc_bs != pe) throw Error(1, "hostsim conv_fwd: panel stride does not match conv_precut_elems");
    for (int z = 0; z < nb; ++z) {
      const float* f = reinterpret_cast<const float*>(a.wpc + (size_t)z * a.wpc_bs);
      if (reinterpret_cast<const int*>(f + (size_t)K * NP)[0] != kPanelMagic + bn)
        throw Error(1, "hostsim conv_fwd: pre-cut panel trailer not found (wrong offset, stride, K or tile): K " + std::to_string(K) + " Npad " +
                           std::to_string(a.Npad) + " bn " + std::to_string(bn) + " panel " + std::to_string(z) + "/" + std::to_string(nb) + " KH " +
                           std::to_string(a.g.KH) + " stride " + std::to_string(a.g.stride) + " xC " + std::to_string(a.x.C) + " found " +
                           std::to_string(reinterpret_cast<const int*>(f + (size_t)K * NP)[0]));
    }
    ConvFwdArgs c = a;
    c.wpc = nullptr; c.x_amax = nullptr;
    // SWN_SIM_PAIR: the activation operand of the two-plane ring kernel, cut in its loop (h by truncation) with the scale of the slot
    // it was handed or of its own amax pass -- unless it arrives in pair form (its producer rounded it already)
    std::vector<float> xf;
    if (sim_pair_on() && !a.x_pair_k) {
      const int nbx = a.phases ? 1 : nb;
      const size_t rows = (size_t)a.x.N * a.x.H * a.x.W;
      float am = 0.f;
      if (a.x_amax) am = sim_slot_max(a.x_amax);
      else
        for (int z = 0; z < nbx; ++z)
          for (size_t r = 0; r < rows; ++r)
            for (int ch = 0; ch < a.x.C; ++ch) am = std::max(am, std::fabs(a.x.p[(size_t)z * a.x_bs + r * a.x.cs + ch]));
      const int kA = sim_scale_exp(am, 12);
      xf.resize((size_t)nbx * rows * a.x.C);
      for (int z = 0; z < nbx; ++z)
        for (size_t r = 0; r < rows; ++r)
          for (int ch = 0; ch < a.x.C; ++ch)
            xf[((size_t)z * rows + r) * a.x.C + ch] = sim_cut(a.x.p[(size_t)z * a.x_bs + r * a.x.cs + ch], kA, true);
      c.x.p = xf.data(); c.x.cs = a.x.C;
      if (!a.phases) c.x_bs = rows * (size_t)a.x.C;          // (the phases of one launch read the same input)
    }
    c.x_pair_k = nullptr;
    std::vector<float> wf;
    if (NP == (size_t)a.Npad) {                     // dense panel = the fp32 operand itself
      c.w = reinterpret_cast<const float*>(a.wpc); c.w_bs = a.wpc_bs / 2;
    } else {
      wf.resize((size_t)nb * K * a.Npad);
      for (int z = 0; z < nb; ++z) {
        const float* f = reinterpret_cast<const float*>(a.wpc + (size_t)z * a.wpc_bs);
        for (int k = 0; k < K; ++k)
          for (int n = 0; n < a.Npad; ++n) wf[((size_t)z * K + k) * a.Npad + n] = f[(size_t)k * NP + n];
      }
      c.w = wf.data(); c.w_bs = (size_t)K * a.Npad;
    }
    conv_fwd(s, c);
    return;
  }
  if (!a.w) throw Error(1, "hostsim conv_fwd: no weight operand");
  if (a.tail4) {
    for (int ph = 0; ph < 4; ++ph) {
      ConvFwdArgs c = tail_phase(a, ph);
      c.w = a.w + tail_panel(ph, a.x.C, a.Npad);
      conv_fwd(s, c);
    }
    return;
  }
  const int nb = a.phases ? a.phases : (a.batch > 0 ? a.batch : 1);
  if (route_on()) {
    char nm[128];
    snprintf(nm, sizeof nm, "sim_conv_fwd[M%d,N%d,K%d,b%d]", a.x.N * a.g.Ho * a.g.Wo, a.Cout, a.g.KH * a.g.KW * a.x.C, nb);
    route_note(nm);
  }
  // batched Winograd planes have few rows each: spread the planes over the threads instead
#pragma omp parallel for schedule(dynamic, 1) if (nb >= 8)
  for (int b = 0; b < nb; ++b) {
    ConvFwdArgs c = a;
    take_phase(c, b);
    c.x.p = a.x.p + b * a.x_bs; c.w = a.w + b * a.w_bs; c.y.p = a.y.p + b * a.y_bs;
    conv_fwd_one(c);
  }
}
void conv_fwd_naive(Stream& s, const ConvFwdArgs& a) { conv_fwd(s, a); }

static void conv_wgrad_one(const ConvWgradArgs& a) {
  const TView& X = a.x; const Gather& g = a.g;
  const int He = X.H << g.ups, We = X.W << g.ups;
  const int taps = g.KH * g.KW;
  // each thread owns whole (tap, ci) rows of dW: no cross-thread reduction, fp64 accumulation
#pragma omp parallel for collapse(2) schedule(dynamic, 8)
  for (int tap = 0; tap < taps; ++tap)
    for (int ci = 0; ci < X.C; ++ci) {
      const int kh = tap / g.KW, kw = tap - kh * g.KW;
      std::vector<double> acc(a.Npad, 0.0);
      for (int n = 0; n < X.N; ++n)
        for (int oy = 0; oy < g.Ho; ++oy) {
          const int sy = srcc(oy * g.stride - g.pad_t + kh, He, g.pad_mode, g.ups);
          if (sy < 0) continue;
          for (int ox = 0; ox < g.Wo; ++ox) {
            const int sx = srcc(ox * g.stride - g.pad_l + kw, We, g.pad_mode, g.ups);
            if (sx < 0) continue;
            const float xv = at(X, n, sy, sx)[ci];
            if (xv == 0.f) continue;
            const float* dp = at(a.dy, n, oy * a.om.ymul + a.om.yoff, ox * a.om.xmul + a.om.xoff);
            for (int co = 0; co < a.Cout; ++co) acc[co] += (double)xv * dp[co];
          }
        }
      float* o = a.dw + (size_t)(tap * X.C + ci) * a.Npad;
      for (int co = 0; co < a.Npad; ++co) o[co] = (float)acc[co];
    }
}
void conv_wgrad(Stream& s, const ConvWgradArgs& a) { SIM_TIMED;
  if (a.x_amax || a.dy_amax) {
    const int nbb = a.phases ? 1 : (a.batch > 0 ? a.batch : 1);
    sim_slot_check(a.x_amax, a.x.p, (size_t)a.x.N * a.x.H * a.x.W, a.x.C, (size_t)a.x.cs, nbb, a.x_bs, "conv_wgrad x");
    sim_slot_check(a.dy_amax, a.dy.p, (size_t)a.dy.N * a.dy.H * a.dy.W, a.dy.C, (size_t)a.dy.cs, nbb, a.dy_bs, "conv_wgrad dy");
  }
  if (a.tail4) {
    for (int ph = 0; ph < 4; ++ph) {
      ConvWgradArgs c = tail_phase(a, ph);
      c.dw = a.dw + tail_panel(ph, a.x.C, a.Npad);
      conv_wgrad(s, c);
    }
    return;
  }
  const int nb = a.phases ? a.phases : (a.batch > 0 ? a.batch : 1);
  if (route_on()) {
    char nm[128];
    snprintf(nm, sizeof nm, "sim_conv_wgrad[M%d,N%d,K%d,b%d]", a.x.N * a.g.Ho * a.g.Wo, a.Cout, a.g.KH * a.g.KW * a.x.C, nb);
    route_note(nm);
  }
  // SWN_SIM_PAIR: the ring kernel's two fp16 planes of both operands (x cut by truncation, dY by rounding; pair-form operands were
  // rounded by their producers).  Which launches the device sends to that kernel is approximated by its main condition (Npad > 32).
  if (sim_pair_on() && a.Npad > 32 && a.x.C % 4 == 0 && a.dy.C % 4 == 0 && !(a.x_pair_k && a.dy_pair_k)) {
    ConvWgradArgs c = a;
    const int nbx = a.phases ? 1 : nb;
    auto cut_copy = [&](const TView& v, size_t bs, const float* slot, bool trunc_h, std::vector<float>& store, TView& out, size_t& out_bs) {
      const size_t rows = (size_t)v.N * v.H * v.W;
      float am = 0.f;
      if (slot) am = sim_slot_max(slot);
      else
        for (int z = 0; z < nbx; ++z)
          for (size_t r = 0; r < rows; ++r)
            for (int ch = 0; ch < v.C; ++ch) am = std::max(am, std::fabs(v.p[(size_t)z * bs + r * v.cs + ch]));
      const int k = sim_scale_exp(am, 12);
      store.resize((size_t)nbx * rows * v.C);
      for (int z = 0; z < nbx; ++z)
        for (size_t r = 0; r < rows; ++r)
          for (int ch = 0; ch < v.C; ++ch) store[((size_t)z * rows + r) * v.C + ch] = sim_cut(v.p[(size_t)z * bs + r * v.cs + ch], k, trunc_h);
      out = v; out.p = store.data(); out.cs = v.C;
      if (!a.phases) out_bs = rows * (size_t)v.C;
    };
    std::vector<float> xs, ds;
    if (!a.x_pair_k) cut_copy(a.x, a.x_bs, a.x_amax, true, xs, c.x, c.x_bs);
    if (!a.dy_pair_k) cut_copy(a.dy, a.dy_bs, a.dy_amax, false, ds, c.dy, c.dy_bs);
    c.x_pair_k = c.dy_pair_k = reinterpret_cast<const int*>(&c);      // (marks "already cut" for the recursion below)
    c.x_amax = c.dy_amax = nullptr;
    conv_wgrad(s, c);
    return;
  }
#pragma omp parallel for schedule(dynamic, 1) if (nb >= 8)
  for (int b = 0; b < nb; ++b) {
    ConvWgradArgs c = a;
    take_phase(c, b);
    c.x.p = a.x.p + b * a.x_bs; c.dy.p = a.dy.p + b * a.dy_bs; c.dw = a.dw + b * a.dw_bs;
    conv_wgrad_one(c);
  }
}
void conv_wgrad_naive(Stream& s, const ConvWgradArgs& a) { conv_wgrad(s, a); }

// Winograd F(m x m, r x r) straight from the transform matrices: (2,3), (4,3), (3,4)
struct WinoMats { int m, r, A; const float* BT; const float* G; const float* AT; };
static const float kBT2[16] = {1, 0, -1, 0, 0, 1, 1, 0, 0, -1, 1, 0, 0, 1, 0, -1};
static const float kG2[12] = {1, 0, 0, 0.5f, 0.5f, 0.5f, 0.5f, -0.5f, 0.5f, 0, 0, 1};
static const float kAT2[8] = {1, 1, 1, 0, 0, 1, -1, -1};
static const float kBT6[36] = {4, 0, -5, 0, 1, 0, 0, -4, -4, 1, 1, 0, 0, 4, -4, -1, 1, 0,
                               0, -2, -1, 2, 1, 0, 0, 2, -1, -2, 1, 0, 0, 4, 0, -5, 0, 1};
static const float kG43[18] = {1.f / 4, 0, 0, -1.f / 6, -1.f / 6, -1.f / 6, -1.f / 6, 1.f / 6, -1.f / 6,
                               1.f / 24, 1.f / 12, 1.f / 6, 1.f / 24, -1.f / 12, 1.f / 6, 0, 0, 1};
static const float kAT43[24] = {1, 1, 1, 1, 1, 0, 0, 1, -1, 2, -2, 0, 0, 1, 1, 4, 4, 0, 0, 1, -1, 8, -8, 1};
static const float kG34[24] = {1.f / 4, 0, 0, 0, -1.f / 6, -1.f / 6, -1.f / 6, -1.f / 6, -1.f / 6, 1.f / 6, -1.f / 6, 1.f / 6,
                               1.f / 24, 1.f / 12, 1.f / 6, 1.f / 3, 1.f / 24, -1.f / 12, 1.f / 6, -1.f / 3, 0, 0, 0, 1};
static const float kAT34[18] = {1, 1, 1, 1, 1, 0, 0, 1, -1, 2, -2, 0, 0, 1, 1, 4, 4, 1};
static const float kBT42[25] = {0.5f, -1, -0.5f, 1, 0, 0, -0.5f, 0.5f, 1, 0, 0, 0.5f, -1.5f, 1, 0, 0, -1, 0, 1, 0, 0, 0.5f, -1, -0.5f, 1};
static const float kG42[10] = {2, 0, 1, 1, -1.f / 3, 1.f / 3, -8.f / 3, -4.f / 3, 0, 1};
static const float kAT42[20] = {1, 1, 1, 1, 0, 0, 1, -1, 0.5f, 0, 0, 1, 1, 0.25f, 0, 0, 1, -1, 0.125f, 1};
static WinoMats wino_mats(int m, int r) {
  if (m == 4 && r == 2) return {4, 2, 5, kBT42, kG42, kAT42};
  if (m == 2 && r == 3) return {2, 3, 4, kBT2, kG2, kAT2};
  if (m == 4 && r == 3) return {4, 3, 6, kBT6, kG43, kAT43};
  if (m == 3 && r == 4) return {3, 4, 6, kBT6, kG34, kAT34};
  throw Error(1, "winograd: supported forms are F(2,3), F(4,3) and F(3,4)");
}
static void wino_input_transform_impl(int m, int r, const TView& x, int pad, int pad_mode, int Th, int Tw, float* V) {
  const WinoMats wm = wino_mats(m, r);
  const int A = wm.A;
  const size_t T = (size_t)x.N * Th * Tw;
#pragma omp parallel for collapse(3)
  for (int n = 0; n < x.N; ++n) for (int ty = 0; ty < Th; ++ty) for (int tx = 0; tx < Tw; ++tx) {
    const size_t tile = ((size_t)n * Th + ty) * Tw + tx;
    for (int c = 0; c < x.C; ++c) {
      float d[6][6], t[6][6];
      for (int a = 0; a < A; ++a) for (int b = 0; b < A; ++b) {
        const int sy = srcc(m * ty - pad + a, x.H, pad_mode, 0), sx = srcc(m * tx - pad + b, x.W, pad_mode, 0);
        d[a][b] = (sy >= 0 && sx >= 0) ? at(x, n, sy, sx)[c] : 0.f;
      }
      for (int a = 0; a < A; ++a) for (int b = 0; b < A; ++b) { float s = 0; for (int k = 0; k < A; ++k) s += wm.BT[a * A + k] * d[k][b]; t[a][b] = s; }
      for (int a = 0; a < A; ++a) for (int b = 0; b < A; ++b) { float s = 0; for (int k = 0; k < A; ++k) s += t[a][k] * wm.BT[b * A + k];
        V[((size_t)(a * A + b) * T + tile) * x.C + c] = s; }
    }
  }
}
static void wino_filter_transform_strided(int m, int r, const WShape& w, int mode, const float* packed, float* U, size_t total) {
  // total = floats between consecutive transform-point planes (K * Nn when dense)
  // Weight-sized work (36 planes of 1024 x 1024 for a resblock conv, both directions, every step): it was 1/3 of the simulator's
  // step time as a scalar loop with stride-Npad reads in the transposed modes.  Same arithmetic per element, in the same order
  // (bit-identical results); the tap planes of the transposed modes are transposed once (blocked), and 64 columns go through the
  // two small products together so that the compiler can keep them in vector registers.
  const WinoMats wm = wino_mats(m, r);
  const int A = wm.A, R = wm.r;
  const int K = mode == 0 ? w.Cip : w.Npad, Nn = mode == 0 ? w.Npad : w.Cip;
  std::vector<float> tr;
  const float* src = packed;                       // [tap][K][Nn] from here on
  if (mode != 0) {
    tr.resize((size_t)R * R * K * Nn);
    const int TB = 32;
#pragma omp parallel for collapse(3)
    for (int t = 0; t < R * R; ++t) for (int k0 = 0; k0 < K; k0 += TB) for (int n0 = 0; n0 < Nn; n0 += TB) {
      const int a = t / R, b = t % R;
      const int st = mode == 1 ? (R - 1 - a) * R + (R - 1 - b) : t;
      const float* sp = packed + (size_t)st * w.Cip * w.Npad;         // [n][k]
      float* dp = tr.data() + (size_t)t * K * Nn;                     // [k][n]
      const int k1 = std::min(K, k0 + TB), n1 = std::min(Nn, n0 + TB);
      for (int n = n0; n < n1; ++n) for (int k = k0; k < k1; ++k) dp[(size_t)k * Nn + n] = sp[(size_t)n * w.Npad + k];
    }
    src = tr.data();
  }
  const int NB = 64;
#pragma omp parallel for
  for (int k = 0; k < K; ++k) for (int n0 = 0; n0 < Nn; n0 += NB) {
    const int nb = std::min(NB, Nn - n0);
    float g[4][4][NB], t[6][4][NB];
    for (int a = 0; a < R; ++a) for (int b = 0; b < R; ++b) {
      const float* sp = src + ((size_t)(a * R + b) * K + k) * Nn + n0;
      for (int j = 0; j < nb; ++j) g[a][b][j] = sp[j];
    }
    for (int a = 0; a < A; ++a) for (int b = 0; b < R; ++b) {
      float* tp = t[a][b];
      for (int j = 0; j < nb; ++j) tp[j] = 0.f;
      for (int q = 0; q < R; ++q) { const float c = wm.G[a * R + q]; const float* gp = g[q][b]; for (int j = 0; j < nb; ++j) tp[j] += c * gp[j]; }
    }
    for (int a = 0; a < A; ++a) for (int b = 0; b < A; ++b) {
      float acc[NB];
      for (int j = 0; j < nb; ++j) acc[j] = 0.f;
      for (int q = 0; q < R; ++q) { const float c = wm.G[b * R + q]; const float* tp = t[a][q]; for (int j = 0; j < nb; ++j) acc[j] += tp[j] * c; }
      float* up = U + (size_t)(a * A + b) * total + (size_t)k * Nn + n0;
      for (int j = 0; j < nb; ++j) up[j] = acc[j];
    }
  }
}
void wino_filter_transform(Stream&, int m, int r, const WShape& w, int mode, const float* packed, float* U) { SIM_TIMED;
  const int K = mode == 0 ? w.Cip : w.Npad, Nn = mode == 0 ? w.Npad : w.Cip;
  wino_filter_transform_strided(m, r, w, mode, packed, U, (size_t)K * Nn);
}
void wino_output_transform(Stream&, int m, int r, const float* M, int Cm, int Th, int Tw, const float* bias, int act,
                           const TView& y, int Cout, int accumulate, float* amax_out) { SIM_TIMED;
  const WinoMats wm = wino_mats(m, r);
  const int A = wm.A;
  const size_t T = (size_t)y.N * Th * Tw;
#pragma omp parallel for collapse(3)
  for (int n = 0; n < y.N; ++n) for (int ty = 0; ty < Th; ++ty) for (int tx = 0; tx < Tw; ++tx) {
    const size_t tile = ((size_t)n * Th + ty) * Tw + tx;
    for (int c = 0; c < Cout; ++c) {
      float mm[6][6], s2[4][6];
      for (int a = 0; a < A; ++a) for (int b = 0; b < A; ++b) mm[a][b] = M[((size_t)(a * A + b) * T + tile) * Cm + c];
      for (int a = 0; a < m; ++a) for (int b = 0; b < A; ++b) { float s = 0; for (int q = 0; q < A; ++q) s += wm.AT[a * A + q] * mm[q][b]; s2[a][b] = s; }
      for (int a = 0; a < m; ++a) for (int b = 0; b < m; ++b) {
        const int oy = m * ty + a, ox = m * tx + b;
        if (oy >= y.H || ox >= y.W) continue;
        float s = 0; for (int q = 0; q < A; ++q) s += s2[a][q] * wm.AT[b * A + q];
        if (bias) s += bias[c];
        s = actf(s, act);
        float* d = at(y, n, oy, ox) + c;
        *d = accumulate ? *d + s : s;
      }
    }
  }
  if (amax_out) { TView yc = y; yc.C = Cout; sim_fold_view(amax_out, yc); }
}
static void wino_dy_transform_impl(int m, int r, const TView& dy, int Th, int Tw, float* dM) {
  const WinoMats wm = wino_mats(m, r);
  const int A = wm.A;
  const size_t T = (size_t)dy.N * Th * Tw;
#pragma omp parallel for collapse(3)
  for (int n = 0; n < dy.N; ++n) for (int ty = 0; ty < Th; ++ty) for (int tx = 0; tx < Tw; ++tx) {
    const size_t tile = ((size_t)n * Th + ty) * Tw + tx;
    for (int c = 0; c < dy.C; ++c) {
      float g[4][4];
      for (int a = 0; a < m; ++a) for (int b = 0; b < m; ++b) {
        const int oy = m * ty + a, ox = m * tx + b;
        g[a][b] = (oy < dy.H && ox < dy.W) ? at(dy, n, oy, ox)[c] : 0.f;
      }
      for (int i = 0; i < A; ++i) for (int j = 0; j < A; ++j) {     // dM = A g A^T, A = AT^T
        float s = 0;
        for (int a = 0; a < m; ++a) for (int b = 0; b < m; ++b) s += wm.AT[a * A + i] * g[a][b] * wm.AT[b * A + j];
        dM[((size_t)(i * A + j) * T + tile) * dy.C + c] = s;
      }
    }
  }
}
// ---- folded tail conv in Winograd form (ops.h tailw_*): plain loops
static size_t tailw_off(int Cip, int Npad, int ph) { static const int pre[4] = {0, 4, 10, 16}; return (size_t)pre[ph] * Cip * Npad; }
void tailw_filter_transform(Stream&, const WShape& w, const float* folded, float* U) { SIM_TIMED;
  const WinoMats wm = wino_mats(4, 3);
  const int A = 6, R = 3, N4 = 4 * w.Npad;
  const size_t total = (size_t)w.Cip * N4;
  for (int ci = 0; ci < w.Cip; ++ci) for (int n = 0; n < N4; ++n) {
    const int ph = n / w.Npad, co = n % w.Npad, a = ph >> 1, b = ph & 1;
    const float* f = folded + tailw_off(w.Cip, w.Npad, ph);
    float g[3][3], t[6][3];
    for (int r = 0; r < R; ++r) for (int c = 0; c < R; ++c)
      g[r][c] = (r < 2 + a && c < 2 + b) ? f[((size_t)(r * (2 + b) + c) * w.Cip + ci) * w.Npad + co] : 0.f;
    for (int r = 0; r < A; ++r) for (int c = 0; c < R; ++c) { float s = 0; for (int q = 0; q < R; ++q) s += wm.G[r * R + q] * g[q][c]; t[r][c] = s; }
    for (int r = 0; r < A; ++r) for (int j = 0; j < A; ++j) { float s = 0; for (int q = 0; q < R; ++q) s += t[r][q] * wm.G[j * R + q];
      U[(size_t)(r * A + j) * total + (size_t)ci * N4 + n] = s; }
  }
}
void tailw_filter_grad(Stream&, const WShape& w, const float* dU, float* dfolded) { SIM_TIMED;
  const WinoMats wm = wino_mats(4, 3);
  const int A = 6, R = 3, N4 = 4 * w.Npad;
  const size_t total = (size_t)w.Cip * N4;
  for (int ci = 0; ci < w.Cip; ++ci) for (int n = 0; n < N4; ++n) {
    const int ph = n / w.Npad, co = n % w.Npad, a = ph >> 1, b = ph & 1;
    float* f = dfolded + tailw_off(w.Cip, w.Npad, ph);
    for (int r = 0; r < 2 + a; ++r) for (int c = 0; c < 2 + b; ++c) {
      float s = 0;
      for (int p = 0; p < A; ++p) for (int j = 0; j < A; ++j) s += wm.G[p * R + r] * dU[(size_t)(p * A + j) * total + (size_t)ci * N4 + n] * wm.G[j * R + c];
      f[((size_t)(r * (2 + b) + c) * w.Cip + ci) * w.Npad + co] = s;
    }
  }
}
void tailw_output_transform(Stream&, const float* M, int Th, int Tw, int Npad, const float* bias, int act, const TView& y, int Cout) {
  const WinoMats wm = wino_mats(4, 3);
  const int A = 6, CM = 4 * Npad;
  const size_t T = (size_t)y.N * Th * Tw;
  for (int n = 0; n < y.N; ++n) for (int ty = 0; ty < Th; ++ty) for (int tx = 0; tx < Tw; ++tx) {
    const size_t tile = ((size_t)n * Th + ty) * Tw + tx;
    for (int ph = 0; ph < 4; ++ph) for (int c = 0; c < Cout; ++c) {
      const int a = ph >> 1, b = ph & 1;
      for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        const int oy = 2 * (4 * ty + i) + a, ox = 2 * (4 * tx + j) + b;
        if (oy >= y.H || ox >= y.W) continue;
        float s = 0;
        for (int p = 0; p < A; ++p) for (int q = 0; q < A; ++q)
          s += wm.AT[i * A + p] * M[((size_t)(p * A + q) * T + tile) * CM + ph * Npad + c] * wm.AT[j * A + q];
        if (bias) s += bias[c];
        at(y, n, oy, ox)[c] = actf(s, act);
      }
    }
  }
}
static void tailw_dy_transform_impl(const TView& dy, int Th, int Tw, int Npad, float* dM) {
  const WinoMats wm = wino_mats(4, 3);
  const int A = 6, CM = 4 * Npad;
  const size_t T = (size_t)dy.N * Th * Tw;
  for (int n = 0; n < dy.N; ++n) for (int ty = 0; ty < Th; ++ty) for (int tx = 0; tx < Tw; ++tx) {
    const size_t tile = ((size_t)n * Th + ty) * Tw + tx;
    for (int ph = 0; ph < 4; ++ph) for (int c = 0; c < Npad; ++c) {
      const int a = ph >> 1, b = ph & 1;
      float g[4][4];
      for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        const int oy = 2 * (4 * ty + i) + a, ox = 2 * (4 * tx + j) + b;
        g[i][j] = (oy < dy.H && ox < dy.W && c < dy.C) ? at(dy, n, oy, ox)[c] : 0.f;
      }
      for (int p = 0; p < A; ++p) for (int q = 0; q < A; ++q) {
        float s = 0;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += wm.AT[i * A + p] * g[i][j] * wm.AT[j * A + q];
        dM[((size_t)(p * A + q) * T + tile) * CM + ph * Npad + c] = s;
      }
    }
  }
}
// ---- strided Winograd F(4x4, 2x2) (ops.h): plain loops straight from the definition
static void wino_s2_input_transform_impl(const TView& x, int Th, int Tw, float* V) {
  const WinoMats wm = wino_mats(4, 2);
  const int A = 5, C = x.C;
  const size_t T = (size_t)x.N * Th * Tw;
#pragma omp parallel for collapse(3)
  for (int n = 0; n < x.N; ++n) for (int ty = 0; ty < Th; ++ty) for (int tx = 0; tx < Tw; ++tx) {
    const size_t tile = ((size_t)n * Th + ty) * Tw + tx;
    for (int q = 0; q < 4; ++q) for (int c = 0; c < C; ++c) {
      const int s = q >> 1, t = q & 1;
      float d[5][5], u[5][5];
      for (int a = 0; a < A; ++a) for (int b = 0; b < A; ++b) {
        const int sy = 2 * (4 * ty + a) - 1 + s, sx = 2 * (4 * tx + b) - 1 + t;
        d[a][b] = (sy >= 0 && sy < x.H && sx >= 0 && sx < x.W) ? at(x, n, sy, sx)[c] : 0.f;
      }
      for (int a = 0; a < A; ++a) for (int b = 0; b < A; ++b) { float acc = 0; for (int k = 0; k < A; ++k) acc += wm.BT[a * A + k] * d[k][b]; u[a][b] = acc; }
      for (int a = 0; a < A; ++a) for (int b = 0; b < A; ++b) { float acc = 0; for (int k = 0; k < A; ++k) acc += u[a][k] * wm.BT[b * A + k];
        V[((size_t)(a * A + b) * T + tile) * 4 * C + q * C + c] = acc; }
    }
  }
}
void wino_s2_input_adjoint(Stream&, float* dV, int Cf, int Th, int Tw, const TView& dx, const float* bias, int accumulate) {
  const WinoMats wm = wino_mats(4, 2);
  const int A = 5;
  const size_t T = (size_t)dx.N * Th * Tw;
  if (!accumulate)
    for (size_t e = 0; e < dx.pixels(); ++e) std::memset(dx.p + e * dx.cs, 0, Cf * sizeof(float));
  for (int n = 0; n < dx.N; ++n) for (int ty = 0; ty < Th; ++ty) for (int tx = 0; tx < Tw; ++tx) {
    const size_t tile = ((size_t)n * Th + ty) * Tw + tx;
    for (int q = 0; q < 4; ++q) for (int c = 0; c < Cf; ++c) {
      const int s = q >> 1, t = q & 1;
      float v[5][5], u[5][5];
      for (int a = 0; a < A; ++a) for (int b = 0; b < A; ++b) v[a][b] = dV[((size_t)(a * A + b) * T + tile) * 4 * Cf + q * Cf + c];
      for (int a = 0; a < A; ++a) for (int b = 0; b < A; ++b) { float acc = 0; for (int k = 0; k < A; ++k) acc += wm.BT[k * A + a] * v[k][b]; u[a][b] = acc; }
      for (int a = 0; a < A; ++a) {
        const int sy = 2 * (4 * ty + a) - 1 + s;
        if (sy < 0 || sy >= dx.H) continue;
        for (int b = 0; b < A; ++b) {
          const int sx = 2 * (4 * tx + b) - 1 + t;
          if (sx < 0 || sx >= dx.W) continue;
          float acc = 0; for (int k = 0; k < A; ++k) acc += u[a][k] * wm.BT[k * A + b];
          at(dx, n, sy, sx)[c] += acc;
        }
      }
    }
  }
  if (bias)
    for (size_t e = 0; e < dx.pixels(); ++e) for (int c = 0; c < Cf; ++c) dx.p[e * dx.cs + c] += bias[c];
}
static size_t s2_widx(const WShape& w, int kh, int kw, int cf, int cc) {
  if (w.kind == WK_CONV) return ((size_t)(kh * 4 + kw) * w.Cip + cf) * w.Npad + cc;
  const int a = (3 - kh) & 1, dy = (3 - kh - a) >> 1, b = (3 - kw) & 1, dx = (3 - kw - b) >> 1;
  return (size_t)(a * 2 + b) * 4 * w.Cip * w.Npad + ((size_t)(dy * 2 + dx) * w.Cip + cc) * w.Npad + cf;
}
void wino_s2_filter_transform(Stream&, const WShape& w, int mode, const float* packed, float* U) { SIM_TIMED;
  const WinoMats wm = wino_mats(4, 2);
  const int A = 5, Cf = w.kind == WK_CONV ? w.Cip : w.Npad, Cc = w.kind == WK_CONV ? w.Npad : w.Cip;
  const size_t plane = (size_t)4 * Cf * Cc;
#pragma omp parallel for
  for (int cf = 0; cf < Cf; ++cf) for (int cc = 0; cc < Cc; ++cc) for (int q = 0; q < 4; ++q) {
    const int s = q >> 1, t = q & 1;
    float g[2][2], u[5][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) g[a][b] = packed[s2_widx(w, 2 * a + s, 2 * b + t, cf, cc)];
    for (int a = 0; a < A; ++a) for (int b = 0; b < 2; ++b) u[a][b] = wm.G[a * 2] * g[0][b] + wm.G[a * 2 + 1] * g[1][b];
    const size_t o = mode == 0 ? ((size_t)(q * Cf + cf) * Cc + cc) : ((size_t)cc * 4 * Cf + q * Cf + cf);
    for (int a = 0; a < A; ++a) for (int b = 0; b < A; ++b) U[(size_t)(a * A + b) * plane + o] = u[a][0] * wm.G[b * 2] + u[a][1] * wm.G[b * 2 + 1];
  }
}
void wino_s2_filter_grad(Stream&, const WShape& w, const float* dU, float* dpacked) { SIM_TIMED;
  const WinoMats wm = wino_mats(4, 2);
  const int A = 5, Cf = w.kind == WK_CONV ? w.Cip : w.Npad, Cc = w.kind == WK_CONV ? w.Npad : w.Cip;
  const size_t plane = (size_t)4 * Cf * Cc;
#pragma omp parallel for
  for (int cf = 0; cf < Cf; ++cf) for (int cc = 0; cc < Cc; ++cc) for (int q = 0; q < 4; ++q) {
    const int s = q >> 1, t = q & 1;
    const size_t o = (size_t)(q * Cf + cf) * Cc + cc;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) {
      float acc = 0;
      for (int i = 0; i < A; ++i) for (int j = 0; j < A; ++j) acc += wm.G[i * 2 + a] * dU[(size_t)(i * A + j) * plane + o] * wm.G[j * 2 + b];
      dpacked[s2_widx(w, 2 * a + s, 2 * b + t, cf, cc)] = acc;
    }
  }
}
// adjoint of wino_input_transform as a plain scatter over (tile, a, b): the second opinion on the HIP gather kernel
void wino_input_adjoint(Stream&, int m, int r, float* dV, int C, int pad, int pad_mode, int Th, int Tw, const TView& dx, int accumulate) { SIM_TIMED;
  const WinoMats wm = wino_mats(m, r);
  const int A = wm.A;
  const size_t T = (size_t)dx.N * Th * Tw;
  if (!accumulate)
    for (size_t e = 0; e < dx.pixels(); ++e) std::memset(dx.p + e * dx.cs, 0, C * sizeof(float));
  for (int n = 0; n < dx.N; ++n) for (int ty = 0; ty < Th; ++ty) for (int tx = 0; tx < Tw; ++tx) {
    const size_t tile = ((size_t)n * Th + ty) * Tw + tx;
    for (int c = 0; c < C; ++c) {
      float v[6][6], t[6][6];
      for (int a = 0; a < A; ++a) for (int b = 0; b < A; ++b) v[a][b] = dV[((size_t)(a * A + b) * T + tile) * C + c];
      // patch = BT^T v BT
      for (int a = 0; a < A; ++a) for (int b = 0; b < A; ++b) { float s = 0; for (int k = 0; k < A; ++k) s += wm.BT[k * A + a] * v[k][b]; t[a][b] = s; }
      for (int a = 0; a < A; ++a) {
        const int sy = srcc(m * ty - pad + a, dx.H, pad_mode, 0);
        if (sy < 0) continue;
        for (int b = 0; b < A; ++b) {
          const int sx = srcc(m * tx - pad + b, dx.W, pad_mode, 0);
          if (sx < 0) continue;
          float s = 0; for (int k = 0; k < A; ++k) s += t[a][k] * wm.BT[k * A + b];
          at(dx, n, sy, sx)[c] += s;
        }
      }
    }
  }
}
void wino_filter_grad(Stream&, int m, int r, const WShape& w, const float* dU, float* dpacked) { SIM_TIMED;
  // dg = G^T dU G per (ci, co) pair; 64 pairs at a time (the 36 plane values of a pair are 36 separate streams), the sum of a
  // tap in the original order: p-major, q-minor, (G[p][a] * dU) * G[q][b]
  const WinoMats wm = wino_mats(m, r);
  const int A = wm.A, R = wm.r;
  const size_t total = (size_t)w.Cip * w.Npad;
  const long NB = 64;
#pragma omp parallel for
  for (long i0 = 0; i0 < (long)total; i0 += NB) {
    const int nb = (int)std::min<long>(NB, (long)total - i0);
    float d[36][NB];
    for (int pq = 0; pq < A * A; ++pq) { const float* sp = dU + (size_t)pq * total + i0; for (int j = 0; j < nb; ++j) d[pq][j] = sp[j]; }
    for (int a = 0; a < R; ++a) for (int b = 0; b < R; ++b) {
      float acc[NB];
      for (int j = 0; j < nb; ++j) acc[j] = 0.f;
      for (int p = 0; p < A; ++p) for (int q = 0; q < A; ++q) {
        const float ga = wm.G[p * R + a], gb = wm.G[q * R + b]; const float* dp = d[p * A + q];
        for (int j = 0; j < nb; ++j) acc[j] += ga * dp[j] * gb;
      }
      float* op = dpacked + (size_t)(a * R + b) * total + i0;
      for (int j = 0; j < nb; ++j) op[j] = acc[j];
    }
  }
}

void bias_grad(Stream&, const TView& dy, float* db) {
  std::vector<double> acc(dy.C, 0.0);
  for (size_t e = 0; e < dy.pixels(); ++e)
    for (int c = 0; c < dy.C; ++c) acc[c] += dy.p[e * dy.cs + c];
  for (int c = 0; c < dy.C; ++c) db[c] = (float)acc[c];
}

void reflect_fold(Stream&, const TView& sp, const TView& d, int accumulate) {
  const int H = d.H, W = d.W;
  if (!accumulate)
    for (size_t e = 0; e < d.pixels(); ++e) std::memset(d.p + e * d.cs, 0, d.C * sizeof(float));
  for (int n = 0; n < d.N; ++n)
    for (int u = 0; u < H + 2; ++u)
      for (int v = 0; v < W + 2; ++v) {
        int y = u - 1, x = v - 1;
        if (y < 0) y = -y; else if (y >= H) y = 2 * H - 2 - y;
        if (x < 0) x = -x; else if (x >= W) x = 2 * W - 2 - x;
        const float* s = at(sp, n, u, v);
        float* o = at(d, n, y, x);
        for (int c = 0; c < d.C; ++c) o[c] += s[c];
      }
}

static inline float hdrop(uint64_t seed, uint64_t idx, float p) {
  uint64_t z = seed * 0xD1342543DE82EF95ull + idx + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
  const float u = (float)((uint32_t)(z >> 11) & 0xFFFFFFu) * (1.0f / 16777216.0f);
  return u >= p ? 1.0f / (1.0f - p) : 0.0f;
}

void dropout_mask(Stream&, int N, int H, int W, int C, float p, uint64_t seed, float* out) {
  const size_t HW = (size_t)H * W;
  for (size_t n = 0; n < (size_t)N; ++n)
    for (int c = 0; c < C; ++c)
      for (size_t pix = 0; pix < HW; ++pix) out[(n * C + c) * HW + pix] = hdrop(seed, (n * HW + pix) * C + c, p);
}

void act_pattern(Stream&, const TView& y, uint8_t* out) {
  const size_t HW = (size_t)y.H * y.W;
  for (size_t n = 0; n < (size_t)y.N; ++n)
    for (int c = 0; c < y.C; ++c)
      for (size_t pix = 0; pix < HW; ++pix) out[(n * y.C + c) * HW + pix] = y.p[(n * HW + pix) * y.cs + c] > 0.f ? 1 : 0;
}
void pool_pattern(Stream&, const TView& x, const TView& y, uint8_t* out) {
  if (x.H != y.H * 2 || x.W != y.W * 2 || x.C != y.C) throw Error(1, "pool_pattern: shape mismatch");
  for (int n = 0; n < y.N; ++n)
    for (int c = 0; c < y.C; ++c)
      for (int oy = 0; oy < y.H; ++oy)
        for (int ox = 0; ox < y.W; ++ox) {
          float m = 0.f; int am = 0;
          for (int t = 0; t < 4; ++t) {
            const float v = x.p[(((size_t)n * x.H + oy * 2 + (t >> 1)) * x.W + ox * 2 + (t & 1)) * x.cs + c];
            if (t == 0 || v > m || v != v) { m = v; am = t; }
          }
          out[(((size_t)n * y.C + c) * y.H + oy) * y.W + ox] = (uint8_t)am;
        }
}

void norm_act_fwd(Stream&, const NormActArgs& a0) { SIM_TIMED;
  NormActArgs a = a0;
  if (a.seed_base) a.seed = *a.seed_base * 0x9E3779B1ull + a.salt;       // captured-step form of the seed (ops.h)
  const int N = a.x.N, HW = a.x.H * a.x.W, C = a.x.C;
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      float mean = 0.f, rstd = 1.f;
      if (a.norm) {
        double s = 0, ss = 0;
        for (int p = 0; p < HW; ++p) { const double v = a.x.p[((size_t)n * HW + p) * a.x.cs + c]; s += v; ss += v * v; }
        const double m = s / HW; double var = ss / HW - m * m; if (var < 0) var = 0;
        mean = (float)m; rstd = (float)(1.0 / std::sqrt(var + 1e-5));
        a.stats[((size_t)n * C + c) * 2] = mean; a.stats[((size_t)n * C + c) * 2 + 1] = rstd;
      }
      for (int p = 0; p < HW; ++p) {
        const size_t e = (size_t)n * HW + p;
        float v = (a.x.p[e * a.x.cs + c] - mean) * rstd;
        v = actf(v, a.act);
        if (a.drop_p > 0.f) v *= hdrop(a.seed, e * C + c, a.drop_p);
        if (a.residual) v += a.residual->p[e * a.residual->cs + c];
        a.y.p[e * a.y.cs + c] = v;
      }
    }
  sim_fold_view(a.amax_out, a.y);
}
void norm_act_bwd(Stream&, const NormActBwdArgs& a0) { SIM_TIMED;
  NormActBwdArgs a = a0;
  if (a.seed_base) a.seed = *a.seed_base * 0x9E3779B1ull + a.salt;
  const int N = a.x.N, HW = a.x.H * a.x.W, C = a.x.C;
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      float mean = 0.f, rstd = 1.f;
      if (a.norm) { mean = a.stats[((size_t)n * C + c) * 2]; rstd = a.stats[((size_t)n * C + c) * 2 + 1]; }
      double s1 = 0, s2 = 0;
      std::vector<float> gh(HW), xh(HW);
      for (int p = 0; p < HW; ++p) {
        const size_t e = (size_t)n * HW + p;
        xh[p] = (a.x.p[e * a.x.cs + c] - mean) * rstd;
        float g = a.dy.p[e * a.dy.cs + c] * actg_in(xh[p], a.act);
        if (a.drop_p > 0.f) g *= hdrop(a.seed, e * C + c, a.drop_p);
        gh[p] = g; s1 += g; s2 += (double)g * xh[p];
      }
      const float m1 = (float)(s1 / HW), m2 = (float)(s2 / HW);
      for (int p = 0; p < HW; ++p) {
        const size_t e = (size_t)n * HW + p;
        a.dx.p[e * a.dx.cs + c] = a.norm ? rstd * (gh[p] - m1 - xh[p] * m2) : gh[p];
      }
    }
  sim_fold_view(a.amax_out, a.dx);
}

bool norm_act_bwd_emits_colsum(int, int) { return false; }
void bias_grad_from_colsums(Stream&, const double* partial, int N, int C, float* db) {
  for (int c = 0; c < C; ++c) { double a = 0; for (int n = 0; n < N; ++n) a += partial[(size_t)n * C + c]; db[c] = (float)a; }
}
void act_fwd(Stream&, const TView& x, const TView& y, int act, float* amax_out) {
  for (size_t e = 0; e < x.pixels(); ++e)
    for (int c = 0; c < x.C; ++c) y.p[e * y.cs + c] = actf(x.p[e * x.cs + c], act);
  sim_fold_view(amax_out, y);
}
void act_bwd(Stream&, const TView& dy, const TView& y, const TView& dx, int act, int accumulate, float* amax_out) {
  for (size_t e = 0; e < dy.pixels(); ++e)
    for (int c = 0; c < dy.C; ++c) {
      float v = dy.p[e * dy.cs + c] * actg_out(y.p[e * y.cs + c], act);
      if (accumulate) v += dx.p[e * dx.cs + c];
      dx.p[e * dx.cs + c] = v;
    }
  sim_fold_view(amax_out, dx);
}
void axpy(Stream&, const TView& src, const TView& dst, float alpha, int accumulate, float shift, float* amax_out) {
  for (size_t e = 0; e < src.pixels(); ++e)
    for (int c = 0; c < src.C; ++c) {
      float v = src.p[e * src.cs + c] * alpha + shift;
      if (accumulate) v += dst.p[e * dst.cs + c];
      dst.p[e * dst.cs + c] = v;
    }
  sim_fold_view(amax_out, dst);
}

void upsample_nearest_fwd(Stream&, const TView& x, const TView& y, int f, float* amax_out) {
  for (int n = 0; n < y.N; ++n) for (int oy = 0; oy < y.H; ++oy) for (int ox = 0; ox < y.W; ++ox)
    std::memcpy(at(y, n, oy, ox), at(x, n, oy / f, ox / f), x.C * sizeof(float));
  sim_fold_view(amax_out, y);
}
void upsample_nearest_bwd(Stream&, const TView& dy, const TView& dx, int f, int accumulate) {
  for (int n = 0; n < dx.N; ++n) for (int iy = 0; iy < dx.H; ++iy) for (int ix = 0; ix < dx.W; ++ix) {
    float* o = at(dx, n, iy, ix);
    for (int c = 0; c < dx.C; ++c) {
      float s = 0;
      for (int a = 0; a < f; ++a) for (int b = 0; b < f; ++b) s += at(dy, n, iy * f + a, ix * f + b)[c];
      o[c] = accumulate ? o[c] + s : s;
    }
  }
}
void maxpool2_fwd(Stream&, const TView& x, const TView& y, float* amax_out) {
  for (int n = 0; n < y.N; ++n) for (int oy = 0; oy < y.H; ++oy) for (int ox = 0; ox < y.W; ++ox)
    for (int c = 0; c < y.C; ++c) {
      float m = at(x, n, oy * 2, ox * 2)[c];
      for (int t = 1; t < 4; ++t) { const float v = at(x, n, oy * 2 + (t >> 1), ox * 2 + (t & 1))[c]; if (v > m) m = v; }
      at(y, n, oy, ox)[c] = m;
    }
  sim_fold_view(amax_out, y);
}
void maxpool2_bwd(Stream&, const TView& dy, const TView& x, const TView& y, const TView& dx, int accumulate) {
  for (int n = 0; n < y.N; ++n) for (int oy = 0; oy < y.H; ++oy) for (int ox = 0; ox < y.W; ++ox)
    for (int c = 0; c < y.C; ++c) {
      float m = at(x, n, oy * 2, ox * 2)[c]; int am = 0;
      for (int t = 1; t < 4; ++t) { const float v = at(x, n, oy * 2 + (t >> 1), ox * 2 + (t & 1))[c]; if (v > m) { m = v; am = t; } }
      for (int t = 0; t < 4; ++t) { float* d = at(dx, n, oy * 2 + (t >> 1), ox * 2 + (t & 1)) + c; const float v = t == am ? at(dy, n, oy, ox)[c] : 0.f; *d = accumulate ? *d + v : v; }
    }
}

struct RoiS { int yl, yh, xl, xh; float w1, w2, w3, w4; bool valid; };
static RoiS roi_s(const float* roi, int ph, int pw, int PH, int PW, int H, int W) {
  RoiS r;
  volatile float sw = roi[0], sh = roi[1], ew = roi[2], eh = roi[3];
  volatile float rw = std::fmax((float)(ew - sw), 1.0f), rh = std::fmax((float)(eh - sh), 1.0f);
  volatile float bw = rw / (float)PW, bh = rh / (float)PH;
  volatile float t1 = (float)ph * bh, t2 = 0.5f * bh; volatile float t3 = t2 / 1.0f; volatile float t4 = sh + t1;
  float y = t4 + t3;
  volatile float u1 = (float)pw * bw, u2 = 0.5f * bw; volatile float u3 = u2 / 1.0f; volatile float u4 = sw + u1;
  float x = u4 + u3;
  r.valid = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
  y = std::fmax(y, 0.f); x = std::fmax(x, 0.f);
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
  volatile float ly = y - (float)yl, lx = x - (float)xl; volatile float hy = 1.f - ly, hx = 1.f - lx;
  r.yl = yl; r.yh = yh; r.xl = xl; r.xh = xh;
  volatile float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  r.w1 = w1; r.w2 = w2; r.w3 = w3; r.w4 = w4;
  return r;
}
void roi_align_fwd(Stream&, const TView& tex, int C, const float* rois, int R, const TView& out) {
  for (int b = 0; b < tex.N; ++b) for (int r = 0; r < R; ++r) for (int ph = 0; ph < out.H; ++ph) for (int pw = 0; pw < out.W; ++pw) {
    const RoiS s = roi_s(rois + ((size_t)b * R + r) * 4, ph, pw, out.H, out.W, tex.H, tex.W);
    float* o = at(out, b, ph, pw) + r * C;
    for (int c = 0; c < C; ++c) {
      float v = 0.f;
      if (s.valid) {
        volatile float a1 = s.w1 * at(tex, b, s.yl, s.xl)[c], a2 = s.w2 * at(tex, b, s.yl, s.xh)[c];
        volatile float a3 = s.w3 * at(tex, b, s.yh, s.xl)[c], a4 = s.w4 * at(tex, b, s.yh, s.xh)[c];
        volatile float s12 = a1 + a2; volatile float s123 = s12 + a3;
        v = s123 + a4;
      }
      o[c] = v;
    }
  }
}
void roi_align_indices(Stream&, const float* rois, int K, int H, int W, int PH, int PW, int32_t* idx, uint8_t* valid) {
  for (int k = 0; k < K; ++k) for (int ph = 0; ph < PH; ++ph) for (int pw = 0; pw < PW; ++pw) {
    const RoiS s = roi_s(rois + (size_t)k * 4, ph, pw, PH, PW, H, W);
    const size_t i = ((size_t)k * PH + ph) * PW + pw;
    idx[i * 4] = s.yl; idx[i * 4 + 1] = s.yh; idx[i * 4 + 2] = s.xl; idx[i * 4 + 3] = s.xh; valid[i] = s.valid;
  }
}

void affine_gather(Stream&, const float* src, float* dst, int B, int C, int H, int W, const double* maps, int nmaps) {
  for (int bc = 0; bc < B * C; ++bc)
    for (int y0 = 0; y0 < H; ++y0)
      for (int x0 = 0; x0 < W; ++x0) {
        long long x = x0, y = y0;
        bool inside = true;
        for (int k = nmaps - 1; k >= 0 && inside; --k) {
          const double* m = maps + ((size_t)bc * nmaps + k) * 9;
          const int kind = (int)m[0];
          long long xin = x, yin = y;
          if (kind == 1) {
            xin = ((long long)m[3] + (long long)m[2] * y + (long long)m[1] * x) >> 16;
            yin = ((long long)m[6] + (long long)m[5] * y + (long long)m[4] * x) >> 16;
          } else if (kind == 2) {
            const double xc = (double)x + 0.5, yc = (double)y + 0.5;
            const double den = m[7] * xc + m[8] * yc + 1.0;
            const double fx = (m[1] * xc + m[2] * yc + m[3]) / den, fy = (m[4] * xc + m[5] * yc + m[6]) / den;
            xin = fx < 0.0 ? -1 : (long long)(int)fx;
            yin = fy < 0.0 ? -1 : (long long)(int)fy;
          }
          inside = xin >= 0 && xin < W && yin >= 0 && yin < H;
          x = xin; y = yin;
        }
        dst[((size_t)bc * H + y0) * W + x0] = inside ? src[((size_t)bc * H + y) * W + x] : 0.f;
      }
}

void nchw_to_nhwc(Stream&, const float* src, int N, int C, int H, int W, const TView& dst) {
  const int HW = H * W;
  for (int n = 0; n < N; ++n) for (int c = 0; c < C; ++c) for (int p = 0; p < HW; ++p)
    dst.p[((size_t)n * HW + p) * dst.cs + c] = src[((size_t)n * C + c) * HW + p];
}
void nhwc_to_nchw(Stream&, const TView& src, float* dst, int C) {
  const int HW = src.H * src.W;
  for (int n = 0; n < src.N; ++n) for (int c = 0; c < C; ++c) for (int p = 0; p < HW; ++p)
    dst[((size_t)n * C + c) * HW + p] = src.p[((size_t)n * HW + p) * src.cs + c];
}
static const uint8_t kPal[19][3] = {{0,0,0},{128,0,0},{255,0,0},{0,85,0},{255,85,0},{0,0,85},{0,119,221},{85,85,0},{0,85,85},
  {85,51,0},{52,86,128},{0,128,0},{0,0,255},{51,170,221},{0,255,255},{85,255,170},{170,255,85},{255,255,0},{255,170,0}};
static int amax(const float* p, int C) { int b = 0; float m = p[0]; for (int c = 1; c < C; ++c) if (p[c] > m) { m = p[c]; b = c; } return b; }
void decode_labels(Stream&, const TView& x, int C, uint8_t* rgb) {
  const int HW = x.H * x.W;
  for (int n = 0; n < x.N; ++n) for (int p = 0; p < HW; ++p) {
    const int b = amax(x.p + ((size_t)n * HW + p) * x.cs, C);
    for (int ch = 0; ch < 3; ++ch) rgb[((size_t)n * 3 + ch) * HW + p] = b < 19 ? kPal[b][ch] : 0;
  }
}
void argmax_labels(Stream&, const TView& x, int C, int32_t* labels) {
  for (size_t e = 0; e < x.pixels(); ++e) labels[e] = amax(x.p + e * x.cs, C);
}
void labels_to_onehot(Stream&, const int32_t* labels, const TView& y, int C) {
  for (size_t e = 0; e < y.pixels(); ++e)
    for (int c = 0; c < C; ++c) y.p[e * y.cs + c] = (c == labels[e] && labels[e] != 0) ? 1.f : 0.f;
}

// ---- 1-channel head conv as taps-on-N (ops.h) ---------------------------------------------
void head_pack(Stream&, const WShape& w, const float* packed, float* wt, float* wt2) {
  for (int t = 0; t < 16; ++t) for (int c = 0; c < w.Cip; ++c) {
    const float v = packed[(size_t)(t * w.Cip + c) * w.Npad];
    wt[c * 16 + t] = v; wt2[(size_t)t * w.Cip + c] = v;
  }
}
void head_unpack_grad(Stream&, const WShape& w, const float* dwt, float* dpacked) {
  for (int t = 0; t < 16; ++t) for (int c = 0; c < w.Cip; ++c) {
    float* d = dpacked + (size_t)(t * w.Cip + c) * w.Npad;
    d[0] = dwt[c * 16 + t];
    for (int j = 1; j < w.Npad; ++j) d[j] = 0.f;
  }
}
void head_gather(Stream&, const TView& z, const float* bias, const TView& y) {
  for (int n = 0; n < y.N; ++n) for (int oy = 0; oy < y.H; ++oy) for (int ox = 0; ox < y.W; ++ox) {
    float acc = 0.f;
    for (int kh = 0; kh < 4; ++kh) for (int kw = 0; kw < 4; ++kw) {
      const int sy = oy - 1 + kh, sx = ox - 1 + kw;
      if (sy < 0 || sy >= z.H || sx < 0 || sx >= z.W) continue;
      acc += at(z, n, sy, sx)[kh * 4 + kw];
    }
    float* d = at(y, n, oy, ox);
    d[0] = acc + (bias ? bias[0] : 0.f); d[1] = d[2] = d[3] = 0.f;
  }
}
void head_scatter(Stream&, const TView& dy, const TView& dz) {
  for (int n = 0; n < dz.N; ++n) for (int iy = 0; iy < dz.H; ++iy) for (int ix = 0; ix < dz.W; ++ix)
    for (int kh = 0; kh < 4; ++kh) for (int kw = 0; kw < 4; ++kw) {
      const int oy = iy + 1 - kh, ox = ix + 1 - kw;
      at(dz, n, iy, ix)[kh * 4 + kw] = (oy >= 0 && oy < dy.H && ox >= 0 && ox < dy.W) ? at(dy, n, oy, ox)[0] : 0.f;
    }
}

// ---- losses ----------------------------------------------------------------------------
static void gan(const TView& pred, float label, float scale, float* out, const TView* dp, int mode) {
  const size_t n = pred.pixels(); double acc = 0;
  for (size_t i = 0; i < n; ++i) {
    const float x = pred.p[i * pred.cs]; float l, g;
    if (mode == 0) { l = std::fmax(x, 0.f) - x * label + std::log1p(std::exp(-std::fabs(x))); g = 1.f / (1.f + std::exp(-x)) - label; }
    else if (mode == 1) { const float d = x - label; l = d * d; g = 2 * d; }
    else { l = label * x; g = label; }
    acc += l;
    if (dp) dp->p[i * dp->cs] = g * scale / (float)n;
  }
  *out = (float)(acc / n);
}
void bce_logits_loss(Stream&, const TView& p, float l, float s, float* o, const TView* d, const float* ld) { gan(p, ld ? *ld : l, s, o, d, 0); }
void lsgan_loss(Stream&, const TView& p, float l, float s, float* o, const TView* d, const float* ld) { gan(p, ld ? *ld : l, s, o, d, 1); }
void wgan_loss(Stream&, const TView& p, float l, float s, float* o, const TView* d) { gan(p, l, s, o, d, 2); }

void ce_argmax_loss(Stream&, const TView& lg, const TView& tg, int C, float scale, float* out, const TView* dl, int accumulate) {
  const size_t P = lg.pixels(); double acc = 0;
  for (size_t e = 0; e < P; ++e) {
    const float* lp = lg.p + e * lg.cs; const float* tp = tg.p + e * tg.cs;
    const int label = amax(tp, C);
    float lmax = lp[0]; for (int c = 1; c < C; ++c) lmax = std::fmax(lmax, lp[c]);
    float se = 0; for (int c = 0; c < C; ++c) se += std::exp(lp[c] - lmax);
    acc += (lmax + std::log(se)) - lp[label];
    if (dl) { float* dp = dl->p + e * dl->cs;
      for (int c = 0; c < C; ++c) { float g = (std::exp(lp[c] - lmax) / se - (c == label)) * scale / (float)P; dp[c] = accumulate ? dp[c] + g : g; } }
  }
  *out = (float)(acc / P);
}
void l1_loss(Stream&, const TView& a, const TView& b, int C, float scale, float* out, const TView* da, int accumulate) {
  const size_t P = a.pixels(); double acc = 0; const float n = (float)(P * C);
  for (size_t e = 0; e < P; ++e) for (int c = 0; c < C; ++c) {
    const float d = a.p[e * a.cs + c] - b.p[e * b.cs + c]; acc += std::fabs(d);
    if (da) { float g = (d > 0 ? 1.f : d < 0 ? -1.f : 0.f) * scale / n; float* dp = da->p + e * da->cs + c; *dp = accumulate ? *dp + g : g; }
  }
  *out = (float)(acc / n);
}
void normed_mse_loss(Stream&, const TView& f, const TView& t, float scale, float* out, const TView* df, int accumulate) {
  const size_t P = f.pixels(); const int C = f.C; double acc = 0; const float numel = (float)(P * C);
  std::vector<float> g(C);
  for (size_t e = 0; e < P; ++e) {
    const float* fp = f.p + e * f.cs; const float* tp = t.p + e * t.cs;
    float sf = 0, st = 0; for (int c = 0; c < C; ++c) { sf += fp[c] * fp[c]; st += tp[c] * tp[c]; }
    sf = std::sqrt(sf); st = std::sqrt(st);
    const float inf = 1.f / (sf + 1e-8f), intt = 1.f / (st + 1e-8f);
    float dot = 0;
    for (int c = 0; c < C; ++c) { g[c] = fp[c] * inf - tp[c] * intt; acc += (double)g[c] * g[c]; dot += g[c] * fp[c]; }
    if (df) { const float k2 = sf > 0 ? dot * inf * inf / sf : 0.f; float* dp = df->p + e * df->cs;
      for (int c = 0; c < C; ++c) { float o = (g[c] * inf - fp[c] * k2) * 2.f * scale / numel; dp[c] = accumulate ? dp[c] + o : o; } }
  }
  *out = (float)(acc / numel);
}
void gram_style_loss(Stream&, const TView& a, const TView& b, int C, float scale, float* out, const TView* da, int accumulate,
                     int n0, int nloc) {
  const int R = a.N * C, HW = a.H * a.W;
  if (nloc < 0) { n0 = 0; nloc = a.N; }
  std::vector<double> Ga((size_t)R * R, 0.0), Gb((size_t)R * R, 0.0);
  auto val = [&](const TView& v, int r, int p) { return v.p[((size_t)(r / C) * HW + p) * v.cs + (r % C)]; };
  for (int r1 = 0; r1 < R; ++r1) for (int r2 = 0; r2 < R; ++r2) { double s1 = 0, s2 = 0;
    for (int p = 0; p < HW; ++p) { s1 += (double)val(a, r1, p) * val(a, r2, p); s2 += (double)val(b, r1, p) * val(b, r2, p); }
    Ga[(size_t)r1 * R + r2] = s1; Gb[(size_t)r1 * R + r2] = s2; }
  double acc = 0; std::vector<float> dG((size_t)R * R);
  for (size_t i = 0; i < Ga.size(); ++i) { const double d = Ga[i] - Gb[i]; acc += d * d; dG[i] = (float)(2.0 * d) * scale / (float)(R * R); }
  *out = (float)(acc / ((double)R * R));
  if (da) for (int p = 0; p < HW; ++p) for (int rl = 0; rl < nloc * C; ++rl) { float s = 0;
    const int r = n0 * C + rl;
    for (int r2 = 0; r2 < R; ++r2) s += (dG[(size_t)r * R + r2] + dG[(size_t)r2 * R + r]) * val(a, r2, p);
    float* dp = da->p + ((size_t)(rl / C) * HW + p) * da->cs + (rl % C); *dp = accumulate ? *dp + s : s; }
}
void gp_interpolate(Stream&, const TView& a, const TView* b, const float* alpha, const TView* beta, const float* half_std,
                    const TView& out) {
  const size_t HW = (size_t)a.H * a.W;
  for (size_t e = 0; e < a.pixels(); ++e)
    for (int c = 0; c < a.C; ++c) {
      const float av = a.p[e * a.cs + c];
      const float bv = b ? b->p[e * b->cs + c] : av + half_std[0] * beta->p[e * beta->cs + c];
      out.p[e * out.cs + c] = av + alpha[e / HW] * (bv - av);
    }
}
void gp_half_std(Stream&, const TView& a, size_t numel, float* out) {
  double s1 = 0, s2 = 0;
  for (size_t e = 0; e < a.pixels(); ++e)
    for (int c = 0; c < a.C; ++c) { const double v = a.p[e * a.cs + c]; s1 += v; s2 += v * v; }
  const double mean = s1 / (double)numel;
  double var = (s2 - (double)numel * mean * mean) / ((double)numel - 1.0);
  if (var < 0) var = 0;
  out[0] = (float)(0.5 * std::sqrt(var));
}
void gp_penalty(Stream&, const TView& g, int lp, float scale, float* loss_out, const TView& u) {
  const size_t HW = (size_t)g.H * g.W;
  double loss = 0;
  for (int n = 0; n < g.N; ++n) {
    double s2 = 0;
    for (size_t e = n * HW; e < (n + 1) * HW; ++e)
      for (int c = 0; c < g.C; ++c) { const double v = g.p[e * g.cs + c]; s2 += v * v; }
    const double norm = std::sqrt(s2);
    double d = norm - 1.0;
    if (lp && d < 0) d = 0;
    loss += d * d;
    const float coef = norm > 0 ? (float)((double)scale * 2.0 * d / ((double)g.N * norm)) : 0.f;
    for (size_t e = n * HW; e < (n + 1) * HW; ++e)
      for (int c = 0; c < g.C; ++c) u.p[e * u.cs + c] = coef * g.p[e * g.cs + c];
  }
  loss_out[0] = (float)(loss / g.N);
}
void gp_uniform(Stream&, const TView& v, int Clog, uint64_t seed, const int32_t* cimap) {
  for (size_t e = 0; e < v.pixels(); ++e)
    for (int c = 0; c < v.C; ++c) {
      uint64_t z = seed * 0xD1342543DE82EF95ull + (e * v.C + c) + 0x9E3779B97F4A7C15ull;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
      const bool live = c < Clog && (!cimap || cimap[c] >= 0);
      v.p[e * v.cs + c] = live ? (float)((uint32_t)(z >> 11) & 0xFFFFFFu) * (1.0f / 16777216.0f) : 0.f;
    }
}
void norm_act_bwd2(Stream&, const NormActBwd2Args& a) {
  const int N = a.x.N, HW = a.x.H * a.x.W, C = a.x.C;
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      const float mean = a.stats[((size_t)n * C + c) * 2], rstdf = a.stats[((size_t)n * C + c) * 2 + 1];
      const double rstd = rstdf;
      double m[5] = {0, 0, 0, 0, 0};
      for (int p = 0; p < HW; ++p) {
        const size_t e = (size_t)n * HW + p;
        const float xv = a.x.p[e * a.x.cs + c];
        const double xh = ((double)xv - mean) * rstd;
        const double uu = a.u.p[e * a.u.cs + c];
        const double gm = (double)(a.gy.p[e * a.gy.cs + c] * actg_in((xv - mean) * rstdf, a.act));
        m[0] += uu; m[1] += gm; m[2] += uu * xh; m[3] += gm * xh; m[4] += uu * gm;
      }
      for (double& v : m) v /= HW;
      for (int p = 0; p < HW; ++p) {
        const size_t e = (size_t)n * HW + p;
        const float xv = a.x.p[e * a.x.cs + c];
        const float ad = actg_in((xv - mean) * rstdf, a.act);
        const double xh = ((double)xv - mean) * rstd;
        const double uu = a.u.p[e * a.u.cs + c], gm = (double)(a.gy.p[e * a.gy.cs + c] * ad);
        const double ju = uu - m[0] - xh * m[2];
        a.uy.p[e * a.uy.cs + c] = (float)((double)ad * rstd * ju);
        a.ax.p[e * a.ax.cs + c] = (float)(-rstd * rstd * (xh * (m[4] - m[0] * m[1] - m[2] * m[3]) + m[3] * ju + m[2] * (gm - m[1] - xh * m[3])));
      }
    }
}
void scalar_axpby(Stream&, const float* a, float ca, const float* b, float cb, float* out) {
  out[0] = (a ? a[0] * ca : 0.f) + (b ? b[0] * cb : 0.f);
}

// ---- optimizer / layouts -----------------------------------------------------------------
void adamw_schedule(float lr, float beta1, float beta2, int step, float out[2]) {
  const double bc1 = 1.0 - std::pow((double)beta1, step), bc2 = 1.0 - std::pow((double)beta2, step);
  out[0] = (float)(lr / bc1); out[1] = (float)(1.0 / std::sqrt(bc2));
}
void adamw_step(Stream&, const AdamWArgs& a) { SIM_TIMED;  // (elementwise: threads split the arena)
  float sched[2];
  adamw_schedule(a.lr, a.beta1, a.beta2, a.step, sched);
  if (a.sched_dev) { sched[0] = a.sched_dev[0]; sched[1] = a.sched_dev[1]; }
  const float decay = 1.f - a.lr * a.weight_decay, ss = sched[0], isb = sched[1];
#pragma omp parallel for
  for (long i = 0; i < (long)a.n; ++i) {
    float p = a.p[i] * decay;
    const float m = a.m[i] * a.beta1 + (1.f - a.beta1) * a.g[i];
    const float v = a.v[i] * a.beta2 + (1.f - a.beta2) * a.g[i] * a.g[i];
    p -= ss * (m / (std::sqrt(v) * isb + a.eps));
    a.p[i] = p; a.m[i] = m; a.v[i] = v;
  }
}
size_t packed_elems(const WShape& w) {
  return w.kind == WK_CONV ? (size_t)w.KH * w.KW * w.Cip * w.Npad : (size_t)16 * w.Cip * w.Npad;
}
static int refch(const WShape& w, int cb) { return w.cimap ? w.cimap[cb] : (cb < w.Ci ? cb : -1); }
void pack_weight(Stream&, const WShape& w, const float* src, float* dst) { SIM_TIMED;
  std::memset(dst, 0, packed_elems(w) * sizeof(float));
#pragma omp parallel for
  for (int cb = 0; cb < w.Cip; ++cb) {
    const int ci = refch(w, cb);
    if (ci < 0) continue;
    for (int co = 0; co < w.Co; ++co) for (int ky = 0; ky < w.KH; ++ky) for (int kx = 0; kx < w.KW; ++kx) {
      if (w.kind == WK_CONV) {
        dst[((size_t)(ky * w.KW + kx) * w.Cip + cb) * w.Npad + co] = src[(((size_t)co * w.Ci + ci) * w.KH + ky) * w.KW + kx];
      } else {   // out[2i+a] += in[i + a - 1 + dy] * W[3 - a - 2dy]
        const int a = (3 - ky) & 1, dy = (3 - ky) >> 1, b = (3 - kx) & 1, dx = (3 - kx) >> 1;
        dst[((size_t)(a * 2 + b) * 4 * w.Cip + (dy * 2 + dx) * w.Cip + cb) * w.Npad + co] =
            src[(((size_t)ci * w.Co + co) * 4 + ky) * 4 + kx];
      }
    }
  }
}
void unpack_weight(Stream&, const WShape& w, const float* src, float* dst) { SIM_TIMED;
#pragma omp parallel for
  for (int cb = 0; cb < w.Cip; ++cb) {
    const int ci = refch(w, cb);
    if (ci < 0) continue;
    for (int co = 0; co < w.Co; ++co) for (int ky = 0; ky < w.KH; ++ky) for (int kx = 0; kx < w.KW; ++kx) {
      if (w.kind == WK_CONV) {
        dst[(((size_t)co * w.Ci + ci) * w.KH + ky) * w.KW + kx] = src[((size_t)(ky * w.KW + kx) * w.Cip + cb) * w.Npad + co];
      } else {
        const int a = (3 - ky) & 1, dy = (3 - ky) >> 1, b = (3 - kx) & 1, dx = (3 - kx) >> 1;
        dst[(((size_t)ci * w.Co + co) * 4 + ky) * 4 + kx] =
            src[((size_t)(a * 2 + b) * 4 * w.Cip + (dy * 2 + dx) * w.Cip + cb) * w.Npad + co];
      }
    }
  }
}
static int ftap(int a, int k) { return a ? (k + 1) >> 1 : k >> 1; }
size_t tail_fold_offset(const WShape& w, int phase) {
  size_t off = 0;
  for (int p = 0; p < phase && p < 4; ++p) off += (size_t)(2 + (p >> 1)) * (2 + (p & 1)) * w.Cip * w.Npad;
  return off;
}
void tail_fold_weights(Stream&, const WShape& w, const float* src, float* dst) {
  const size_t pt = (size_t)w.Cip * w.Npad;
  std::memset(dst, 0, tail_fold_offset(w, 4) * sizeof(float));
  for (int ph = 0; ph < 4; ++ph) { const int a = ph >> 1, b = ph & 1;
    float* d = dst + tail_fold_offset(w, ph);
    for (int ky = 0; ky < 4; ++ky) for (int kx = 0; kx < 4; ++kx) {
      float* dd = d + (size_t)(ftap(a, ky) * (2 + b) + ftap(b, kx)) * pt;
      const float* ss = src + (size_t)(ky * 4 + kx) * pt;
      for (size_t i = 0; i < pt; ++i) dd[i] += ss[i]; } }
}
void tail_unfold_wgrad(Stream&, const WShape& w, const float* src, float* dst) {
  const size_t pt = (size_t)w.Cip * w.Npad;
  std::memset(dst, 0, 16 * pt * sizeof(float));
  for (int ph = 0; ph < 4; ++ph) { const int a = ph >> 1, b = ph & 1;
    const float* sp = src + tail_fold_offset(w, ph);
    for (int ky = 0; ky < 4; ++ky) for (int kx = 0; kx < 4; ++kx) {
      const float* ss = sp + (size_t)(ftap(a, ky) * (2 + b) + ftap(b, kx)) * pt;
      float* dd = dst + (size_t)(ky * 4 + kx) * pt;
      for (size_t i = 0; i < pt; ++i) dd[i] += ss[i]; } }
}
size_t dgrad_elems(const WShape& w, int mode, int Cop, int Ndg) {
  switch (mode) { case 0: return (size_t)16 * Cop * Ndg; case 1: return (size_t)w.KH * w.KW * Cop * Ndg;
                  case 2: return (size_t)16 * Cop * Ndg; default: return (size_t)25 * Cop * Ndg; }
}
// source-indexed scatter (the HIP kernel is destination-indexed)
void repack_dgrad(Stream&, const WShape& w, int mode, int Cop, int Ndg, const float* src, float* dg) { SIM_TIMED;
  std::memset(dg, 0, dgrad_elems(w, mode, Cop, Ndg) * sizeof(float));
  auto W = [&](int ky, int kx, int ci, int co) -> float {
    if (w.kind == WK_CONV) return src[((size_t)(ky * w.KW + kx) * w.Cip + ci) * w.Npad + co];
    const int a = (3 - ky) & 1, dy = (3 - ky) >> 1, b = (3 - kx) & 1, dx = (3 - kx) >> 1;
    return src[((size_t)(a * 2 + b) * 4 * w.Cip + (dy * 2 + dx) * w.Cip + ci) * w.Npad + co];
  };
  // (Ndg < Cip: the operand of an input gradient formed for the leading channels only, ops.h repack_dgrad)
  // (modes 0-2 are permutations: every element has its own destination, the taps can go to different threads; the tail's folded
  // taps overlap -- serial, and small)
#pragma omp parallel for collapse(2) if (mode != 3)
  for (int ky = 0; ky < w.KH; ++ky) for (int kx = 0; kx < w.KW; ++kx) for (int ci = 0; ci < std::min(w.Cip, Ndg); ++ci) for (int co = 0; co < w.Co; ++co) {
    const float v = W(ky, kx, ci, co);
    if (mode == 0) {          // dX[2i+a] += dY[i + a - 1 + dy] * W[3 - a - 2dy]
      const int a = (3 - ky) & 1, dy = (3 - ky) >> 1, b = (3 - kx) & 1, dx = (3 - kx) >> 1;
      dg[((size_t)(a * 2 + b) * 4 * Cop + (dy * 2 + dx) * Cop + co) * Ndg + ci] = v;
    } else if (mode == 1) {
      dg[((size_t)((w.KH - 1 - ky) * w.KW + (w.KW - 1 - kx)) * Cop + co) * Ndg + ci] = v;
    } else if (mode == 2) {
      dg[((size_t)(ky * 4 + kx) * Cop + co) * Ndg + ci] = v;
    } else {                  // tail: tap r = a - ky + 3 for a in {0,1}
      for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) {
        const int r = a - ky + 3, c = b - kx + 3;
        dg[((size_t)(r * 5 + c) * Cop + co) * Ndg + ci] += v;
      }
    }
  }
}


// ---- amax slots (ops.h ConvFwdArgs::x_amax): the producers fold max |v| of what they wrote into the slot (entry 0 here), the
// consumers CHECK it against the operand they are about to read -- a slot that was not zeroed, not filled, or filled from another
// tensor is an engine bug the simulator must catch (on the device it would silently mis-scale the fp16 planes)
static float sim_amax(const float* p, size_t n) {
  float m = 0.f;
  for (size_t i = 0; i < n; ++i) m = std::max(m, std::fabs(p[i]));
  return m;
}
static void sim_slot_fold(float* slot, float v) { if (slot) slot[0] = std::max(slot[0], v); }
float sim_slot_max(const float* slot) {
  float m = 0.f;
  for (int i = 0; i < AMAX_SLOT; ++i) m = std::max(m, slot[i]);
  return m;
}
void sim_slot_check(const float* slot, const float* x, size_t rows, int C, size_t rs, int batch, size_t bs, const char* what) {
  if (!slot) return;
  float actual = 0.f;
  for (int b = 0; b < batch; ++b)
    for (size_t r = 0; r < rows; ++r) actual = std::max(actual, sim_amax(x + (size_t)b * bs + r * rs, (size_t)C));
  const float have = sim_slot_max(slot);
  if (getenv("SWN_SIM_SLOT_REPORT") && actual > 0.f && have > (float)atof(getenv("SWN_SIM_SLOT_REPORT")) * actual)
    fprintf(stderr, "[slot] %-16s slot %.4e  operand amax %.4e  ratio %.1f  rows %zu C %d batch %d\n", what, have, actual, have / actual, rows, C, batch);
  // a slot BOUNDS its operand: equal for a tensor with one producer, larger for a slice of a concatenation buffer whose slot
  // covers all slices -- but never smaller (overflow of the fp16 planes on the device) and never absurdly larger (a stale slot)
  if (!(have >= actual) || (actual > 0.f && have > 4096.f * actual))
    throw Error(1, std::string("hostsim ") + what + ": amax slot holds " + std::to_string(have) + ", the operand's amax is " + std::to_string(actual));
}
void sim_fold_view(float* slot, const TView& v) {
  if (!slot) return;
  float m = 0.f;
  for (size_t e = 0; e < v.pixels(); ++e) m = std::max(m, sim_amax(v.p + e * v.cs, (size_t)v.C));
  sim_slot_fold(slot, m);
}
void tensor_amax(Stream&, const TView& x, float* slot, float floor) {
  for (int i = 0; i < AMAX_SLOT; ++i) slot[i] = 0.f;
  slot[0] = floor;
  sim_fold_view(slot, x);
}
// nearest fp16 value of x, as a float (round to nearest even; subnormals below 2^-14; overflow -> inf)
static float sim_f16_rn(float x) {
  if (x == 0.f || !std::isfinite(x)) return x;
  int e;
  std::frexp(std::fabs(x), &e);                       // |x| = m 2^e, m in [0.5, 1)
  const int ue = std::max(e - 11, -24);               // exponent of the ulp
  const float r = std::ldexp((float)std::nearbyint(std::ldexp((double)x, -ue)), ue);
  return std::fabs(r) > 65504.f ? std::copysign(INFINITY, x) : r;
}
// what the device's pair-form store keeps of a plane tensor: k from the bound gain * amax(input slot) (wino.hip pair_scale_exp)
static void sim_pair_round(float* P, size_t n, const float* in_amax, float gain, int* kscale_out, const TView& src, const char* what) {
  const float have = sim_slot_max(in_amax);
  float actual = 0.f;
  for (size_t e = 0; e < src.pixels(); ++e) actual = std::max(actual, sim_amax(src.p + e * src.cs, (size_t)src.C));
  if (!(have >= actual) || (actual > 0.f && have > 4096.f * actual))
    throw Error(1, std::string("hostsim ") + what + ": the input's amax slot holds " + std::to_string(have) + ", its amax is " + std::to_string(actual));
  const float m = have * gain;
  int k = 0;
  if (m > 0.f && m <= 3.0e38f) { int e; std::frexp(m, &e); k = 14 - (e - 1); k = std::max(-100, std::min(100, k)); }
  if (kscale_out) *kscale_out = k;
  const float pm = sim_amax(P, n);
  if (std::ldexp(pm, k) >= 65504.f) throw Error(1, std::string("hostsim ") + what + ": pair-form plane overflows fp16 under its bound");
  if (getenv("SWN_SIM_SLOT_REPORT"))
    fprintf(stderr, "[pair] %-24s slot/input %.1f  plane amax * 2^k = 2^%.1f (top 2^15)  n %zu\n", what, actual > 0 ? have / actual : 0.f,
            pm > 0 ? std::log2(std::ldexp(pm, k)) : 0.f, n);
  for (size_t i = 0; i < n; ++i) {
    const float x = std::ldexp(P[i], k);
    const float h = sim_f16_rn(x), l = sim_f16_rn(x - h);
    P[i] = std::ldexp(h + l, -k);
  }
}
void wino_input_transform(Stream&, int m, int r, const TView& x, int pad, int pad_mode, int Th, int Tw, float* V, float* amax_out,
                          const float* in_amax, int* kscale_out) { SIM_TIMED;
  wino_input_transform_impl(m, r, x, pad, pad_mode, Th, Tw, V);
  const int A = m + r - 1;
  const size_t n = (size_t)A * A * x.N * Th * Tw * x.C;
  if (in_amax) { sim_pair_round(V, n, in_amax, 100.f, kscale_out, x, "wino_input_transform"); return; }
  if (amax_out && !(m == 2 && r == 3)) sim_slot_fold(amax_out, sim_amax(V, n));
}
void wino_dy_transform(Stream&, int m, int r, const TView& dy, int Th, int Tw, float* dM, float* amax_out, const float* in_amax, int* kscale_out) { SIM_TIMED;
  wino_dy_transform_impl(m, r, dy, Th, Tw, dM);
  const int A = m + r - 1;
  const size_t n = (size_t)A * A * dy.N * Th * Tw * dy.C;
  if (in_amax) { sim_pair_round(dM, n, in_amax, (m == 4 && r == 3) ? 225.f : ((m == 3 && r == 4) ? 49.f : 16.f), kscale_out, dy, "wino_dy_transform"); return; }
  if (amax_out && !(m == 2 && r == 3)) sim_slot_fold(amax_out, sim_amax(dM, n));
}
void tailw_dy_transform(Stream&, const TView& dy, int Th, int Tw, int Npad, float* dM, float* amax_out, const float* in_amax, int* kscale_out) {
  tailw_dy_transform_impl(dy, Th, Tw, Npad, dM);
  const size_t n = (size_t)36 * dy.N * Th * Tw * 4 * Npad;
  if (in_amax) { sim_pair_round(dM, n, in_amax, 225.f, kscale_out, dy, "tailw_dy_transform"); return; }
  sim_slot_fold(amax_out, sim_amax(dM, n));
}
void wino_s2_input_transform(Stream&, const TView& x, int Th, int Tw, float* V, float* amax_out, const float* in_amax, int* kscale_out) {
  wino_s2_input_transform_impl(x, Th, Tw, V);
  const size_t n = (size_t)25 * x.N * Th * Tw * 4 * x.C;
  if (in_amax) { sim_pair_round(V, n, in_amax, 9.f, kscale_out, x, "wino_s2_input_transform"); return; }
  sim_slot_fold(amax_out, sim_amax(V, n));
}

}  // namespace swn
