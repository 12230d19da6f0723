// swapnet_amd -- engine core: arenas, op tape, accumulate planner.
#include "engine.h"
#include <memory>

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace swn {

// ---- routing trace (ops.h route_*) ------------------------------------------------------
namespace {
int g_route = 0;
std::string g_route_label = "-";
char g_route_phase = '-';
std::vector<std::string> g_route_lines;
std::map<std::string, int> g_route_seen;
}  // namespace
void route_enable(int on) {
  g_route = on;
  if (on) { g_route_lines.clear(); g_route_seen.clear(); g_route_label = "-"; g_route_phase = '-'; }
}
bool route_on() { return g_route != 0; }
void route_label(const char* label, char phase) { g_route_label = label ? label : "-"; g_route_phase = phase; }
void route_note(const char* kernel) {
  if (!g_route) return;
  std::string line = g_route_label + " " + std::string(1, g_route_phase) + " " + kernel;
  if (g_route_seen.emplace(line, 1).second) g_route_lines.push_back(line);
}
int route_report(char* buf, int len) {
  std::string out;
  for (auto& l : g_route_lines) { out += l; out += "\n"; }
  if (buf && len > 0) { strncpy(buf, out.c_str(), len - 1); buf[len - 1] = 0; }
  return (int)out.size();
}

// ---------------------------------------------------------------------------------------
Ctx::Ctx(void* stream, size_t ws_bytes) {
  s.handle = stream;
  s.ws_bytes = ws_bytes;
  s.ws = static_cast<char*>(dev_alloc(ws_bytes));
}
Ctx::Ctx(const Stream& shared) : s(shared), owns_ws(false) {}
Ctx::~Ctx() {
  // nothing may still be running on a stream whose events and handle are about to be destroyed (round-4 advice)
  try {
    if (owned_comm_stream) stream_sync(comm_stream);
    if (has_side) stream_sync(side);
  } catch (...) {}
  release(allocs);
  if (owns_ws) dev_free(s.ws);
  if (owned_comm_stream) {
    for (void* e : comm_events) event_destroy(e);
    event_destroy(comm_join_event);
    stream_destroy(owned_comm_stream);
  }
  if (has_side) {
    dev_free(side.ws);
    for (void* e : fork_events) event_destroy(e);
    event_destroy(join_event);
    stream_destroy(owned_side_stream);
  }
}
void Ctx::enable_side(int device) {
  if (has_side || !is_device_build()) return;
  owned_side_stream = stream_create(device);
  side.handle = owned_side_stream;
  side.ws_bytes = s.ws_bytes;
  side.ws = static_cast<char*>(dev_alloc(side.ws_bytes));
  for (int i = 0; i < 8; ++i) fork_events.push_back(event_create());
  join_event = event_create();
  has_side = true;
}
Stream& Ctx::fork_side() {
  void* ev = fork_events[fork_i++ % fork_events.size()];
  event_record(ev, s);
  stream_wait_event(side, ev);
  side_dirty = true;
  return side;
}
void Ctx::join_side() {
  if (!has_side || !side_dirty) return;
  event_record(join_event, side);
  stream_wait_event(s, join_event);
  side_dirty = false;
}
void Ctx::attach_comm(AllReduceFn fn, void* comm, int world) {
  comm_join();
  if (!fn) {
    // detaching: the caller is about to destroy the communicator -- no collective of ours may still be in flight on its stream
    if (owned_comm_stream) stream_sync(comm_stream);
    comm_fn = nullptr; comm_handle = nullptr; comm_world = 1;
    return;
  }
  if (world < 1) throw Error(1, "attach_comm: world size must be >= 1");
  comm_fn = fn; comm_handle = comm; comm_world = world;
  if (is_device_build() && !owned_comm_stream) {
    owned_comm_stream = stream_create(device_index);
    comm_stream.handle = owned_comm_stream;       // no scratch: nothing launched there uses the split-K workspace
    for (int i = 0; i < 8; ++i) comm_events.push_back(event_create());
    comm_join_event = event_create();
  }
}
Stream& Ctx::comm_fork() {
  if (!owned_comm_stream) return s;               // host simulator: one in-order stream
  void* ev = comm_events[comm_ev_i++ % comm_events.size()];
  event_record(ev, s);
  stream_wait_event(comm_stream, ev);
  comm_dirty = true;
  return comm_stream;
}
void Ctx::comm_join() {
  if (!owned_comm_stream || !comm_dirty) return;
  event_record(comm_join_event, comm_stream);
  stream_wait_event(s, comm_join_event);
  comm_dirty = false;
}
void Ctx::all_reduce_sum(Stream& on, float* buf, size_t count) {
  if (!comm_fn) throw Error(1, "all_reduce: no communicator attached (swn_ctx_attach_comm)");
  if (count == 0) return;
  // ncclFloat32 = 7, ncclSum = 0 (nccl.h / rccl.h)
  const int rc = comm_fn(buf, buf, count, 7, 0, comm_handle, on.handle);
  if (rc != 0) throw Error(2, "all_reduce: the attached all-reduce returned " + std::to_string(rc));
}
void* Ctx::alloc(size_t bytes) {
  void* p = dev_alloc(bytes);
  (sink ? *sink : allocs).push_back({p, bytes});
  bytes_allocated += bytes;
  return p;
}
void Ctx::release(AllocList& list) {
  for (auto& a : list) { dev_free(a.first); bytes_allocated -= a.second; }
  list.clear();
}
Model::~Model() {
  for (void*& g : step_graph_) { graph_destroy(g); g = nullptr; }
  if (cap_stream_) stream_destroy(cap_stream_);
  gp_.reset(); D3_.reset(); G.reset(); D2.reset(); D1.reset();
  if (ctx) ctx->release(owned_allocs);
}

Var Var::slice(int c0, int c) const {
  Var r = *this;
  r.v = v.slice(c0, c);
  if (has_grad) r.g = g.slice(c0, c);
  return r;
}
Var Var::batch(int n0, int n) const {
  Var r = *this;
  const size_t off = (size_t)n0 * v.H * v.W * v.cs;
  r.v.p = v.p + off; r.v.N = n;
  if (has_grad) { r.g.p = g.p + off; r.g.N = n; }
  return r;
}

// ---------------------------------------------------------------------------------------
int ParamArena::add_weight(const std::string& name, int kind, int Co, int Ci, int KH, int KW, int Cip,
                           const std::vector<int32_t>* cimap) {
  if (frozen) {
    auto it = index.find(name);
    if (it == index.end()) throw Error(1, "shared arena has no parameter " + name);
    const WShape& e = params[it->second].ws;
    if (e.kind != kind || e.Co != Co || e.Ci != Ci || e.KH != KH || e.KW != KW || e.Cip != Cip)
      throw Error(1, "shared arena: shape mismatch for " + name);
    return it->second;
  }
  ParamDesc d;
  d.name = name;
  d.ws.kind = kind; d.ws.Co = Co; d.ws.Ci = Ci; d.ws.KH = KH; d.ws.KW = KW; d.ws.Cip = Cip;
  d.ws.Npad = round_up(Co, 4);
  if (cimap) d.cimap = *cimap;
  d.elems = packed_elems(d.ws);
  d.off = n;
  n += (d.elems + 3) / 4 * 4;
  index[name] = (int)params.size();
  params.push_back(d);
  return (int)params.size() - 1;
}
int ParamArena::add_bias(const std::string& name, int nb) {
  if (frozen) {
    auto it = index.find(name);
    if (it == index.end() || params[it->second].n_logical != nb) throw Error(1, "shared arena: bad bias " + name);
    return it->second;
  }
  ParamDesc d;
  d.name = name; d.is_bias = true; d.n_logical = nb;
  d.elems = round_up(nb, 4);
  d.off = n;
  n += d.elems;
  index[name] = (int)params.size();
  params.push_back(d);
  return (int)params.size() - 1;
}
void ParamArena::allocate(Ctx& c) {
  const size_t bytes = std::max<size_t>(n, 4) * sizeof(float);
  w = static_cast<float*>(c.alloc(bytes));
  g = static_cast<float*>(c.alloc(bytes));
  m = static_cast<float*>(c.alloc(bytes));
  v = static_cast<float*>(c.alloc(bytes));
  for (auto& d : params) {
    if (!d.cimap.empty()) {
      int32_t* dev = static_cast<int32_t*>(c.alloc(d.cimap.size() * sizeof(int32_t)));
      dev_upload(c.s, dev, d.cimap.data(), d.cimap.size() * sizeof(int32_t));
      d.ws.cimap = dev;
    }
  }
  frozen = true;
}

// ---------------------------------------------------------------------------------------
Var Net::alloc_var(int N, int H, int W, int C, bool need_grad) {
  if (C % 4) throw Error(1, "alloc_var: C must be a multiple of 4");
  Var r;
  const size_t bytes = (size_t)N * H * W * C * sizeof(float);
  r.v.p = static_cast<float*>(ctx.alloc(bytes));
  r.v.N = N; r.v.H = H; r.v.W = W; r.v.C = C; r.v.cs = C;
  r.vbase = r.v.p;
  r.has_grad = need_grad;
  if (need_grad) {
    r.g = r.v;
    r.g.p = static_cast<float*>(ctx.alloc(bytes));
    r.gbase = r.g.p;
  }
  return r;
}

size_t Net::reserve_dg(Op* op, size_t elems) {
  const size_t off = dg_n;
  dg_n += (elems + 3) / 4 * 4;
  dg_layout.push_back({op, off});
  return off;
}

namespace {
struct ConvGeom {
  Gather fwd;       // forward gather (Ho,Wo filled)
  int Ho, Wo;
};
ConvGeom conv_geom(ConvKind kind, int H, int W) {
  ConvGeom c;
  Gather& g = c.fwd;
  switch (kind) {
    case CK_K4S2: g.KH = g.KW = 4; g.stride = 2; g.pad_t = g.pad_l = 1; c.Ho = H / 2; c.Wo = W / 2; break;
    case CK_K3S1_REFLECT: g.KH = g.KW = 3; g.stride = 1; g.pad_t = g.pad_l = 1; g.pad_mode = PAD_REFLECT; c.Ho = H; c.Wo = W; break;
    case CK_K3S1_ZERO: g.KH = g.KW = 3; g.stride = 1; g.pad_t = g.pad_l = 1; c.Ho = H; c.Wo = W; break;
    case CK_K4S1: g.KH = g.KW = 4; g.stride = 1; g.pad_t = g.pad_l = 1; c.Ho = H - 1; c.Wo = W - 1; break;
    case CK_K1S1: g.KH = g.KW = 1; g.stride = 1; g.pad_t = g.pad_l = 0; c.Ho = H; c.Wo = W; break;
    case CK_TAIL_UP:   // Upsample(x2) + ZeroPad2d((1,0,1,0)) + Conv(k4,p1): pad 2 top/left in upsampled coords
      g.KH = g.KW = 4; g.stride = 1; g.pad_t = g.pad_l = 2; g.ups = 1; c.Ho = H * 2; c.Wo = W * 2; break;
  }
  g.Ho = c.Ho; g.Wo = c.Wo;
  return c;
}
}  // namespace


// ---- strided Winograd F(4x4, 2x2) for the k4 s2 p1 convolutions and their transposes (ops.h wino_s2_*, wino.hip) ----------
// Used where the activation side dominates: the transformed filters are 6.25x the weights (per GEMM direction) and are
// re-derived every optimizer step, so the 8-16 M-parameter layers at 4x4 / 8x8 maps stay on the direct kernels.
namespace {
bool s2_wino_wanted(int Cfine, int Ccoarse, int Hc, int Wc) {
  const bool off = (getenv("SWN_WINOGRAD") && atoi(getenv("SWN_WINOGRAD")) == 0) ||       // read per layer built (tests toggle it)
                   (getenv("SWN_WINO_S2") && atoi(getenv("SWN_WINO_S2")) == 0);
  if (off) return false;
  const int minc_env = getenv("SWN_WINO_MINC") ? atoi(getenv("SWN_WINO_MINC")) : 0;      // (tests: small channel counts too)
  const int minc = minc_env > 0 ? minc_env : 256;
  if (Ccoarse < minc || Cfine < 32 || Cfine % 4 || Ccoarse % 16 || Hc < 2 || Wc < 2) return false;
  // coarse maps of at least 16 x 16: below that the GEMMs are a handful of tiles (nothing to gain), and the 8 x 8 ... 2 x 2
  // levels of the pix2pix U-Net sit in front of InstanceNorms over 64 ... 4 pixels, which amplify the (2.5x larger) round-off
  // of the Winograd form: with them on it, the texture generator's gradients were 1e-4 ... 3e-4 off the pinned float64 oracle,
  // without 6e-5 (tests/test_pattern_replay.py)
  if (minc_env <= 0 && Hc * Wc < 256) return false;
  return (size_t)16 * Cfine * Ccoarse <= ((size_t)1 << 22);
}
TView plane_mat(float* p, size_t T, int C) {
  TView v; v.p = p; v.N = 1; v.H = 1; v.W = (int)T; v.C = C; v.cs = C; return v;     // T x C matrix
}
}  // namespace

// ---- Conv2d ---------------------------------------------------------------------------
// y.v receives act(conv(x)+bias).  In backward y.g is the gradient w.r.t. that output.
// Column tile of a pre-cut operand written by the 6-point FILTER TRANSFORM (ops.h wino_filter_transform_pc): that kernel lays out
// tiles of 64 or 128 columns only.  conv_precut_tile answers 192 for widths in (128, 192] (the tile of the tail conv's input
// gradient, produced by conv_precut) -- a stride-1 Winograd layer of such a width keeps its fp32 U and the kernels that read it,
// and with them fp32 planes: planning and launch agree instead of failing at the first operand refresh (latent: no network of
// the reference has a 3x3 / 4x4 stride-1 conv between 129 and 192 channels; swn_op_conv can ask for one).
static int wino_precut_tile(int xC, int Npad) {
  const int t = conv_precut_tile(xC, Npad);
  return (t == 64 || t == 128) ? t : 0;
}
static bool wino_fwd_takes_pairs(int xC, int Npad) { return wino_precut_tile(xC, Npad) != 0 && conv_fwd_takes_pairs(xC, Npad); }

void Net::conv(const std::string& name, const Var& x, const Var& y, ConvKind kind, int Ci, int Co, bool bias,
               int actf, const std::vector<int32_t>* cimap, bool x_is_input, int dgrad_C) {
  const int KH = kind == CK_K1S1 ? 1 : ((kind == CK_K3S1_REFLECT || kind == CK_K3S1_ZERO) ? 3 : 4);
  const ConvGeom geo = conv_geom(kind, x.v.H, x.v.W);
  if (y.v.H != geo.Ho || y.v.W != geo.Wo) throw Error(1, "conv " + name + ": output view has the wrong size");
  const int Cip = x.v.C;
  if (Ci > Cip) throw Error(1, "conv " + name + ": Ci exceeds the input view's channels");
  const int wi = arena.add_weight(name + ".weight", WK_CONV, Co, Ci, KH, KH, Cip, cimap);
  const int bi = bias ? arena.add_bias(name + ".bias", Co) : -1;
  const int Cop = round_up(Co, 4);
  if (y.v.C != Cop) throw Error(1, "conv " + name + ": output view must have round_up(Co,4) channels");
  note_act(actf, y.v);

  auto op = std::make_unique<Op>();
  Op* self = op.get();
  op->label = name;
  op->param_off = arena.params[wi].off;
  op->reads_net_input = x_is_input;
  ParamArena* A = &arena;
  const Gather gf = geo.fwd;
  const TView xv = x.v, yv = y.v;
  // PatchGAN's 1-channel head conv (discriminators.py:131): taps on the N axis (ops.h head_*): the input
  // is read once by a 1x1 conv with N = 16 instead of 16 times by an N = 1 implicit GEMM.
  const bool head_on = !(getenv("SWN_HEAD_TAPN") && atoi(getenv("SWN_HEAD_TAPN")) == 0);
  if (head_on && kind == CK_K4S1 && Co == 1 && Cip % 32 == 0 && actf == ACT_NONE && x.has_grad == y.has_grad) {
    const size_t wt_off = reserve_dg(self, (size_t)Cip * 16), wt2_off = reserve_dg(self, (size_t)16 * Cip);
    const size_t dwt_off = reserve_dg(self, (size_t)Cip * 16);
    note_writer(y.vbase, false);
    Var Z = alloc_var(xv.N, xv.H, xv.W, 16, false);          // Z forward, dZ backward (same scratch)
    const TView zv = Z.v, ygv = y.g, xgv = x.g;
    const bool has_grad = y.has_grad;
    op->repack = [=](Net& n) {
      const ParamDesc& wd = A->params[wi];
      head_pack(n.ctx.s, wd.ws, A->w + wd.off, n.dg + wt_off, n.dg + wt2_off);
    };
    op->fwd = [=](Net& n) {
      n.need(self);
      ConvFwdArgs a;
      a.x = xv; a.g.Ho = xv.H; a.g.Wo = xv.W;               // 1x1, stride 1
      a.w = n.dg + wt_off; a.Npad = 16; a.Cout = 16; a.y = zv;
      conv_fwd(n.ctx.s, a);
      head_gather(n.ctx.s, zv, bi >= 0 ? A->w + A->params[bi].off : nullptr, yv);
    };
    if (has_grad) op->grad_targets.push_back(x);
    op->bwd = [=](Net& n, Op& me, bool wgrad, bool igrad) {
      if (!has_grad) return;
      const ParamDesc& wd = A->params[wi];
      head_scatter(n.ctx.s, ygv, zv);
      if (wgrad) {
        Stream& sw = n.wgrad_stream();
        ConvWgradArgs wa;
        wa.x = xv; wa.g.Ho = xv.H; wa.g.Wo = xv.W; wa.dy = zv;
        wa.dw = n.dg + dwt_off; wa.Npad = 16; wa.Cout = 16;
        conv_wgrad(sw, wa);
        head_unpack_grad(sw, wd.ws, n.dg + dwt_off, A->g + wd.off);
        if (bi >= 0) n.bias_grad_of(sw, ygv, A->g + A->params[bi].off);
      }
      if (me.reads_net_input && !igrad) return;
      ConvFwdArgs d;
      d.x = zv; d.g.Ho = xv.H; d.g.Wo = xv.W;
      d.w = n.dg + wt2_off; d.Npad = Cip; d.Cout = Cip; d.y = xgv; d.accumulate = me.acc.empty() ? 0 : me.acc[0];
      conv_fwd(n.ctx.s, d);
    };
    ops.push_back(std::move(op));
    return;
  }
  // k4 s2 convs with enough channels: strided Winograd F(4x4,2x2) (four polyphase 2x2 convolutions in one batched GEMM)
  // (never the layers that read a network input: their buffers carry layout pad channels -- 19 -> 32, 22 -> 32 -- that would
  // be transformed for nothing, and the channel threshold of the test routing must not reach them through the padding)
  if (kind == CK_K4S2 && s2_wino && !x_is_input && s2_wino_wanted(Cip, Cop, y.v.H, y.v.W) && Co % 4 == 0) {
    const int sP = 25, sTh = ceil_div(y.v.H, 4), sTw = ceil_div(y.v.W, 4), CV = 4 * Cip;
    const size_t sT = (size_t)x.v.N * sTh * sTw;
    const bool want_dx = x.has_grad && y.has_grad;
    const size_t uf_off = reserve_dg(self, (size_t)sP * CV * Cop);                      // U  [25][4 Cip][Cop]
    const size_t ut_off = want_dx ? reserve_dg(self, (size_t)sP * Cop * CV) : 0;        // U^T[25][Cop][4 Cip]
    float* keepV = (keep_wino_inputs && y.has_grad) ? static_cast<float*>(ctx.alloc(sP * sT * CV * sizeof(float))) : nullptr;
    wsV_need = std::max(wsV_need, sP * sT * (size_t)std::max(CV, Cop));
    wsM_need = std::max(wsM_need, sP * sT * (size_t)std::max(CV, Cop));
    wsU_need = std::max(wsU_need, (size_t)sP * CV * Cop);
    const int pcf = conv_precut_tile(CV, Cop), pct = want_dx ? conv_precut_tile(Cop, CV) : 0;
    const size_t pcf_bs = pcf ? conv_precut_elems(CV, Cop, pcf) : 0, pct_bs = pct ? conv_precut_elems(Cop, CV, pct) : 0;
    const size_t pcf_off = pcf ? reserve_dgp(pcf_bs * sP) : 0, pct_off = pct ? reserve_dgp(pct_bs * sP) : 0;
    const size_t slV = reserve_slot(), slD = reserve_slot();      // amax of V (forward planes) and of dM (transformed dY)
    note_writer(y.vbase, false);
    // pair-form planes (ops.h wino_input_transform) where every GEMM that reads them takes the kernels that can, and -- decided
    // per pass -- the transform's input has a complete amax slot to bound the planes with
    const bool pairV_ok = conv_fwd_takes_pairs(CV, Cop) && (!y.has_grad || conv_wgrad_takes_pairs(sT, CV, Cop));
    const bool pairD_ok = conv_wgrad_takes_pairs(sT, CV, Cop) && (!(x.has_grad && y.has_grad) || conv_fwd_takes_pairs(Cop, CV));
    const size_t kV = reserve_k(), kD = reserve_k();
    const float* xbase = x.vbase; const float* ygbase = y.gbase;
    // dM = A dY A^T serves the weight gradient (side stream) and the input gradient (main stream): one buffer per layer
    float* keepdM = (y.has_grad && want_dx && share_dy()) ? static_cast<float*>(ctx.alloc(sP * sT * Cop * sizeof(float))) : nullptr;
    op->repack = [=](Net& n) {
      const ParamDesc& wd = A->params[wi];
      const float* wmax = nullptr;       // U^T holds U's values: one amax pass serves both operands (ops.h conv_precut amax_io)
      wino_s2_filter_transform(n.ctx.s, wd.ws, 0, A->w + wd.off, n.dg + uf_off);
      if (pcf) conv_precut(n.ctx.s, n.dg + uf_off, CV, Cop, pcf, sP, (size_t)CV * Cop, n.dgp + pcf_off, &wmax);
      if (want_dx) {
        wino_s2_filter_transform(n.ctx.s, wd.ws, 1, A->w + wd.off, n.dg + ut_off);
        if (pct) conv_precut(n.ctx.s, n.dg + ut_off, Cop, CV, pct, sP, (size_t)Cop * CV, n.dgp + pct_off, &wmax);
      }
    };
    op->fwd = [=](Net& n) {
      n.need(self);
      float* V = keepV ? keepV : n.wsV;
      const float* xin = pairV_ok ? n.slot_if_complete(xbase) : nullptr;
      wino_s2_input_transform(n.ctx.s, xv, sTh, sTw, V, n.amax + slV, xin, n.kscale + kV);
      ConvFwdArgs g;
      g.x = plane_mat(V, sT, CV); g.g.Ho = 1; g.g.Wo = (int)sT;
      g.w = n.dg + uf_off; g.Npad = Cop; g.Cout = Co;
      g.y = plane_mat(n.wsM, sT, Cop);
      g.batch = sP; g.x_bs = sT * CV; g.w_bs = (size_t)CV * Cop; g.y_bs = sT * Cop;
      if (xin) g.x_pair_k = n.kscale + kV; else g.x_amax = n.amax + slV;
      if (pcf) { g.wpc = n.dgp + pcf_off; g.wpc_bn = pcf; g.wpc_bs = pcf_bs; }
      conv_fwd(n.ctx.s, g);
      wino_output_transform(n.ctx.s, 4, 2, n.wsM, Cop, sTh, sTw, bi >= 0 ? A->w + A->params[bi].off : nullptr, actf, yv, Co, 0);
    };
    Var scratch;
    if (y.has_grad && actf != ACT_NONE) scratch = alloc_var(yv.N, yv.H, yv.W, Cop, false);
    if (want_dx) op->grad_targets.push_back(x);
    const TView ygv = y.g, xgv = x.g, scr = scratch.v;
    const bool has_ygrad = y.has_grad;
    op->bwd = [=](Net& n, Op& me, bool wgrad, bool igrad) {
      if (!has_ygrad) return;
      TView dY = ygv;
      // (a fused activation's dR has no amax slot in this branch: its planes stay fp32 and the GEMM takes their amax)
      if (actf != ACT_NONE) { act_bwd(n.ctx.s, ygv, yv, scr, actf, 0); dY = scr; }
      const ParamDesc& wd = A->params[wi];
      const bool dx_now = want_dx && !(me.reads_net_input && !igrad);
      const bool shared = keepdM && wgrad && dx_now;       // one transform of dY on the main stream, read by both gradients
      const float* xin = pairV_ok ? n.slot_if_complete(xbase) : nullptr;
      const float* din = (pairD_ok && actf == ACT_NONE) ? n.slot_if_complete(ygbase) : nullptr;
      if (shared) wino_dy_transform(n.ctx.s, 4, 2, dY, sTh, sTw, keepdM, n.amax + slD, din, n.kscale + kD);
      if (wgrad) {
        Stream& sw = n.wgrad_stream();
        float* V = keepV ? keepV : n.wsV;
        float* dM = shared ? keepdM : n.wgrad_planes(sw);
        if (!keepV) wino_s2_input_transform(sw, xv, sTh, sTw, V, n.amax + slV, xin, n.kscale + kV);
        if (!shared) wino_dy_transform(sw, 4, 2, dY, sTh, sTw, dM, n.amax + slD, din, n.kscale + kD);
        ConvWgradArgs g;
        g.x = plane_mat(V, sT, CV); g.g.Ho = 1; g.g.Wo = (int)sT;
        g.dy = plane_mat(dM, sT, Cop);
        g.dw = n.wsU; g.Npad = Cop; g.Cout = Co;
        g.batch = sP; g.x_bs = sT * CV; g.dy_bs = sT * Cop; g.dw_bs = (size_t)CV * Cop;
        if (xin) g.x_pair_k = n.kscale + kV; else g.x_amax = n.amax + slV;
        if (din) g.dy_pair_k = n.kscale + kD; else g.dy_amax = n.amax + slD;
        conv_wgrad(sw, g);
        wino_s2_filter_grad(sw, wd.ws, n.wsU, A->g + wd.off);
        if (bi >= 0) n.bias_grad_of(sw, dY, A->g + A->params[bi].off);
      }
      if (!dx_now) return;
      float* dMx = shared ? keepdM : n.wsV;
      if (!shared) wino_dy_transform(n.ctx.s, 4, 2, dY, sTh, sTw, dMx, n.amax + slD, din, n.kscale + kD);
      ConvFwdArgs g;
      g.x = plane_mat(dMx, sT, Cop); g.g.Ho = 1; g.g.Wo = (int)sT;
      g.w = n.dg + ut_off; g.Npad = CV; g.Cout = CV;
      g.y = plane_mat(n.wsM, sT, CV);
      g.batch = sP; g.x_bs = sT * Cop; g.w_bs = (size_t)Cop * CV; g.y_bs = sT * CV;
      if (din) g.x_pair_k = n.kscale + kD; else g.x_amax = n.amax + slD;
      if (pct) { g.wpc = n.dgp + pct_off; g.wpc_bn = pct; g.wpc_bs = pct_bs; }
      conv_fwd(n.ctx.s, g);
      wino_s2_input_adjoint(n.ctx.s, n.wsM, Cip, sTh, sTw, xgv, nullptr, me.acc.empty() ? 0 : me.acc[0]);
    };
    ops.push_back(std::move(op));
    return;
  }
  // The tail conv in Winograd form (ops.h tailw_*): the four folded sub-pixel phases share one F(4x4,3x3) input transform of the
  // un-upsampled map and one batched GEMM with N = 4 Npad; forward and weight gradient (the input gradient stays the folded
  // 5x5 stride-2 conv on the ring kernel: its Winograd form would be bound by the adjoint transform of a 0.9 GB operand)
  if (kind == CK_TAIL_UP) {
    const int twminc_env = getenv("SWN_WINO_MINC") ? atoi(getenv("SWN_WINO_MINC")) : 0;
    const bool tw_on = !(getenv("SWN_WINOGRAD") && atoi(getenv("SWN_WINOGRAD")) == 0);
    if (tw_on && Cip % 32 == 0 && Cip >= (twminc_env > 0 ? twminc_env : 64) && x.v.H >= 4 && x.v.W >= 4 && Cop <= 32) {
      const ParamDesc wd0 = arena.params[wi];
      const int tP = 36, tTh = ceil_div(x.v.H, 4), tTw = ceil_div(x.v.W, 4), N4 = 4 * Cop;
      const size_t tT = (size_t)x.v.N * tTh * tTw;
      const size_t fe = tail_fold_offset(wd0.ws, 4);
      const size_t fold_off = reserve_dg(self, fe), dfold_off = reserve_dg(self, fe);
      const size_t tu_off = reserve_dg(self, (size_t)tP * Cip * N4);
      float* keepV = (keep_wino_inputs && y.has_grad) ? static_cast<float*>(ctx.alloc(tP * tT * Cip * sizeof(float))) : nullptr;
      wsV_need = std::max(wsV_need, tP * tT * (size_t)std::max(Cip, N4));
      wsM_need = std::max(wsM_need, tP * tT * (size_t)std::max(Cip, N4));
      wsU_need = std::max(wsU_need, (size_t)tP * Cip * N4);
      const int pcu = conv_precut_tile(Cip, N4);
      const size_t pcu_bs = pcu ? conv_precut_elems(Cip, N4, pcu) : 0, pcu_off = pcu ? reserve_dgp(pcu_bs * tP) : 0;
      const size_t slV = reserve_slot(), slD = reserve_slot();
      note_writer(y.vbase, false);
      const bool pairV_ok = conv_fwd_takes_pairs(Cip, N4) && (!y.has_grad || conv_wgrad_takes_pairs(tT, Cip, N4));
      const bool pairD_ok = conv_wgrad_takes_pairs(tT, Cip, N4);
      const size_t kV = reserve_k(), kD = reserve_k();
      const float* xbase = x.vbase;
      // input gradient: the folded 5x5 stride-2 conv over dR (32-channel buffer, see CopD below)
      const bool want_dx = x.has_grad && y.has_grad;
      const int CopD = (actf != ACT_NONE && want_dx && conv_precut_tile(32, Cip) == 192) ? 32 : Cop;
      const size_t dg_off = want_dx ? reserve_dg(self, dgrad_elems(wd0.ws, 3, CopD, Cip)) : 0;
      const int pc_d = want_dx ? conv_precut_tile(CopD, Cip) : 0;
      const size_t pcd_bs = pc_d ? conv_precut_elems(25 * CopD, Cip, pc_d) : 0, pcd_off = pc_d ? reserve_dgp(pcd_bs) : 0;
      op->repack = [=](Net& n) {
        const ParamDesc& wd = A->params[wi];
        tail_fold_weights(n.ctx.s, wd.ws, A->w + wd.off, n.dg + fold_off);
        tailw_filter_transform(n.ctx.s, wd.ws, n.dg + fold_off, n.dg + tu_off);
        if (pcu) conv_precut(n.ctx.s, n.dg + tu_off, Cip, N4, pcu, tP, (size_t)Cip * N4, n.dgp + pcu_off);
        if (want_dx) {
          repack_dgrad(n.ctx.s, wd.ws, 3, CopD, Cip, A->w + wd.off, n.dg + dg_off);
          if (pc_d) conv_precut(n.ctx.s, n.dg + dg_off, 25 * CopD, Cip, pc_d, 1, 0, n.dgp + pcd_off);
        }
      };
      op->fwd = [=](Net& n) {
        n.need(self);
        float* V = keepV ? keepV : n.wsV;
        const float* xin = pairV_ok ? n.slot_if_complete(xbase) : nullptr;
        wino_input_transform(n.ctx.s, 4, 3, xv, 1, PAD_ZERO, tTh, tTw, V, n.amax + slV, xin, n.kscale + kV);
        ConvFwdArgs g;
        g.x = plane_mat(V, tT, Cip); g.g.Ho = 1; g.g.Wo = (int)tT;
        g.w = n.dg + tu_off; g.Npad = N4; g.Cout = N4;
        g.y = plane_mat(n.wsM, tT, N4);
        g.batch = tP; g.x_bs = tT * Cip; g.w_bs = (size_t)Cip * N4; g.y_bs = tT * N4;
        if (xin) g.x_pair_k = n.kscale + kV; else g.x_amax = n.amax + slV;
        if (pcu) { g.wpc = n.dgp + pcu_off; g.wpc_bn = pcu; g.wpc_bs = pcu_bs; }
        conv_fwd(n.ctx.s, g);
        tailw_output_transform(n.ctx.s, n.wsM, tTh, tTw, Cop, bi >= 0 ? A->w + A->params[bi].off : nullptr, actf, yv, Co);
      };
      Var scratch;
      if (y.has_grad && actf != ACT_NONE) scratch = alloc_var(yv.N, yv.H, yv.W, CopD, false);
      if (want_dx) op->grad_targets.push_back(x);
      const TView ygv = y.g, xgv = x.g, scr = CopD != Cop ? scratch.v.slice(0, Cop) : scratch.v, scr_full = scratch.v;
      const bool has_ygrad = y.has_grad;
      const size_t scrSlot = (y.has_grad && actf != ACT_NONE) ? note_writer(scratch.v.p, true) : 0;
      op->bwd = [=](Net& n, Op& me, bool wgrad, bool igrad) {
        if (!has_ygrad) return;
        TView dY = ygv;
        const float* dy_slot = nullptr;
        if (actf != ACT_NONE) { act_bwd(n.ctx.s, ygv, yv, scr, actf, 0, n.amax + scrSlot); dY = scr; dy_slot = n.amax + scrSlot; }
        const ParamDesc& wd = A->params[wi];
        if (wgrad) {
          Stream& sw = n.wgrad_stream();
          float* V = keepV ? keepV : n.wsV;
          float* dM = n.wgrad_planes(sw);
          const float* xin = pairV_ok ? n.slot_if_complete(xbase) : nullptr;
          const float* din = pairD_ok ? dy_slot : nullptr;
          if (!keepV) wino_input_transform(sw, 4, 3, xv, 1, PAD_ZERO, tTh, tTw, V, n.amax + slV, xin, n.kscale + kV);
          tailw_dy_transform(sw, dY, tTh, tTw, Cop, dM, n.amax + slD, din, n.kscale + kD);
          ConvWgradArgs g;
          g.x = plane_mat(V, tT, Cip); g.g.Ho = 1; g.g.Wo = (int)tT;
          g.dy = plane_mat(dM, tT, N4);
          g.dw = n.wsU; g.Npad = N4; g.Cout = N4;
          g.batch = tP; g.x_bs = tT * Cip; g.dy_bs = tT * N4; g.dw_bs = (size_t)Cip * N4;
          if (xin) g.x_pair_k = n.kscale + kV; else g.x_amax = n.amax + slV;
          if (din) g.dy_pair_k = n.kscale + kD; else g.dy_amax = n.amax + slD;
          conv_wgrad(sw, g);
          tailw_filter_grad(sw, wd.ws, n.wsU, n.dg + dfold_off);
          tail_unfold_wgrad(sw, wd.ws, n.dg + dfold_off, A->g + wd.off);
          if (bi >= 0) n.bias_grad_of(sw, dY, A->g + A->params[bi].off);
        }
        if (!want_dx || (me.reads_net_input && !igrad)) return;
        ConvFwdArgs d;
        d.x = CopD != Cop ? scr_full : dY;
        d.g.KH = d.g.KW = 5; d.g.stride = 2; d.g.pad_t = d.g.pad_l = 1; d.g.Ho = xv.H; d.g.Wo = xv.W;
        d.w = n.dg + dg_off; d.Npad = Cip; d.Cout = Cip; d.y = xgv; d.accumulate = me.acc.empty() ? 0 : me.acc[0];
        d.x_amax = dy_slot;
        if (pc_d) { d.wpc = n.dgp + pcd_off; d.wpc_bn = pc_d; d.wpc_bs = pcd_bs; }
        conv_fwd(n.ctx.s, d);
      };
      ops.push_back(std::move(op));
      return;
    }
  }
  // tail conv: run as 4 folded sub-pixel phases on the un-upsampled input (25 instead of 64
  // taps per 2x2 outputs; see ops.h tail_fold_weights).  The folded weights and the folded
  // weight-gradient scratch live next to the dgrad operands and follow arena.version.
  const bool folded = kind == CK_TAIL_UP;
  size_t fold_off = 0, dfold_off = 0;
  if (folded) {
    const size_t fe = tail_fold_offset(arena.params[wi].ws, 4);
    fold_off = reserve_dg(self, fe);
    dfold_off = reserve_dg(self, fe);
  }
  // Stride-1 convs with MFMA-friendly channel counts run as Winograd F(m x m, r x r): transform,
  // (m+r-1)^2 batched GEMMs, inverse transform (wino.hip).  3x3: F(4x4,3x3) when H and W are
  // multiples of 4 (4x fewer multiplies), else F(2x2,3x3) (2.25x); PatchGAN's k4 s1: F(3x3,4x4) (4x).
  const bool wino_on = !(getenv("SWN_WINOGRAD") && atoi(getenv("SWN_WINOGRAD")) == 0);
  // ... when the channel counts are large enough for the GEMMs (K = Cin each) to run at MFMA speed
  // and to amortise the HBM-bound transforms: measured on VGG16 (texture C3), the 6-point forms win
  // from 64 channels up, F(2x2,3x3) (4x instead of 2.25x transform data per input) only from 256
  const int wino_minc_env = getenv("SWN_WINO_MINC") ? atoi(getenv("SWN_WINO_MINC")) : 0;
  const int wino_force_m = getenv("SWN_WINO_M") ? atoi(getenv("SWN_WINO_M")) : 0;
  const bool wino_k4 = true;
  const bool is_k3 = kind == CK_K3S1_REFLECT || kind == CK_K3S1_ZERO;
  const bool is_k4 = kind == CK_K4S1 && wino_k4;
  const int wr = is_k4 ? 4 : 3;
  const int wm = is_k4 ? 3 : ((wino_force_m != 2 && x.v.H % 4 == 0 && x.v.W % 4 == 0) ? 4 : 2);
  const int wino_minc = wino_minc_env > 0 ? wino_minc_env : (wm == 2 ? 256 : 64);
  const bool wino = wino_on && Cip % 32 == 0 && Co % 32 == 0 && Cip >= wino_minc && Co >= wino_minc && x.v.H >= 4 &&
                    x.v.W >= 4 && ((is_k3 && x.v.H % 2 == 0 && x.v.W % 2 == 0) || is_k4);
  const int wP = (wm + wr - 1) * (wm + wr - 1);
  const int wN = x.v.N, wTh = ceil_div(y.v.H, wm), wTw = ceil_div(y.v.W, wm);
  const size_t wT = (size_t)wN * wTh * wTw;
  // dgrad of the REFLECT-padded convs (the resblocks) = the adjoint of the forward Winograd pipeline, in the forward tiling:
  // dV = (A dY A^T) U^T, dx = adjoint input transform (ops.h wino_input_adjoint) -- 16 tiles per 16x16 map where the
  // transposed-conv form below walks the 18x18 padded gradient grid in 25.  The zero-padded convs (VGG16, PatchGAN's k4 s1)
  // keep the transposed stride-1 conv over dY (pad r-1-p): same tile count either way, and it is the better-conditioned form
  // (its large-entry matrices B^T / A^T act on data, not on the GEMM's output).  SWN_WINO_ADJOINT=0/2: never / always adjoint.
  const int wadj_env = 1;
  const bool wadj = wadj_env == 2 || (wadj_env == 1 && kind == CK_K3S1_REFLECT);
  const int wpad2 = kind == CK_K3S1_ZERO ? 1 : 2;
  const int wTh2 = ceil_div(x.v.H + (kind == CK_K3S1_REFLECT ? 2 : 0), wm);
  const int wTw2 = ceil_div(x.v.W + (kind == CK_K3S1_REFLECT ? 2 : 0), wm);
  const size_t wT2 = wadj ? 0 : (size_t)wN * wTh2 * wTw2;
  size_t uf_off = 0, ub_off = 0;
  float* keepV = nullptr;         // V = B^T d B of the forward input, reused by the weight gradient
  if (wino && keep_wino_inputs && y.has_grad) keepV = static_cast<float*>(ctx.alloc((size_t)wP * wT * Cip * sizeof(float)));
  // 6-point forms: the transformed filters go straight into the pre-cut operand layout of the ring kernel (no fp32 U)
  const int pcw = (wino && wm != 2) ? wino_precut_tile(Cip, Cop) : 0;
  const size_t pcw_bs = pcw ? conv_precut_elems(Cip, Cop, pcw) : 0;
  size_t pcw_off = 0, pcwt_off = 0;
  int pcwt = 0;
  size_t pcwt_bs = 0;
  // amax slots of the Winograd-domain operands: V (forward planes), dM (transformed dY), dX (the padded dY planes of the
  // transposed-conv form of the input gradient); the 6-point forms only (F(2,3) planes feed the fp32-operand kernels)
  const bool wslots = wino && wm != 2;
  const size_t slV = wslots ? reserve_slot() : 0, slD = wslots ? reserve_slot() : 0, slX = wslots ? reserve_slot() : 0;
  // pair-form planes (ops.h wino_input_transform): V feeds the forward GEMM and the weight gradient, dM the weight gradient and
  // the adjoint-form input gradient, dX (transposed-conv form of the input gradient) its one GEMM
  const bool wgrad_pairs = wslots && conv_wgrad_takes_pairs(wT, Cip, Cop);
  const bool pairV_ok = wslots && wino_fwd_takes_pairs(Cip, Cop) && (!y.has_grad || wgrad_pairs);
  const bool pairD_ok = wslots && wgrad_pairs && (!(wadj && x.has_grad && y.has_grad) || wino_fwd_takes_pairs(Cop, Cip));
  const bool pairX_ok = wslots && wino_fwd_takes_pairs(Cop, Cip);
  const size_t kV = wslots ? reserve_k() : 0, kD = wslots ? reserve_k() : 0, kX = wslots ? reserve_k() : 0;
  float* keepdM = nullptr;        // dM = A dY A^T, shared by the weight gradient (side stream) and the adjoint-form input gradient
  if (wino && wadj && y.has_grad && x.has_grad && share_dy()) keepdM = static_cast<float*>(ctx.alloc((size_t)wP * wT * Cop * sizeof(float)));
  if (wino) {
    if (pcw) pcw_off = reserve_dgp(pcw_bs * wP);
    else uf_off = reserve_dg(self, (size_t)wP * Cip * Cop);
    wsV_need = std::max(wsV_need, std::max((size_t)wP * wT * std::max(Cip, Cop), (size_t)wP * wT2 * Cop));
    wsM_need = std::max(wsM_need, std::max((size_t)wP * wT * std::max(Cip, Cop), (size_t)wP * wT2 * Cip));
    wsU_need = std::max(wsU_need, (size_t)wP * Cip * Cop);
  }
  auto plane_view = [](float* p, size_t T, int C) {
    TView v; v.p = p; v.N = 1; v.H = 1; v.W = (int)T; v.C = C; v.cs = C; return v;   // T x C matrix
  };
  // weight panel pre-cut for the ring kernel (ops.h conv_precut): plain forward convs multiply by the arena weights themselves
  const int Kf = KH * KH * Cip;
  const int pc_f = (!wino && !folded) ? conv_precut_tile(Cip, arena.params[wi].ws.Npad) : 0;
  const size_t pcf_off = pc_f ? reserve_dgp(conv_precut_elems(Kf, arena.params[wi].ws.Npad, pc_f)) : 0;
  // amax slot of the output buffer: a direct conv with a fused activation feeds the next GEMM without a normalisation in
  // between (UNetDown without InstanceNorm, PatchGAN model.0), so it folds its output's amax (in the ring kernel's epilogue, else
  // by a pass: ops.h ConvFwdArgs::y_amax); everything else here is followed by a norm_act, which folds for its own output
  // (round 5: the 6-point Winograd output transform folds too -- VGG16's conv + ReLU layers feed the next conv directly)
  const bool y_folds = !folded && actf != ACT_NONE && actf != ACT_TANH && (!wino || wm != 2);
  const size_t ySlot = note_writer(y.vbase, y_folds);
  const float* xbase = x.vbase;
  // Conv + InstanceNorm fusion: a plain direct conv above the register-resident InstanceNorm's 1024 pixels offers the
  // statistics' partial sums of its raw output from its epilogue (taken up by Net::norm_act if one normalises this buffer)
  double* stat_partial = nullptr;
  std::shared_ptr<bool> stat_use;
  {
    const int HoWo = yv.H * yv.W;
    const bool stats_on = !(getenv("SWN_CONV_STATS") && atoi(getenv("SWN_CONV_STATS")) == 0);       // (A/B: read when a model is built)
    const int chunk = (stats_on && !wino && !folded && pc_f == 128 && actf == ACT_NONE && HoWo > 1024 && yv.cs == yv.C && yv.p == y.vbase)
                          ? conv_fwd_stat_chunk(Cip, arena.params[wi].ws.Npad, HoWo, yv.N, Kf) : 0;
    if (chunk) {
      const int chunks = HoWo / chunk;
      stat_partial = static_cast<double*>(ctx.alloc((size_t)yv.N * chunks * yv.C * 2 * sizeof(double)));
      stat_use = std::make_shared<bool>(false);
      conv_stats[y.vbase] = StatLink{stat_partial, chunks, stat_use};
    }
  }
  op->fwd = [=](Net& n) {
    const ParamDesc& wd = A->params[wi];
    ConvFwdArgs a;
    a.x = xv; a.g = gf; a.w = A->w + wd.off; a.Npad = wd.ws.Npad;
    a.x_amax = n.slot_if_complete(xbase);
    if (y_folds) a.y_amax = n.amax + ySlot;
    if (stat_use && *stat_use) a.stat_partial = stat_partial;
    if (pc_f) { n.need(self); a.wpc = n.dgp + pcf_off; a.wpc_bn = pc_f; }
    a.bias = bi >= 0 ? A->w + A->params[bi].off : nullptr;
    a.act = actf; a.y = yv; a.Cout = Co;
    if (wino) {
      n.need(self);
      float* V = keepV ? keepV : n.wsV;
      const float* xin = pairV_ok ? n.slot_if_complete(xbase) : nullptr;
      wino_input_transform(n.ctx.s, wm, wr, xv, 1, gf.pad_mode, wTh, wTw, V, wslots ? n.amax + slV : nullptr, xin, n.kscale + kV);
      ConvFwdArgs g;
      g.x = plane_view(V, wT, Cip); g.g.Ho = 1; g.g.Wo = (int)wT;
      g.w = pcw ? nullptr : n.dg + uf_off; g.Npad = Cop; g.Cout = Co;
      if (xin) g.x_pair_k = n.kscale + kV; else if (wslots) g.x_amax = n.amax + slV;
      if (pcw) { g.wpc = n.dgp + pcw_off; g.wpc_bn = pcw; g.wpc_bs = pcw_bs; }
      g.y = plane_view(n.wsM, wT, Cop);
      g.batch = wP; g.x_bs = wT * Cip; g.w_bs = (size_t)Cip * Cop; g.y_bs = wT * Cop;
      conv_fwd(n.ctx.s, g);
      wino_output_transform(n.ctx.s, wm, wr, n.wsM, Cop, wTh, wTw, a.bias, actf, yv, Co, 0, y_folds ? n.amax + ySlot : nullptr);
      return;
    }
    if (!folded) { conv_fwd(n.ctx.s, a); return; }
    n.need(self);                            // folded weights are derived operands too
    a.x_amax = nullptr;
    a.g = Gather(); a.g.KH = a.g.KW = 3; a.g.stride = 1; a.g.pad_t = a.g.pad_l = 1; a.g.Ho = xv.H; a.g.Wo = xv.W;
    a.om.ymul = a.om.xmul = 2; a.tail4 = 1;
    a.w = n.dg + fold_off;
    conv_fwd(n.ctx.s, a);
  };

  // ---- backward plan
  Var scratch;       // dR when an activation is fused into the epilogue
  // tail conv: dR lives in a 32-channel buffer (pads stay zero) so that its input gradient -- a 5x5 stride-2 conv over dR with
  // K = 25 x Cop -- meets the ring kernel's 16-channel stages (K = 800 on the bf16-split ring kernel instead of K = 500 on the
  // register-staged f32-MFMA one)
  const int CopD = (kind == CK_TAIL_UP && actf != ACT_NONE && Cop <= 32 && x.has_grad && conv_precut_tile(32, Cip) == 192) ? 32 : Cop;
  if (y.has_grad && actf != ACT_NONE) scratch = alloc_var(yv.N, yv.H, yv.W, CopD, false);
  Var dxpad;
  int dg_mode = 0;
  Gather gd;         // dgrad gather over dY
  int dHo = x.v.H, dWo = x.v.W;
  switch (kind) {
    case CK_K4S2: dg_mode = 0; break;
    case CK_K3S1_REFLECT:
      dg_mode = 1; gd.KH = gd.KW = 3; gd.stride = 1; gd.pad_t = gd.pad_l = 2; dHo = x.v.H + 2; dWo = x.v.W + 2; break;
    case CK_K3S1_ZERO: dg_mode = 1; gd.KH = gd.KW = 3; gd.stride = 1; gd.pad_t = gd.pad_l = 1; break;
    case CK_K4S1: dg_mode = 1; gd.KH = gd.KW = 4; gd.stride = 1; gd.pad_t = gd.pad_l = 2; break;
    case CK_K1S1: dg_mode = 1; gd.KH = gd.KW = 1; gd.stride = 1; gd.pad_t = gd.pad_l = 0; break;
    case CK_TAIL_UP: dg_mode = 3; gd.KH = gd.KW = 5; gd.stride = 2; gd.pad_t = gd.pad_l = 1; break;
  }
  gd.Ho = dHo; gd.Wo = dWo;
  const bool want_dx = x.has_grad && y.has_grad;
  size_t dg_off = 0, pcd_off = 0;
  int pc_d = 0, dpanels = 1, dKp = 0;
  // the forward operand's weight amax, handed from the pc_f re-pack to the input-gradient operand's (ops.h conv_precut amax_io)
  auto fwd_wmax = std::make_shared<const float*>(nullptr);
  // dgrad output channels = input buffer channels -- or, for a layer that reads a network input of which only the leading
  // channels are anyone's output (the generator's image inside the conditional discriminator's input), just those
  const bool narrow_dx = dgrad_C > 0 && dgrad_C % 4 == 0 && dgrad_C < Cip && !wino && (kind == CK_K4S2 || kind == CK_K3S1_ZERO);
  const int Ndg = narrow_dx ? dgrad_C : Cip;
  const Var xg_target = narrow_dx ? x.slice(0, Ndg) : x;
  if (want_dx) {
    if (wino) {
      pcwt = wm != 2 ? wino_precut_tile(Cop, Cip) : 0;
      pcwt_bs = pcwt ? conv_precut_elems(Cop, Cip, pcwt) : 0;
      if (pcwt) pcwt_off = reserve_dgp(pcwt_bs * wP);
      else ub_off = reserve_dg(self, (size_t)wP * Cop * Cip);
    }
    else dg_off = reserve_dg(self, dgrad_elems(arena.params[wi].ws, dg_mode, CopD, Ndg));
    if (kind == CK_K3S1_REFLECT && !(wino && wadj)) dxpad = alloc_var(x.v.N, dHo, dWo, Cip, false);
    op->grad_targets.push_back(xg_target);
    if (!wino) {
      // K4S2: four phase panels of 2x2 taps; stride-1: one panel of KH x KW taps over dY (Cop channels)
      dpanels = kind == CK_K4S2 ? 4 : 1;
      dKp = (kind == CK_K4S2 ? 4 : gd.KH * gd.KW) * CopD;
      pc_d = conv_precut_tile(CopD, Ndg);
      if (pc_d) pcd_off = reserve_dgp(conv_precut_elems(dKp, Ndg, pc_d) * dpanels);
      const int pcd = pc_d, dK = dKp, dP = dpanels; const size_t pcdo = pcd_off;
      op->repack = [=](Net& n) {
        const ParamDesc& wd = A->params[wi];
        repack_dgrad(n.ctx.s, wd.ws, dg_mode, CopD, Ndg, A->w + wd.off, n.dg + dg_off);
        // (the re-pack is a permutation of the parameter -- of its leading input channels if the gradient is narrow: the partial
        // maxima the forward operand's pre-cut just took, if it ran in front of us, bound it)
        const float* wmax = *fwd_wmax;
        if (pcd) conv_precut(n.ctx.s, n.dg + dg_off, dK, Ndg, pcd, dP, (size_t)dK * Ndg, n.dgp + pcdo, &wmax);
        *fwd_wmax = nullptr;
      };
    }
  }
  if (wino) {
    const bool wdx = want_dx;
    const int pcwt_c = pcwt; const size_t pcwt_o = pcwt_off, pcwt_b = pcwt_bs;
    op->repack = [=](Net& n) {
      const ParamDesc& wd = A->params[wi];
      const float* wmax = nullptr;       // both directions transform the same packed parameter: one amax pass
      if (pcw) wino_filter_transform_pc(n.ctx.s, wm, wr, wd.ws, 0, A->w + wd.off, pcw, n.dgp + pcw_off, pcw_bs, &wmax);
      else wino_filter_transform(n.ctx.s, wm, wr, wd.ws, 0, A->w + wd.off, n.dg + uf_off);
      if (wdx) {
        if (pcwt_c) wino_filter_transform_pc(n.ctx.s, wm, wr, wd.ws, wadj ? 2 : 1, A->w + wd.off, pcwt_c, n.dgp + pcwt_o, pcwt_b, &wmax);
        else wino_filter_transform(n.ctx.s, wm, wr, wd.ws, wadj ? 2 : 1, A->w + wd.off, n.dg + ub_off);
      }
    };
  }
  if (folded) {
    auto prev = op->repack;
    op->repack = [=](Net& n) {
      if (prev) prev(n);
      const ParamDesc& wd = A->params[wi];
      tail_fold_weights(n.ctx.s, wd.ws, A->w + wd.off, n.dg + fold_off);
    };
  }
  if (pc_f) {
    auto prev = op->repack;
    op->repack = [=](Net& n) {
      const ParamDesc& wd = A->params[wi];
      const float* wmax = nullptr;
      conv_precut(n.ctx.s, A->w + wd.off, Kf, wd.ws.Npad, pc_f, 1, 0, n.dgp + pcf_off, &wmax);     // forward operand first
      // handed to the input-gradient operand's pre-cut -- only where that is the whole of `prev` (a folded / Winograd layer runs
      // other amax passes in between, which reuse the scratch the partials live in)
      *fwd_wmax = (!wino && !folded) ? wmax : nullptr;
      if (prev) prev(n);
      *fwd_wmax = nullptr;
    };
  }
  const int pcd_k = pc_d; const size_t pcd_o = pcd_off, pcd_bs = pc_d ? conv_precut_elems(dKp, Ndg, pc_d) : 0;
  const TView ygv = y.g, xgv = narrow_dx ? x.g.slice(0, Ndg) : x.g, scr = CopD != Cop ? scratch.v.slice(0, Cop) : scratch.v, scr_full = scratch.v, dxp = dxpad.v;
  const bool has_ygrad = y.has_grad;
  // dY as a GEMM operand: dR from act_bwd (which folds its amax into the scratch buffer's slot) or y.g itself, whose slot --
  // if it has one -- the norm_act behind this conv fills in its backward
  const size_t scrSlot = (y.has_grad && actf != ACT_NONE) ? note_writer(scratch.v.p, true) : 0;
  const float* ygbase = y.gbase;
  op->bwd = [=](Net& n, Op& me, bool wgrad, bool igrad) {
    if (!has_ygrad) return;
    TView dY = ygv;
    const float* dy_slot = nullptr;
    if (actf != ACT_NONE) { act_bwd(n.ctx.s, ygv, yv, scr, actf, 0, n.amax + scrSlot); dY = scr; dy_slot = n.amax + scrSlot; }
    else dy_slot = n.slot_if_complete(ygbase);
    const ParamDesc& wd = A->params[wi];
    const bool dx_now = want_dx && !(me.reads_net_input && !igrad);
    // adjoint-form layers: the weight gradient and the input gradient multiply by the same dM planes -- transformed once, on the
    // main stream, into the layer's own buffer (the side stream reads it while the main stream moves on)
    const bool shared = keepdM && wino && wadj && wgrad && dx_now;
    const float* xin = pairV_ok ? n.slot_if_complete(xbase) : nullptr;
    const float* din = pairD_ok ? dy_slot : nullptr;            // amax of dY bounds its planes
    if (shared) wino_dy_transform(n.ctx.s, wm, wr, dY, wTh, wTw, keepdM, wslots ? n.amax + slD : nullptr, din, n.kscale + kD);
    if (wgrad) {
      Stream& sw = n.wgrad_stream();          // dY is final: the weight-gradient work may run beside the dgrad chain
      ConvWgradArgs wa;
      wa.x = xv; wa.g = gf; wa.dy = dY; wa.dw = A->g + wd.off; wa.Npad = wd.ws.Npad; wa.Cout = Co;
      wa.x_amax = n.slot_if_complete(xbase); wa.dy_amax = dy_slot;
      if (wino) {
        // dU[t] = V[t]^T dM[t] (wP batched reductions over the tiles), then dW = G^T dU G
        float* V = keepV ? keepV : n.wsV;
        float* dM = shared ? keepdM : n.wgrad_planes(sw);
        if (!keepV) wino_input_transform(sw, wm, wr, xv, 1, gf.pad_mode, wTh, wTw, V, wslots ? n.amax + slV : nullptr, xin, n.kscale + kV);
        if (!shared) wino_dy_transform(sw, wm, wr, dY, wTh, wTw, dM, wslots ? n.amax + slD : nullptr, din, n.kscale + kD);
        ConvWgradArgs g;
        g.x = plane_view(V, wT, Cip); g.g.Ho = 1; g.g.Wo = (int)wT;
        g.dy = plane_view(dM, wT, Cop);
        g.dw = n.wsU; g.Npad = Cop; g.Cout = Co;
        g.batch = wP; g.x_bs = wT * Cip; g.dy_bs = wT * Cop; g.dw_bs = (size_t)Cip * Cop;
        if (xin) g.x_pair_k = n.kscale + kV; else if (wslots) g.x_amax = n.amax + slV;
        if (din) g.dy_pair_k = n.kscale + kD; else if (wslots) g.dy_amax = n.amax + slD;
        conv_wgrad(sw, g);
        wino_filter_grad(sw, wm, wr, wd.ws, n.wsU, A->g + wd.off);
      } else if (!folded) {
        conv_wgrad(sw, wa);
      } else {
        wa.x_amax = wa.dy_amax = nullptr;
        wa.g = Gather(); wa.g.KH = wa.g.KW = 3; wa.g.stride = 1; wa.g.pad_t = wa.g.pad_l = 1; wa.g.Ho = xv.H; wa.g.Wo = xv.W;
        wa.om.ymul = wa.om.xmul = 2; wa.tail4 = 1;
        wa.dw = n.dg + dfold_off;
        conv_wgrad(sw, wa);
        tail_unfold_wgrad(sw, wd.ws, n.dg + dfold_off, A->g + wd.off);
      }
      // bias gradient: beside the weight gradient on the second stream -- unless this layer forms no input gradient (the first layer of
      // a pass: PatchGAN's model.0 in backward_D, whose weight gradient is the step's exposed tail), where the main stream has nothing
      // else to do and the column sums run there, beside the weight gradient instead of behind it (round 6; same kernel, bit-identical)
      static const bool bias_on_main = !(getenv("SWN_BIAS_MAIN") && atoi(getenv("SWN_BIAS_MAIN")) == 0);
      if (bi >= 0) n.bias_grad_of((!dx_now && bias_on_main) ? n.ctx.s : sw, dY, A->g + A->params[bi].off);
    }
    if (!dx_now) return;
    const int accf = me.acc.empty() ? 0 : me.acc[0];
    if (wino && !wadj) {
      // input gradient = the transposed stride-1 conv over dY (flipped, channel-transposed filter)
      const float* xdin = pairX_ok ? dy_slot : nullptr;
      wino_input_transform(n.ctx.s, wm, wr, dY, wpad2, PAD_ZERO, wTh2, wTw2, n.wsV, wslots ? n.amax + slX : nullptr, xdin, n.kscale + kX);
      ConvFwdArgs g;
      g.x = plane_view(n.wsV, wT2, Cop); g.g.Ho = 1; g.g.Wo = (int)wT2;
      g.w = pcwt ? nullptr : n.dg + ub_off; g.Npad = Cip; g.Cout = Cip;
      if (xdin) g.x_pair_k = n.kscale + kX; else if (wslots) g.x_amax = n.amax + slX;
      if (pcwt) { g.wpc = n.dgp + pcwt_off; g.wpc_bn = pcwt; g.wpc_bs = pcwt_bs; }
      g.y = plane_view(n.wsM, wT2, Cip);
      g.batch = wP; g.x_bs = wT2 * Cop; g.w_bs = (size_t)Cop * Cip; g.y_bs = wT2 * Cip;
      conv_fwd(n.ctx.s, g);
      if (kind == CK_K3S1_REFLECT) {
        wino_output_transform(n.ctx.s, wm, wr, n.wsM, Cip, wTh2, wTw2, nullptr, ACT_NONE, dxp, Cip, 0);
        reflect_fold(n.ctx.s, dxp, xgv, accf);
      } else {
        wino_output_transform(n.ctx.s, wm, wr, n.wsM, Cip, wTh2, wTw2, nullptr, ACT_NONE, xgv, Cip, accf);
      }
      return;
    }
    if (wino) {
      // input gradient in the forward tiling: dM = A dY A^T, dV = dM U^T (U with the channel axes swapped), then the adjoint
      // of the input transform scatters the patches BT^T dV BT back through the forward gather (padding rule included)
      float* dMx = shared ? keepdM : n.wsV;
      if (!shared) wino_dy_transform(n.ctx.s, wm, wr, dY, wTh, wTw, dMx, wslots ? n.amax + slD : nullptr, din, n.kscale + kD);
      ConvFwdArgs g;
      g.x = plane_view(dMx, wT, Cop); g.g.Ho = 1; g.g.Wo = (int)wT;
      g.w = pcwt ? nullptr : n.dg + ub_off; g.Npad = Cip; g.Cout = Cip;
      if (din) g.x_pair_k = n.kscale + kD; else if (wslots) g.x_amax = n.amax + slD;
      if (pcwt) { g.wpc = n.dgp + pcwt_off; g.wpc_bn = pcwt; g.wpc_bs = pcwt_bs; }
      g.y = plane_view(n.wsM, wT, Cip);
      g.batch = wP; g.x_bs = wT * Cop; g.w_bs = (size_t)Cop * Cip; g.y_bs = wT * Cip;
      conv_fwd(n.ctx.s, g);
      wino_input_adjoint(n.ctx.s, wm, wr, n.wsM, Cip, 1, gf.pad_mode, wTh, wTw, xgv, accf);
      return;
    }
    if (kind == CK_K4S2) {
      // four sub-pixel phases of the transposed conv (2x2 taps each) in one launch
      ConvFwdArgs d;
      d.x = dY; d.g.KH = d.g.KW = 2; d.g.stride = 1; d.g.pad_t = 1; d.g.pad_l = 1;
      d.g.Ho = dY.H; d.g.Wo = dY.W;
      d.w = n.dg + dg_off; d.w_bs = (size_t)4 * Cop * Ndg; d.Npad = Ndg;
      d.y = xgv; d.om.ymul = 2; d.om.xmul = 2; d.phases = 4; d.Cout = Ndg; d.accumulate = accf;
      d.x_amax = dy_slot;
      if (pcd_k) { d.wpc = n.dgp + pcd_o; d.wpc_bn = pcd_k; d.wpc_bs = pcd_bs; }
      conv_fwd(n.ctx.s, d);
    } else {
      ConvFwdArgs d;
      d.x = CopD != Cop ? scr_full : dY; d.g = gd; d.w = n.dg + dg_off; d.Npad = Ndg; d.Cout = Ndg;
      d.x_amax = dy_slot;
      if (pcd_k) { d.wpc = n.dgp + pcd_o; d.wpc_bn = pcd_k; d.wpc_bs = pcd_bs; }
      if (kind == CK_K3S1_REFLECT) {
        d.y = dxp; d.accumulate = 0;
        conv_fwd(n.ctx.s, d);
        reflect_fold(n.ctx.s, dxp, xgv, accf);
      } else {
        d.y = xgv; d.accumulate = accf;
        conv_fwd(n.ctx.s, d);
      }
    }
  };
  ops.push_back(std::move(op));
}

// ---- ConvTranspose2d k4 s2 p1 (modules/layers.py:31, pix2pix_modules.py:226-246) ------
void Net::convT(const std::string& name, const Var& x, const Var& y, int Co, bool bias) {
  if (y.v.H != x.v.H * 2 || y.v.W != x.v.W * 2) throw Error(1, "convT " + name + ": output view has the wrong size");
  const int Cip = x.v.C, Cop = round_up(Co, 4);
  if (y.v.C != Cop) throw Error(1, "convT " + name + ": output view must have round_up(Co,4) channels");
  const int wi = arena.add_weight(name + ".weight", WK_CONVT, Co, Cip, 4, 4, Cip, nullptr);
  const int bi = bias ? arena.add_bias(name + ".bias", Co) : -1;
  auto op = std::make_unique<Op>();
  Op* self = op.get();
  op->label = name;
  op->param_off = arena.params[wi].off;
  ParamArena* A = &arena;
  const TView xv = x.v, yv = y.v, ygv = y.g, xgv = x.g;
  note_writer(y.vbase, false);        // (raw output: always followed by an InstanceNorm, never a GEMM operand itself)
  const float* xbase = x.vbase; const float* ygbase = y.gbase;
  // enough channels: strided Winograd F(4x4,2x2) -- the transposed conv is the ADJOINT of a k4 s2 conv fine -> coarse, so its
  // forward is the coarse -> fine pipeline (dM = A x A^T, dV = dM U^T, adjoint polyphase transform) and its input gradient the
  // fine -> coarse one
  if (s2_wino && s2_wino_wanted(Cop, Cip, x.v.H, x.v.W) && Co % 4 == 0) {
    const int sP = 25, sTh = ceil_div(x.v.H, 4), sTw = ceil_div(x.v.W, 4), CV = 4 * Cop;
    const size_t sT = (size_t)x.v.N * sTh * sTw;
    const bool want_dx = x.has_grad && y.has_grad;
    const size_t ut_off = reserve_dg(self, (size_t)sP * Cip * CV);                      // U^T[25][Cip][4 Cop]   (forward)
    const size_t uf_off = want_dx ? reserve_dg(self, (size_t)sP * CV * Cip) : 0;        // U  [25][4 Cop][Cip]   (input gradient)
    float* keepM = (keep_wino_inputs && y.has_grad) ? static_cast<float*>(ctx.alloc(sP * sT * Cip * sizeof(float))) : nullptr;
    wsV_need = std::max(wsV_need, sP * sT * (size_t)std::max(CV, Cip));
    wsM_need = std::max(wsM_need, sP * sT * (size_t)std::max(CV, Cip));
    wsU_need = std::max(wsU_need, (size_t)sP * CV * Cip);
    const int pct = conv_precut_tile(Cip, CV), pcf = want_dx ? conv_precut_tile(CV, Cip) : 0;
    const size_t pct_bs = pct ? conv_precut_elems(Cip, CV, pct) : 0, pcf_bs = pcf ? conv_precut_elems(CV, Cip, pcf) : 0;
    const size_t pct_off = pct ? reserve_dgp(pct_bs * sP) : 0, pcf_off = pcf ? reserve_dgp(pcf_bs * sP) : 0;
    const size_t slM = reserve_slot(), slV = reserve_slot();     // amax of dM (planes of the coarse input) and of V (planes of dY, fine)
    const bool wg_pairs = conv_wgrad_takes_pairs(sT, CV, Cip);
    const bool pairM_ok = conv_fwd_takes_pairs(Cip, CV) && (!y.has_grad || wg_pairs);
    const bool pairV_ok = wg_pairs && (!(x.has_grad && y.has_grad) || conv_fwd_takes_pairs(CV, Cip));
    const size_t kM = reserve_k(), kVg = reserve_k();
    // V = polyphase transform of dY serves the weight gradient (side stream) and the input gradient (main stream): one buffer per layer
    float* keepVg = (y.has_grad && want_dx && share_dy()) ? static_cast<float*>(ctx.alloc(sP * sT * CV * sizeof(float))) : nullptr;
    op->repack = [=](Net& n) {
      const ParamDesc& wd = A->params[wi];
      wino_s2_filter_transform(n.ctx.s, wd.ws, 1, A->w + wd.off, n.dg + ut_off);
      const float* wmax = nullptr;
      if (pct) conv_precut(n.ctx.s, n.dg + ut_off, Cip, CV, pct, sP, (size_t)Cip * CV, n.dgp + pct_off, &wmax);
      if (want_dx) {
        wino_s2_filter_transform(n.ctx.s, wd.ws, 0, A->w + wd.off, n.dg + uf_off);
        if (pcf) conv_precut(n.ctx.s, n.dg + uf_off, CV, Cip, pcf, sP, (size_t)CV * Cip, n.dgp + pcf_off, &wmax);
      }
    };
    op->fwd = [=](Net& n) {
      n.need(self);
      const ParamDesc& wd = A->params[wi];
      (void)wd;
      float* dM = keepM ? keepM : n.wsV;
      const float* xin = pairM_ok ? n.slot_if_complete(xbase) : nullptr;
      wino_dy_transform(n.ctx.s, 4, 2, xv, sTh, sTw, dM, n.amax + slM, xin, n.kscale + kM);
      ConvFwdArgs g;
      g.x = plane_mat(dM, sT, Cip); g.g.Ho = 1; g.g.Wo = (int)sT;
      g.w = n.dg + ut_off; g.Npad = CV; g.Cout = CV;
      g.y = plane_mat(n.wsM, sT, CV);
      g.batch = sP; g.x_bs = sT * Cip; g.w_bs = (size_t)Cip * CV; g.y_bs = sT * CV;
      if (xin) g.x_pair_k = n.kscale + kM; else g.x_amax = n.amax + slM;
      if (pct) { g.wpc = n.dgp + pct_off; g.wpc_bn = pct; g.wpc_bs = pct_bs; }
      conv_fwd(n.ctx.s, g);
      wino_s2_input_adjoint(n.ctx.s, n.wsM, Cop, sTh, sTw, yv, bi >= 0 ? A->w + A->params[bi].off : nullptr, 0);
    };
    if (want_dx) op->grad_targets.push_back(x);
    const bool has_ygrad = y.has_grad;
    op->bwd = [=](Net& n, Op& me, bool wgrad, bool igrad) {
      if (!has_ygrad) return;
      const ParamDesc& wd = A->params[wi];
      const bool dx_now = want_dx && !(me.reads_net_input && !igrad);
      const bool shared = keepVg && wgrad && dx_now;
      const float* xin = pairM_ok ? n.slot_if_complete(xbase) : nullptr;
      const float* din = pairV_ok ? n.slot_if_complete(ygbase) : nullptr;
      if (shared) wino_s2_input_transform(n.ctx.s, ygv, sTh, sTw, keepVg, n.amax + slV, din, n.kscale + kVg);
      if (wgrad) {
        // dU[25][4 Cop][Cip] = V(dY fine)^T dM(x coarse)
        Stream& sw = n.wgrad_stream();
        float* V = shared ? keepVg : n.wgrad_planes(sw);
        float* dM = keepM ? keepM : n.wsV;
        if (!shared) wino_s2_input_transform(sw, ygv, sTh, sTw, V, n.amax + slV, din, n.kscale + kVg);
        if (!keepM) wino_dy_transform(sw, 4, 2, xv, sTh, sTw, dM, n.amax + slM, xin, n.kscale + kM);
        ConvWgradArgs g;
        g.x = plane_mat(V, sT, CV); g.g.Ho = 1; g.g.Wo = (int)sT;
        g.dy = plane_mat(dM, sT, Cip);
        g.dw = n.wsU; g.Npad = Cip; g.Cout = Cip;
        g.batch = sP; g.x_bs = sT * CV; g.dy_bs = sT * Cip; g.dw_bs = (size_t)CV * Cip;
        if (din) g.x_pair_k = n.kscale + kVg; else g.x_amax = n.amax + slV;
        if (xin) g.dy_pair_k = n.kscale + kM; else g.dy_amax = n.amax + slM;
        conv_wgrad(sw, g);
        wino_s2_filter_grad(sw, wd.ws, n.wsU, A->g + wd.off);
        if (bi >= 0) n.bias_grad_of(sw, ygv, A->g + A->params[bi].off);
      }
      if (!dx_now) return;
      float* Vx = shared ? keepVg : n.wsV;
      if (!shared) wino_s2_input_transform(n.ctx.s, ygv, sTh, sTw, Vx, n.amax + slV, din, n.kscale + kVg);
      ConvFwdArgs g;
      g.x = plane_mat(Vx, sT, CV); g.g.Ho = 1; g.g.Wo = (int)sT;
      g.w = n.dg + uf_off; g.Npad = Cip; g.Cout = Cip;
      g.y = plane_mat(n.wsM, sT, Cip);
      g.batch = sP; g.x_bs = sT * CV; g.w_bs = (size_t)CV * Cip; g.y_bs = sT * Cip;
      if (din) g.x_pair_k = n.kscale + kVg; else g.x_amax = n.amax + slV;
      if (pcf) { g.wpc = n.dgp + pcf_off; g.wpc_bn = pcf; g.wpc_bs = pcf_bs; }
      conv_fwd(n.ctx.s, g);
      wino_output_transform(n.ctx.s, 4, 2, n.wsM, Cip, sTh, sTw, nullptr, ACT_NONE, xgv, Cip, me.acc.empty() ? 0 : me.acc[0]);
    };
    ops.push_back(std::move(op));
    return;
  }
  const size_t phase_elems = (size_t)4 * Cip * round_up(Co, 4);
  // pre-cut panels for the ring kernel: the four forward phase panels (arena layout) and the k4 s2 input-gradient operand
  const int pc_f = conv_precut_tile(Cip, Cop);
  const size_t pcf_bs = pc_f ? conv_precut_elems(4 * Cip, Cop, pc_f) : 0;
  const size_t pcf_off = pc_f ? reserve_dgp(pcf_bs * 4) : 0;
  op->fwd = [=](Net& n) {
    const ParamDesc& wd = A->params[wi];
    ConvFwdArgs f;                          // 4 sub-pixel phases (2x2 taps each), one launch
    if (pc_f) { n.need(self); f.wpc = n.dgp + pcf_off; f.wpc_bn = pc_f; f.wpc_bs = pcf_bs; }
    f.x = xv; f.g.KH = f.g.KW = 2; f.g.stride = 1; f.g.pad_t = 1; f.g.pad_l = 1; f.g.Ho = xv.H; f.g.Wo = xv.W;
    f.w = A->w + wd.off; f.w_bs = phase_elems; f.Npad = wd.ws.Npad;
    f.x_amax = n.slot_if_complete(xbase);
    f.bias = bi >= 0 ? A->w + A->params[bi].off : nullptr;
    f.y = yv; f.om.ymul = 2; f.om.xmul = 2; f.phases = 4; f.Cout = Co;
    conv_fwd(n.ctx.s, f);
  };
  const bool want_dx = x.has_grad && y.has_grad;
  size_t dg_off = 0, pcd_off = 0;
  const int pc_d = want_dx ? conv_precut_tile(Cop, Cip) : 0;
  if (want_dx) {
    dg_off = reserve_dg(self, dgrad_elems(arena.params[wi].ws, 2, Cop, Cip));
    if (pc_d) pcd_off = reserve_dgp(conv_precut_elems(16 * Cop, Cip, pc_d));
    op->grad_targets.push_back(x);
  }
  if (want_dx || pc_f)
    op->repack = [=](Net& n) {
      const ParamDesc& wd = A->params[wi];
      const float* wmax = nullptr;       // the k4 s2 re-pack permutes the four phase panels: one amax pass
      if (pc_f) conv_precut(n.ctx.s, A->w + wd.off, 4 * Cip, Cop, pc_f, 4, phase_elems, n.dgp + pcf_off, &wmax);
      if (want_dx) {
        repack_dgrad(n.ctx.s, wd.ws, 2, Cop, Cip, A->w + wd.off, n.dg + dg_off);
        if (pc_d) conv_precut(n.ctx.s, n.dg + dg_off, 16 * Cop, Cip, pc_d, 1, 0, n.dgp + pcd_off, &wmax);
      }
    };
  const bool has_ygrad = y.has_grad;
  op->bwd = [=](Net& n, Op& me, bool wgrad, bool igrad) {
    if (!has_ygrad) return;
    const ParamDesc& wd = A->params[wi];
    if (wgrad) {
      ConvWgradArgs wa;
      wa.x = xv; wa.g.KH = wa.g.KW = 2; wa.g.stride = 1; wa.g.pad_t = 1; wa.g.pad_l = 1;
      wa.g.Ho = xv.H; wa.g.Wo = xv.W;
      wa.dy = ygv; wa.om.ymul = 2; wa.om.xmul = 2; wa.phases = 4;
      wa.dw = A->g + wd.off; wa.dw_bs = phase_elems; wa.Npad = wd.ws.Npad; wa.Cout = Co;
      wa.x_amax = n.slot_if_complete(xbase); wa.dy_amax = n.slot_if_complete(ygbase);
      Stream& sw = n.wgrad_stream();
      conv_wgrad(sw, wa);
      if (bi >= 0) n.bias_grad_of(sw, ygv, A->g + A->params[bi].off);
    }
    if (!want_dx || (me.reads_net_input && !igrad)) return;
    ConvFwdArgs d;
    d.x = ygv; d.g.KH = d.g.KW = 4; d.g.stride = 2; d.g.pad_t = d.g.pad_l = 1; d.g.Ho = xv.H; d.g.Wo = xv.W;
    d.w = n.dg + dg_off; d.Npad = Cip; d.Cout = Cip; d.y = xgv; d.accumulate = me.acc.empty() ? 0 : me.acc[0];
    d.x_amax = n.slot_if_complete(ygbase);
    if (pc_d) { d.wpc = n.dgp + pcd_off; d.wpc_bn = pc_d; }
    conv_fwd(n.ctx.s, d);
  };
  ops.push_back(std::move(op));
}

// ---- [InstanceNorm] -> act -> [dropout] (+ residual) -----------------------------------
void Net::norm_act(const Var& raw, const Var& y, bool norm, int actf, float drop_p, const Var* residual) {
  auto op = std::make_unique<Op>();
  op->label = "norm_act";
  float* stats = norm ? static_cast<float*>(ctx.alloc((size_t)raw.v.N * raw.v.C * 2 * sizeof(float))) : nullptr;
  last_stats = stats;
  const uint64_t salt = ops.size() + 1;
  if (drop_p > 0.f) drop_sites.push_back({salt, raw.v.N, raw.v.H, raw.v.W, raw.v.C, drop_p});
  if (residual && actf != ACT_NONE) throw Error(1, "norm_act: an activation in front of a residual add is not supported");
  note_act(actf, y.v);        // a dropped element reads 0: its gradient is 0 whichever side the pattern records
  const TView rv = raw.v, rg = raw.g, yv = y.v, yg = y.g;
  const size_t ySlot = note_writer(y.vbase, true);
  const size_t gSlot = (y.has_grad && raw.has_grad) ? note_writer(raw.gbase, true) : 0;
  const bool has_res = residual != nullptr;
  const Var res = has_res ? *residual : Var();
  const double* partial_in = nullptr;
  int partial_chunks = 0;
  if (norm && raw.v.H * raw.v.W > 1024 && raw.v.p == raw.vbase) {
    auto it = conv_stats.find(raw.vbase);
    if (it != conv_stats.end()) { *it->second.use = true; partial_in = it->second.partial; partial_chunks = it->second.chunks; }
  }
  op->fwd = [=](Net& n) {
    NormActArgs a;
    a.x = rv; a.y = yv; a.stats = stats; a.norm = norm; a.act = actf;
    a.partial_in = partial_in; a.partial_chunks = partial_chunks;
    a.drop_p = n.training ? drop_p : 0.f; a.seed = Net::drop_seed(n.seed, salt);
    a.seed_base = n.seed_dev; a.salt = salt;
    a.residual = has_res ? &res.v : nullptr;
    a.amax_out = n.amax + ySlot;
    norm_act_fwd(n.ctx.s, a);
  };
  if (has_res && res.has_grad && y.has_grad) op->grad_targets.push_back(res);
  const bool do_bwd = y.has_grad && raw.has_grad;
  double* colsum = nullptr;
  if (do_bwd && norm && norm_act_bwd_emits_colsum(raw.v.H * raw.v.W, raw.v.C) && raw.g.cs == raw.g.C) {
    colsum = static_cast<double*>(ctx.alloc((size_t)raw.v.N * raw.v.C * sizeof(double)));
    colsums[raw.g.p] = {colsum, raw.v.N};
  }
  op->bwd = [=](Net& n, Op& me, bool, bool) {
    if (!do_bwd) return;
    if (has_res && res.has_grad) axpy(n.ctx.s, yg, res.g, 1.f, me.acc.empty() ? 0 : me.acc[0]);
    NormActBwdArgs b;
    b.dy = yg; b.x = rv; b.stats = stats; b.dx = rg; b.norm = norm; b.act = actf; b.colsum = colsum;
    b.drop_p = n.training ? drop_p : 0.f; b.seed = Net::drop_seed(n.seed, salt);
    b.seed_base = n.seed_dev; b.salt = salt;
    b.amax_out = n.amax + gSlot;
    norm_act_bwd(n.ctx.s, b);
  };
  ops.push_back(std::move(op));
}

void Net::act(const Var& x, const Var& y, int actf) {
  auto op = std::make_unique<Op>();
  op->label = "act";
  const size_t ySlot = note_writer(y.vbase, true);        // (round 5: the element-wise kernel folds its output's amax like every other producer)
  note_act(actf, y.v);
  const TView xv = x.v, yv = y.v, xg = x.g, yg = y.g;
  op->fwd = [=](Net& n) { act_fwd(n.ctx.s, xv, yv, actf, n.amax + ySlot); };
  const bool do_bwd = x.has_grad && y.has_grad;
  if (do_bwd) op->grad_targets.push_back(x);
  op->bwd = [=](Net& n, Op& me, bool, bool) {
    if (do_bwd) act_bwd(n.ctx.s, yg, yv, xg, actf, me.acc.empty() ? 0 : me.acc[0]);
  };
  ops.push_back(std::move(op));
}

void Net::affine(const Var& x, const Var& y, float alpha, float shift) {
  auto op = std::make_unique<Op>();
  op->label = "affine";
  note_writer(y.vbase, false);        // (not a folding writer: forward_from() may skip this op -- the VGG slices are entered behind it with
                                      // a buffer the caller filled -- so its consumer takes the amax itself)
  const TView xv = x.v, yv = y.v, xg = x.g, yg = y.g;
  op->fwd = [=](Net& n) { axpy(n.ctx.s, xv, yv, alpha, 0, shift); };
  const bool do_bwd = x.has_grad && y.has_grad;
  if (do_bwd) op->grad_targets.push_back(x);
  op->bwd = [=](Net& n, Op& me, bool, bool) {
    if (do_bwd) axpy(n.ctx.s, yg, xg, alpha, me.acc.empty() ? 0 : me.acc[0], 0.f);
  };
  ops.push_back(std::move(op));
}

void Net::upsample(const Var& x, const Var& y, int f) {
  auto op = std::make_unique<Op>();
  op->label = "upsample";
  const size_t ySlot = note_writer(y.vbase, true);
  const TView xv = x.v, yv = y.v, xg = x.g, yg = y.g;
  op->fwd = [=](Net& n) { upsample_nearest_fwd(n.ctx.s, xv, yv, f, n.amax + ySlot); };
  const bool do_bwd = x.has_grad && y.has_grad;
  if (do_bwd) op->grad_targets.push_back(x);
  op->bwd = [=](Net& n, Op& me, bool, bool) {
    if (do_bwd) upsample_nearest_bwd(n.ctx.s, yg, xg, f, me.acc.empty() ? 0 : me.acc[0]);
  };
  ops.push_back(std::move(op));
}

void Net::maxpool(const Var& x, const Var& y) {
  auto op = std::make_unique<Op>();
  op->label = "maxpool";
  const size_t ySlot = note_writer(y.vbase, true);
  act_sites.push_back({2, y.v, x.v});
  const TView xv = x.v, yv = y.v, xg = x.g, yg = y.g;
  op->fwd = [=](Net& n) { maxpool2_fwd(n.ctx.s, xv, yv, n.amax + ySlot); };
  const bool do_bwd = x.has_grad && y.has_grad;
  if (do_bwd) op->grad_targets.push_back(x);
  op->bwd = [=](Net& n, Op& me, bool, bool) {
    if (!do_bwd) return;
    maxpool2_bwd(n.ctx.s, yg, xv, yv, xg, me.acc.empty() ? 0 : me.acc[0]);
  };
  ops.push_back(std::move(op));
}

void Net::custom(const std::string& label, std::function<void(Net&)> fwd,
                 std::function<void(Net&, const std::vector<int>& acc)> bwd, const std::vector<Var>& grad_targets) {
  auto op = std::make_unique<Op>();
  op->label = label;
  op->fwd = fwd ? fwd : [](Net&) {};
  op->grad_targets = grad_targets;
  op->bwd = [bwd](Net& n, Op& me, bool, bool) { if (bwd) bwd(n, me.acc); };
  ops.push_back(std::move(op));
}

// ---- accumulate planner -----------------------------------------------------------------
// Walk the tape backwards (the order backward() executes) and decide, per gradient write,
// whether it is the first producer of that (buffer, channel range) -> overwrite, or a later
// one -> accumulate.  A write that covers a partly-initialised range is a builder bug.
void Net::finalize(const std::vector<Var>& pre) {
  std::map<float*, std::vector<char>> seen;
  auto range = [&](const Var& v, int& c0, int& c1) -> std::vector<char>& {
    std::vector<char>& s = seen[v.gbase];
    if (s.empty()) s.assign(v.g.cs, 0);
    c0 = (int)((v.g.p - v.gbase) % v.g.cs);
    c1 = c0 + v.g.C;
    return s;
  };
  for (const Var& v : pre) {
    if (!v.has_grad) continue;
    int c0, c1;
    auto& s = range(v, c0, c1);
    for (int c = c0; c < c1; ++c) s[c] = 1;
  }
  for (int i = (int)ops.size() - 1; i >= 0; --i) {
    Op& op = *ops[i];
    op.acc.assign(op.grad_targets.size(), 0);
    for (size_t t = 0; t < op.grad_targets.size(); ++t) {
      const Var& v = op.grad_targets[t];
      int c0, c1;
      auto& s = range(v, c0, c1);
      int nset = 0;
      for (int c = c0; c < c1; ++c) nset += s[c];
      if (nset != 0 && nset != c1 - c0)
        throw Error(1, "accumulate planner: op '" + op.label + "' writes a partially initialised gradient range");
      op.acc[t] = nset ? 1 : 0;
      for (int c = c0; c < c1; ++c) s[c] = 1;
    }
  }
  dg = dg_n ? static_cast<float*>(ctx.alloc(dg_n * sizeof(float))) : nullptr;
  dgp = dgp_n ? static_cast<uint16_t*>(ctx.alloc(dgp_n * sizeof(uint16_t))) : nullptr;
  amax = amax_n ? static_cast<float*>(ctx.alloc(amax_n * sizeof(float))) : nullptr;
  kscale = kscale_n ? static_cast<int*>(ctx.alloc((kscale_n + 4) * sizeof(int))) : nullptr;
  if (wsM_need && ctx.has_side && keep_wino_inputs) wsM2 = static_cast<float*>(ctx.alloc(wsM_need * sizeof(float)));
  if (wsV_need) wsV = static_cast<float*>(ctx.alloc(wsV_need * sizeof(float)));
  if (wsM_need) wsM = static_cast<float*>(ctx.alloc(wsM_need * sizeof(float)));
  if (wsU_need) wsU = static_cast<float*>(ctx.alloc(wsU_need * sizeof(float)));
  finalized_ = true;
}

// SWN_SHARE_DY=0: the weight gradient transforms dY for itself on the side stream (the round-3 behaviour); read once
bool Net::share_dy() const {
  static const bool on = true;
  return on && keep_wino_inputs && ctx.has_side;
}
void Net::forward() {
  if (!finalized_) throw Error(1, "Net::forward before finalize");
  if (amax) dev_memset(ctx.s, amax, 0, amax_n * sizeof(float));       // every slot: the step's producers fold into zeros
  prefetch_dgrad();
  for (auto& op : ops) { if (route_on()) route_label(op->label.c_str(), 'f'); op->fwd(*this); }
  prefetch_finish();
}
void Net::forward_from(int op_begin) {
  if (!finalized_) throw Error(1, "Net::forward before finalize");
  // every folding writer must lie at or behind op_begin (the frozen VGG16 slices are entered at op 1, behind a non-folding affine): a
  // skipped one would leave its buffer's slot at zero while consumers trust it as complete
  for (size_t w : fold_writer_ops)
    if (w < (size_t)op_begin) throw Error(1, "Net::forward_from(" + std::to_string(op_begin) + ") skips op " + std::to_string(w) +
                                             ", a folding writer of an amax slot");
  if (amax) dev_memset(ctx.s, amax, 0, amax_n * sizeof(float));
  prefetch_dgrad();
  for (size_t i = (size_t)op_begin; i < ops.size(); ++i) { if (route_on()) route_label(ops[i]->label.c_str(), 'f'); ops[i]->fwd(*this); }
  prefetch_finish();
}
// The operand refresh on the second stream, issued PIECEWISE (round 6).  Issuing all of it at the top of the pass -- ~130 launches and
// ~60 event records on the second stream in front of the pass's first kernel -- kept the main queue idle for 2.2 ms of every step: the
// runtime lets the host run only a bounded number of commands ahead of a queue, so the host sat inside that loop until the GPU had
// worked the batch down to its last dozen launches, and only then got to enqueue the forward pass (rocprofv3 --kernel-trace of bench.py,
// of bench.py on a stream of its own and of tools/native_ab alike: profiles/refresh_issue_order_r06.txt; tools/fork_probe.hip shows the
// events themselves release with single-kernel granularity when the host is ahead).  Now the pass starts with the operands of its first
// SWN_PREFETCH_AHEAD layers in flight and every layer, once its own operands are waited for, issues those of the layer that many places
// further down: both queues are fed in the order the GPU consumes them.  SWN_PREFETCH=3: everything at the top (the old order, for the
// A/B); 0: in order on the main stream.
int Net::prefetch_mode() {
  static const int mode = getenv("SWN_PREFETCH") ? atoi(getenv("SWN_PREFETCH")) : 1;
  return mode;
}
static size_t prefetch_ahead() {
  static const int n = getenv("SWN_PREFETCH_AHEAD") ? std::max(1, atoi(getenv("SWN_PREFETCH_AHEAD"))) : 4;
  return (size_t)n;
}
void Net::prefetch_dgrad() {
  const int mode = prefetch_mode();
  if (mode == 0 || !ctx.use_side() || dg_version == arena.version || refresh_open_) return;
  if (repack_ops_.empty())
    for (size_t i = 0; i < ops.size(); ++i)
      if (ops[i]->repack) { ops[i]->repack_index = (int)repack_ops_.size(); repack_ops_.push_back(i); }
  if (repack_ops_.empty()) return;
  ctx.fork_side();                 // after the optimizer step that produced the weights and after every
                                   // main-stream reader of the previous operands
  refresh_open_ = true; refresh_pending = true; repack_next_ = 0;
  dg_version = arena.version;
  prefetch_issue(mode == 3 ? repack_ops_.size() : prefetch_ahead());
}
void Net::prefetch_issue(size_t upto) {
  upto = std::min(upto, repack_ops_.size());
  if (!refresh_open_ || repack_next_ >= upto) return;
  const std::string label_was = g_route_label; const char phase_was = g_route_phase;      // (called from inside a layer's forward)
  struct Relabel { const std::string& l; char p; ~Relabel() { g_route_label = l; g_route_phase = p; } } relabel{label_was, phase_was};
  std::swap(ctx.s, ctx.side);      // the re-pack launchers use ctx.s
  try {
    for (; repack_next_ < upto; ++repack_next_) {
      Op* op = ops[repack_ops_[repack_next_]].get();
      if (route_on()) route_label(op->label.c_str(), 'r');
      op->repack(*this);
      if (!op->ready) op->ready = event_create();
      event_record(op->ready, ctx.s);        // (ctx.s is the side stream here)
      op->ready_pending = true;
    }
  } catch (...) { std::swap(ctx.s, ctx.side); throw; }
  std::swap(ctx.s, ctx.side);
  if (repack_next_ == repack_ops_.size()) {
    if (!refresh_event) refresh_event = event_create();
    event_record(refresh_event, ctx.side);
    refresh_open_ = false;
  }
}
void Net::prefetch_finish() { if (refresh_open_) prefetch_issue(repack_ops_.size()); }
void Net::need(Op* op) {
  if (refresh_open_ && op->repack_index >= 0) prefetch_issue((size_t)op->repack_index + 1);      // (its own: normally issued layers ago)
  if (refresh_pending && op->ready_pending) {
    stream_wait_event(ctx.s, op->ready);
    op->ready_pending = false;
    if (refresh_open_) prefetch_issue((size_t)op->repack_index + 1 + prefetch_ahead());
    return;
  }
  if (!refresh_pending) refresh_dgrad();
}
void Net::refresh_dgrad() {
  if (refresh_pending) {
    prefetch_finish();
    stream_wait_event(ctx.s, refresh_event);
    refresh_pending = false;
    for (auto& op : ops) op->ready_pending = false;
  }
  if (dg_version == arena.version) return;
  for (auto& op : ops)
    if (op->repack) { if (route_on()) route_label(op->label.c_str(), 'r'); op->repack(*this); }
  dg_version = arena.version;
}
void Net::backward(bool wgrad, bool igrad) { backward_range(wgrad, igrad, 0, (int)ops.size()); }
void Net::backward_range(bool wgrad, bool igrad, int op_begin, int op_end, bool join) {
  for (int i = op_end - 1; i >= op_begin; --i) { if (route_on()) route_label(ops[i]->label.c_str(), 'b'); ops[i]->bwd(*this, *ops[i], wgrad, igrad); }
  if (join) ctx.join_side();      // weight gradients of the range are final for whatever the main stream does next
}
int Net::split_point(double frac, size_t* arena_off) const {
  const size_t want = (size_t)(frac * (double)arena.n);
  for (size_t i = 0; i < ops.size(); ++i)
    if (ops[i]->param_off != (size_t)-1 && ops[i]->param_off >= want) {
      if (arena_off) *arena_off = ops[i]->param_off;
      return (int)i;
    }
  if (arena_off) *arena_off = arena.n;
  return (int)ops.size();
}

// Bucket boundaries (fractions of the generator's gradient arena, high to low = backward order).
// WarpModule: [.70,1] = decoder + resblock convs 5-7, [.41,.70) = resblock convs 1-4, [.04,.41) = cloth_down5/6,
// cloth_up1/2, resblock conv 0, [0,.04) = body_down1-4 + cloth_down1-4.  Sized on the measured back-propagation time of the
// buckets (bench.py `dp_buckets`, 1 x MI355X, profiles/README.md): a bucket's exchange runs under the NEXT bucket's backward
// pass, so what matters is (a) every bucket but the last is followed by more backward time than its transfer takes at the
// ~170 GB/s algorithmic all-reduce rate of 8 GPUs over xGMI, and (b) the last bucket -- whose exchange and AdamW are exposed --
// is as small as (a) allows for the bucket before it: 22 MB (0.13 ms) instead of the 55 MB of round 2, with 2.5 ms of
// encoder back-propagation above the 240 MB transfer (1.4 ms) of the third bucket.  SWAPNET_GRAD_CUTS="f1,f2,..." overrides
// (descending fractions of the arena; read once).
static std::vector<double> grad_cuts() {
  static std::vector<double> cuts = [] {
    std::vector<double> c = {0.70, 0.41, 0.04};
    if (const char* e = getenv("SWAPNET_GRAD_CUTS")) {
      std::vector<double> u;
      for (const char* p = e; *p;) {
        char* end = nullptr;
        const double v = strtod(p, &end);
        if (end == p) break;
        u.push_back(v);
        p = *end == ',' ? end + 1 : end;
      }
      bool ok = !u.empty() && u.size() <= 15;
      for (size_t i = 0; i < u.size(); ++i) ok = ok && u[i] > 0.0 && u[i] < 1.0 && (i == 0 || u[i] < u[i - 1]);
      if (ok) c = u;
    }
    return c;
  }();
  return cuts;
}
int Model::backward_G_parts() const { return (int)grad_cuts().size() + 1; }

void Model::backward_G_part(float label_real, int part, size_t* ready_off, size_t* ready_count, bool join) {
  const int np = backward_G_parts();
  if (part < 0 || part >= np) throw Error(1, "backward_G_part: part out of range");
  size_t hi_off = arenaG.n, lo_off = 0;
  int hi_op = (int)G->ops.size(), lo_op = 0;
  const std::vector<double> cuts = grad_cuts();
  if (part > 0) hi_op = G->split_point(cuts[part - 1], &hi_off);
  if (part < np - 1) lo_op = G->split_point(cuts[part], &lo_off);
  if (part == 0) {
    backward_G_head(label_real);
    G->refresh_dgrad();
  }
  G->backward_range(true, false, lo_op, hi_op, join);
  if (ready_off) *ready_off = lo_off;
  if (ready_count) *ready_count = hi_off - lo_off;
}

// ---------------------------------------------------------------------------------------
void Model::optimizer_step(int net) {
  ParamArena& A = arena(net);
  A.step += 1;
  AdamWArgs a;
  a.p = A.w; a.g = A.g; a.m = A.m; a.v = A.v; a.n = A.n;
  a.lr = net == 0 ? hyper.lr : hyper.d_lr;
  a.weight_decay = net == 0 ? hyper.weight_decay : hyper.d_weight_decay;
  a.beta1 = net == 0 ? hyper.b1 : hyper.d_b1; a.beta2 = net == 0 ? hyper.b2 : hyper.d_b2; a.eps = 1e-8f; a.step = A.step;
  if (indirect) a.sched_dev = net == 0 ? reinterpret_cast<const float*>(sp_dev) + 6 : reinterpret_cast<const float*>(sp_dev) + 8;
  adamw_step(ctx->s, a);
  A.version += 1;
}

void Model::discriminate(const float* x_nchw, float* pred_nchw) {
  if (!is_train || !D2) throw Error(1, "discriminate: the model has no discriminator (created with is_train = 0)");
  if (d_cimap_.empty()) throw Error(1, "discriminate: model did not publish its conditional-input channel map");
  if (!D3_) {
    AllocScope mine(*ctx, owned_allocs);
    D3_ = std::make_unique<Net>(*ctx, arenaD);
    d3_in_ = D3_->alloc_var(B, H, W, (int)d_cimap_.size(), false);
    d3_pred_ = build_patchgan(*D3_, d3_in_, d_layers_, d_cimap_);
    D3_->finalize({});
  }
  // scatter the reference-ordered channels into the buffer order: maximal runs of consecutive channels
  const int nb = (int)d_cimap_.size();
  const size_t plane = (size_t)H * W;
  int nref = 0;
  for (int v : d_cimap_) nref += v >= 0;
  for (int b0 = 0; b0 < nb;) {
    if (d_cimap_[b0] < 0) { ++b0; continue; }
    int len = 1;
    while (b0 + len < nb && d_cimap_[b0 + len] == d_cimap_[b0] + len) ++len;
    // NCHW source with nref channels: a channel sub-range is strided per image -> copy image by image
    for (int n = 0; n < B; ++n) {
      // (pad channels of the buffer stay zero: dev_alloc zero-fills and nothing else writes them)
      nchw_to_nhwc(ctx->s, x_nchw + ((size_t)n * nref + d_cimap_[b0]) * plane, 1, len, H, W, d3_in_.batch(n, 1).v.slice(b0, len));
    }
    b0 += len;
  }
  D3_->training = false;
  D3_->refresh_dgrad();
  D3_->forward();
  nhwc_to_nchw(ctx->s, d3_pred_.v, pred_nchw, 1);
}
void Model::set_gp_random(const float* alpha_dev, const float* beta_nchw_dev) {
  if (!is_train || d_cimap_.empty()) throw Error(1, "set_gp_random: the model has no discriminator");
  if (!gp_) {
    AllocScope mine(*ctx, owned_allocs); gp_ = std::make_unique<GradPenalty>(*ctx, arenaD, B, H, W, d_layers_);
  }
  if (alpha_dev) { dev_copy(ctx->s, gp_->alpha_buffer(), alpha_dev, (size_t)B * sizeof(float)); gp_alpha_set_ = true; }
  if (beta_nchw_dev) {                       // reference channel order (B, 22, H, W) -> buffer order, pads stay 0
    const int nb = (int)d_cimap_.size();
    const size_t plane = (size_t)H * W;
    int nref = 0;
    for (int v : d_cimap_) nref += v >= 0;
    const TView bv = gp_->beta_buffer();
    for (int b0 = 0; b0 < nb;) {
      if (d_cimap_[b0] < 0) { ++b0; continue; }
      int len = 1;
      while (b0 + len < nb && d_cimap_[b0 + len] == d_cimap_[b0] + len) ++len;
      for (int n = 0; n < B; ++n) {
        TView dst = bv; dst.p = bv.p + (size_t)n * plane * bv.cs; dst.N = 1;
        nchw_to_nhwc(ctx->s, beta_nchw_dev + ((size_t)n * nref + d_cimap_[b0]) * plane, 1, len, H, W, dst.slice(b0, len));
      }
      b0 += len;
    }
    gp_beta_set_ = true;
  }
}
void Model::run_gradient_penalty(const TView& real, const TView& fake) {
  if (!gp_) {
    AllocScope mine(*ctx, owned_allocs); gp_ = std::make_unique<GradPenalty>(*ctx, arenaD, B, H, W, d_layers_);
  }
  const TView beta = gp_->beta_buffer();
  // library RNG: a function of the step seed the caller handed to forward() (torch.initial_seed(), the step counter and --
  // under data parallelism -- the rank: every rank draws its own alpha / beta) and of the optimizer step
  const uint64_t gp_seed = (G ? G->seed : 0) * 0x9E3779B97F4A7C15ull + (uint64_t)arenaD.step * 7919ull + 13ull;
  gp_->run(real, fake, hyper.gp_mode, hyper.grad_scale, hyper.lambda_gp, gp_alpha_set_ ? gp_->alpha_buffer() : nullptr,
           gp_beta_set_ ? &beta : nullptr, gp_seed, losses + L_D_GP);
  gp_alpha_set_ = gp_beta_set_ = false;
  scalar_axpby(ctx->s, losses + L_D, 1.f, losses + L_D_GP, 1.f, losses + L_D);     // loss_D += loss_D_gp (warp_model.py:136)
}
void Model::set_style_context(const float*, const float*, int, int) {
  throw Error(1, "set_style_context: only the texture model has a style term");
}
void Model::perceptual(const float*, const float*, int, float*, float, float, float*) {
  throw Error(1, "perceptual: only the texture model carries the VGG16 network (not implemented for this model)");
}

void Model::optimizer_step_range(int net, size_t off, size_t count, int first) { optimizer_step_range_on(ctx->s, net, off, count, first); }
void Model::optimizer_step_range_on(Stream& st, int net, size_t off, size_t count, int first) {
  ParamArena& A = arena(net);
  if (off % 4 || off + count > A.n) throw Error(1, "optimizer_step_range: range outside the arena / not 16-byte aligned");
  if (first) A.step += 1;
  if (count == 0) return;
  AdamWArgs a;
  a.p = A.w + off; a.g = A.g + off; a.m = A.m + off; a.v = A.v + off; a.n = (count + 3) / 4 * 4 <= A.n - off ? (count + 3) / 4 * 4 : count;
  a.lr = net == 0 ? hyper.lr : hyper.d_lr;
  a.weight_decay = net == 0 ? hyper.weight_decay : hyper.d_weight_decay;
  a.beta1 = net == 0 ? hyper.b1 : hyper.d_b1; a.beta2 = net == 0 ? hyper.b2 : hyper.d_b2; a.eps = 1e-8f; a.step = A.step;
  if (indirect) a.sched_dev = net == 0 ? reinterpret_cast<const float*>(sp_dev) + 6 : reinterpret_cast<const float*>(sp_dev) + 8;
  adamw_step(st, a);
  A.version += 1;
}

void Model::backward_G_streamed(float label_real) {
  const int np = backward_G_parts();
  const int ver = arenaG.version;
  for (int part = 0; part < np; ++part) {
    size_t off = 0, count = 0;
    backward_G_part(label_real, part, &off, &count, /*join=*/false);
    // the side stream already carries the bucket's weight gradients; the fork orders it behind the bucket's main-stream work
    // too (bias sums, the input-gradient chain that produced their operands).  Nothing later on either stream reads the
    // bucket's weights or gradients again this step: the earlier layers multiply by their own (derived) operands.
    Stream& sd = ctx->use_side() ? ctx->fork_side() : ctx->s;
    optimizer_step_range_on(sd, 0, off, count, part == 0);
  }
  arenaG.version = ver + 1;      // one optimizer step
  ctx->join_side();
}

// BaseGAN.optimize_parameters as a recorded launch sequence (engine.h Model::step_captured)
void Model::step_captured(const float labels[3], bool training, uint64_t seed) {
  static_assert(sizeof(StepParams) == 40 && offsetof(StepParams, schedG) == 24 && offsetof(StepParams, schedD) == 32, "StepParams layout");
  if (hyper.gp_mode) { step(labels, training, seed); return; }          // host-seeded draws per step: not recordable
  if (!sp_dev) { AllocScope mine(*ctx, owned_allocs); sp_dev = static_cast<StepParams*>(ctx->alloc(64)); }
  StepParams h{};
  for (int i = 0; i < 3; ++i) h.labels[i] = labels[i];
  h.seed = seed;
  adamw_schedule(hyper.lr, hyper.b1, hyper.b2, arenaG.step + 1, h.schedG);
  adamw_schedule(hyper.d_lr, hyper.d_b1, hyper.d_b2, arenaD.step + 1, h.schedD);
  dev_store_small(ctx->s, sp_dev, &h, sizeof h);      // stream-ordered, in front of this step's launches, and NO host sync (round 5: the
                                                      // synchronising upload made every replayed step wait for the previous one to drain)
  struct Indirect {
    Model& m;
    explicit Indirect(Model& mm) : m(mm) { m.indirect = true; if (m.G) m.G->seed_dev = &m.sp_dev->seed; }
    ~Indirect() { m.indirect = false; if (m.G) m.G->seed_dev = nullptr; }
  } scope(*this);
  const int gi = training ? 1 : 0;
  if (!is_device_build() || step_warm_[gi] == 0) {      // first call (and the host simulator): the phases, eagerly, on the device block
    step(labels, training, seed);
    step_warm_[gi] = 1;
    return;
  }
  // a recorded sequence bakes in what is not in StepParams: one stream or two, and where AdamW sits (round-4 advice)
  const char* am = getenv("SWN_STREAM_ADAMW");
  const int key = (ctx->use_side() ? 1 : 0) | ((am ? atoi(am) : 1) << 1);
  if (step_graph_[gi] && step_graph_key_[gi] != key) { graph_destroy(step_graph_[gi]); step_graph_[gi] = nullptr; }
  if (!step_graph_[gi]) {
    step_graph_key_[gi] = key;
    // record: the same phases on a private capture stream (the side stream joins the capture through the fork / join events)
    stream_sync(ctx->s);
    Stream& s = ctx->s;
    void* caller = s.handle;
    if (!cap_stream_) cap_stream_ = stream_create_current();
    const int stepG = arenaG.step, stepD = arenaD.step, verG = arenaG.version, verD = arenaD.version;
    s.handle = cap_stream_;
    try {
      graph_begin(s);
      try {
        step(labels, training, seed);
        ctx->join_side();
      } catch (...) { graph_abort(s); throw; }
      step_graph_[gi] = graph_end(s);
    } catch (...) { s.handle = caller; throw; }
    s.handle = caller;
    // recording executed nothing: undo the host-side bookkeeping of the recorded pass, the replay below redoes it
    arenaG.step = stepG; arenaD.step = stepD; arenaG.version = verG; arenaD.version = verD;
  }
  graph_launch(step_graph_[gi], ctx->s);
  if (!hyper.warp_mode_ce_only) { arenaD.step += 1; arenaD.version += 1; }
  arenaG.step += 1; arenaG.version += 1;
}

// Data-parallel optimize_parameters with the library's own exchange (engine.h Model::step_dp)
void Model::step_dp(const float labels[3], bool training, uint64_t seed, bool after_forward) {
  if (!ctx->comm_fn) throw Error(1, "step_dp: no communicator attached (swn_ctx_attach_comm)");
  if (!after_forward) forward(training, seed);
  if (!hyper.warp_mode_ce_only) {
    backward_D(labels[0], labels[1]);            // ends joined: every D gradient is ordered on the main stream
    // D's update precedes the D pass of backward_G by data dependence (base_gan.py:199): this exchange is serial
    Stream& cs = ctx->comm_fork();
    ctx->all_reduce_sum(cs, arenaD.g, arenaD.n);
    ctx->comm_join();
    optimizer_step(1);
  }
  const int np = backward_G_parts();
  const int ver = arenaG.version;
  for (int part = 0; part < np; ++part) {
    size_t off = 0, count = 0;
    backward_G_part(labels[2], part, &off, &count, /*join=*/true);      // the bucket's weight gradients are final on the main stream
    // exchange stream: after the bucket's gradients, behind the previous bucket's all-reduce and AdamW (stream order).  Nothing
    // later on the compute streams reads this bucket's weights or gradients again this step (backward_G_streamed's argument):
    // the earlier layers multiply by their own derived operands, which are refreshed from the weights at the next forward --
    // after the join below.
    Stream& cs = ctx->comm_fork();
    ctx->all_reduce_sum(cs, arenaG.g + off, count);
    optimizer_step_range_on(cs, 0, off, count, part == 0);
  }
  arenaG.version = ver + 1;        // one optimizer step
  ctx->comm_join();
}

// BaseGAN.optimize_parameters (models/base_gan.py:194-203): forward, D step, G step.
void Model::step(const float labels[3], bool training, uint64_t seed) {
  forward(training, seed);
  if (!hyper.warp_mode_ce_only) {
    backward_D(labels[0], labels[1]);
    optimizer_step(1);
  }
  // SWN_STREAM_ADAMW=0: AdamW as one launch behind the whole backward pass; 2: the bucketed order also without a side stream
  // (what the host simulator can exercise).  Read per step.
  const char* e = getenv("SWN_STREAM_ADAMW");
  const int mode = e ? atoi(e) : 1;
  if (mode == 2 || (mode != 0 && ctx->use_side())) {
    backward_G_streamed(labels[2]);
    return;
  }
  backward_G(labels[2]);
  optimizer_step(0);
}

}  // namespace swn
