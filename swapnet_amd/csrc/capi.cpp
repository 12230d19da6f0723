// swapnet_amd -- extern "C" boundary (include/swapnet_hip.h).  Plain pointers and sizes only.
#include <cstring>
#include <memory>
#include <string>

#include "../../include/swapnet_hip.h"
#include <cstdlib>

#include "engine.h"

using namespace swn;

// The context is shared-owned by its handle and by every model / pipeline created in it: destroying the handles in
// any order (interpreter shutdown) is safe, and a model's destructor can always hand its buffers back.
struct CtxBox {
  std::unique_ptr<Ctx> c;
  void* owned_stream = nullptr;
  ~CtxBox() { c.reset(); if (owned_stream) stream_destroy(owned_stream); }
};
struct swn_ctx {
  std::shared_ptr<CtxBox> box;
  Ctx* c = nullptr;
};
struct swn_model {
  std::shared_ptr<CtxBox> keep;     // declared first: released after the model
  std::shared_ptr<Model> sharer;    // the model whose arenas this one uses (swn_model_create_shared): outlives it
  std::shared_ptr<Model> m;
  // what it was created with (a sharing model is created like its sharer, at another batch size)
  int kind = 0, is_train = 1, num_roi = 12, body_channels = 3, cloth_channels = 19, patchgan_layers = 3;
  float dropout = 0.5f;
};
struct swn_pipeline {
  std::shared_ptr<CtxBox> keep;
  std::unique_ptr<Pipeline> p;
};

static thread_local std::string g_err;

template <typename F>
static int guard(F f) {
  try {
    f();
    return 0;
  } catch (const Error& e) {
    g_err = e.what();
    return e.code ? e.code : 1;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  } catch (...) {
    g_err = "unknown error";
    return 1;
  }
}
#define REQUIRE(cond, msg) \
  do { if (!(cond)) throw Error(1, msg); } while (0)

extern "C" {

int swn_abi_version(void) { return 6; }
const char* swn_last_error(void) { return g_err.c_str(); }
int swn_is_device_build(void) { return is_device_build(); }

int swn_ctx_create(int device, void* hip_stream, int create_stream, size_t workspace_bytes, swn_ctx** out) {
  return guard([&] {
    REQUIRE(out, "swn_ctx_create: out is NULL");
    auto h = std::make_unique<swn_ctx>();
    h->box = std::make_shared<CtxBox>();
    void* st = hip_stream;
    if (create_stream) { st = stream_create(device); h->box->owned_stream = st; }
    else device_check(device);
    if (workspace_bytes < (size_t)64 << 20) workspace_bytes = (size_t)64 << 20;
    h->box->c = std::make_unique<Ctx>(st, workspace_bytes);
    h->c = h->box->c.get();
    h->c->device_index = device;
    if (!(getenv("SWN_OVERLAP") && atoi(getenv("SWN_OVERLAP")) == 0)) h->c->enable_side(device);
    *out = h.release();
  });
}
int swn_ctx_destroy(swn_ctx* ctx) {
  return guard([&] {
    if (!ctx) return;
    delete ctx;           // the context itself goes with its last model / pipeline
  });
}
int swn_ctx_set_overlap(swn_ctx* ctx, int on) {
  return guard([&] {
    REQUIRE(ctx, "ctx is NULL");
    ctx->c->join_side();
    stream_sync(ctx->c->s);
    ctx->c->side_enabled = on != 0;
  });
}
int swn_ctx_attach_comm(swn_ctx* ctx, swn_allreduce_fn fn, void* comm, int world_size) {
  return guard([&] {
    REQUIRE(ctx, "ctx is NULL");
    ctx->c->join_side();
    ctx->c->attach_comm(reinterpret_cast<Ctx::AllReduceFn>(fn), comm, fn ? world_size : 1);
  });
}
int swn_ctx_set_patchgan_layers(swn_ctx* ctx, int n_layers) {
  return guard([&] {
    REQUIRE(ctx, "ctx is NULL");
    REQUIRE(n_layers >= 0 && n_layers <= 5, "n_layers_D must be in [1, 5], or 0 for the 1x1 PixelDiscriminator");
    ctx->c->patchgan_layers = n_layers;
  });
}
int swn_ctx_sync(swn_ctx* ctx) {
  return guard([&] { REQUIRE(ctx, "ctx is NULL"); stream_sync(ctx->c->s); });
}
int swn_ctx_bytes_allocated(swn_ctx* ctx, size_t* out) {
  return guard([&] { REQUIRE(ctx && out, "NULL argument"); *out = ctx->c->bytes_allocated; });
}

int swn_prof_enable(int on) { return guard([&] { prof_enable(on); }); }
int swn_prof_reset(void) { return guard([&] { prof_reset(); }); }
int swn_prof_report(char* buf, int len) { return prof_report(buf, len); }
int swn_probe_mfma(swn_ctx* ctx, int zeros, int iters, float* out4) {
  return guard([&] {
    REQUIRE(ctx && out4, "swn_probe_mfma: NULL argument");
    REQUIRE(iters >= 1 && iters <= (1 << 20), "swn_probe_mfma: iters out of range");
    probe_mfma(ctx->c->s, zeros, iters, out4);
  });
}
int swn_route_trace(int on) { return guard([&] { route_enable(on); }); }
int swn_route_report(char* buf, int len) { return route_report(buf, len); }

int swn_warp_model_create_ex(swn_ctx* ctx, int batch, int height, int width, int is_train, float dropout,
                             int body_channels, int cloth_channels, swn_model** out) {
  return guard([&] {
    REQUIRE(ctx && out, "NULL argument");
    REQUIRE(batch > 0 && height > 0 && width > 0, "bad shape");
    auto h = std::make_unique<swn_model>();
    h->keep = ctx->box;
    h->m.reset(create_warp_model(*ctx->c, batch, height, width, is_train != 0, dropout, body_channels, cloth_channels));
    h->kind = 0; h->is_train = is_train != 0; h->dropout = dropout; h->body_channels = body_channels; h->cloth_channels = cloth_channels;
    h->patchgan_layers = ctx->c->patchgan_layers;
    *out = h.release();
  });
}
int swn_warp_model_create(swn_ctx* ctx, int batch, int height, int width, int is_train, float dropout,
                          swn_model** out) {
  return swn_warp_model_create_ex(ctx, batch, height, width, is_train, dropout, 3, 19, out);
}
int swn_texture_model_create_ex(swn_ctx* ctx, int batch, int height, int width, int is_train, int num_roi,
                                int cloth_channels, swn_model** out) {
  return guard([&] {
    REQUIRE(ctx && out, "NULL argument");
    auto h = std::make_unique<swn_model>();
    h->keep = ctx->box;
    h->m.reset(create_texture_model(*ctx->c, batch, height, width, is_train != 0, num_roi, cloth_channels));
    h->kind = 1; h->is_train = is_train != 0; h->num_roi = num_roi; h->cloth_channels = cloth_channels;
    h->patchgan_layers = ctx->c->patchgan_layers;
    *out = h.release();
  });
}
int swn_texture_model_create(swn_ctx* ctx, int batch, int height, int width, int is_train, int num_roi,
                             swn_model** out) {
  return swn_texture_model_create_ex(ctx, batch, height, width, is_train, num_roi, 19, out);
}
int swn_model_create_shared(swn_model* sharer, int batch, int height, int width, swn_model** out) {
  return guard([&] {
    REQUIRE(sharer && out, "NULL argument");
    REQUIRE(batch > 0 && height > 0 && width > 0, "bad shape");
    std::shared_ptr<Model> root = sharer->sharer ? sharer->sharer : sharer->m;        // always the owner of the arenas
    Ctx& c = *root->ctx;
    auto h = std::make_unique<swn_model>();
    h->keep = sharer->keep;
    h->sharer = root;
    h->kind = sharer->kind; h->is_train = sharer->is_train; h->dropout = sharer->dropout; h->num_roi = sharer->num_roi;
    h->body_channels = sharer->body_channels; h->cloth_channels = sharer->cloth_channels; h->patchgan_layers = sharer->patchgan_layers;
    const int layers_was = c.patchgan_layers;
    c.patchgan_layers = sharer->patchgan_layers;
    try {
      if (h->kind == 0)
        h->m.reset(create_warp_model(c, batch, height, width, h->is_train != 0, h->dropout, h->body_channels, h->cloth_channels, root.get()));
      else
        h->m.reset(create_texture_model(c, batch, height, width, h->is_train != 0, h->num_roi, h->cloth_channels, root.get()));
    } catch (...) { c.patchgan_layers = layers_was; throw; }
    c.patchgan_layers = layers_was;
    h->m->hyper = root->hyper;
    *out = h.release();
  });
}
int swn_model_destroy(swn_model* m) {
  return guard([&] { delete m; });
}

int swn_model_set_hyper(swn_model* m, const swn_hyper* h) {
  return guard([&] {
    REQUIRE(m && h, "NULL argument");
    const Hyper before = m->m->hyper;
    struct Recheck {          // a recorded step (swn_model_step_captured) carries the hyper-parameters it was recorded with
      Model& mm; const Hyper& was;
      ~Recheck() { if (memcmp(&was, &mm.hyper, sizeof(Hyper)) != 0) mm.invalidate_step_graphs(); }
    } recheck{*m->m, before};
    Hyper& y = m->m->hyper;
    y.lr = h->lr; y.d_lr = h->d_lr; y.weight_decay = h->weight_decay; y.d_weight_decay = h->d_weight_decay;
    y.b1 = h->b1; y.b2 = h->b2; y.lambda_gan = h->lambda_gan; y.lambda_ce = h->lambda_ce; y.lambda_l1 = h->lambda_l1;
    y.lambda_content = h->lambda_content; y.lambda_style = h->lambda_style;
    REQUIRE(h->gan_mode >= 0 && h->gan_mode <= 2, "gan mode not implemented");
    y.gan_mode = h->gan_mode; y.warp_mode_ce_only = h->warp_mode_ce;
    y.grad_scale = h->grad_scale > 0.f ? h->grad_scale : 1.f;
    // negative = inherit optimizer_G's value; 0 is a legitimate beta (WGAN-GP style (0, 0.9) betas)
    y.d_b1 = h->d_b1 >= 0.f ? h->d_b1 : h->b1; y.d_b2 = h->d_b2 >= 0.f ? h->d_b2 : h->b2;
    REQUIRE(h->gp_mode >= 0 && h->gp_mode <= 3, "gradient penalty mode not implemented");
    REQUIRE(h->gp_mode == 0 || m->m->supports_gradient_penalty(),
            "gradient penalty modes are not implemented for the texture model (the reference's call fails there too) nor for the "
            "1x1 PixelDiscriminator");
    y.gp_mode = h->gp_mode; y.lambda_gp = h->lambda_gp;
  });
}

static ParamArena& arena_of(swn_model* m, int net) {
  REQUIRE(m, "model is NULL");
  ParamArena* a = m->m->arena_ptr(net);
  REQUIRE(a, "this model has no such network");
  return *a;
}

int swn_model_param_count(swn_model* m, int net, int* out) {
  return guard([&] { REQUIRE(out, "NULL"); *out = (int)arena_of(m, net).params.size(); });
}
int swn_model_param_info(swn_model* m, int net, int index, char* name, int name_len, int shape[4], int* ndim) {
  return guard([&] {
    ParamArena& a = arena_of(m, net);
    REQUIRE(index >= 0 && index < (int)a.params.size(), "param index out of range");
    const ParamDesc& d = a.params[index];
    if (name && name_len > 0) { std::strncpy(name, d.name.c_str(), name_len - 1); name[name_len - 1] = 0; }
    if (d.is_bias) {
      if (shape) { shape[0] = d.n_logical; shape[1] = shape[2] = shape[3] = 1; }
      if (ndim) *ndim = 1;
    } else {
      if (shape) {
        if (d.ws.kind == WK_CONV) { shape[0] = d.ws.Co; shape[1] = d.ws.Ci; }
        else { shape[0] = d.ws.Ci; shape[1] = d.ws.Co; }
        shape[2] = d.ws.KH; shape[3] = d.ws.KW;
      }
      if (ndim) *ndim = 4;
    }
  });
}
static const ParamDesc& find_param(ParamArena& a, const char* name) {
  REQUIRE(name, "name is NULL");
  auto it = a.index.find(name);
  if (it == a.index.end()) throw Error(1, std::string("unknown parameter ") + name);
  return a.params[it->second];
}
int swn_model_param_set(swn_model* m, int net, int which, const char* name, const float* src) {
  return guard([&] {
    ParamArena& a = arena_of(m, net);
    REQUIRE(which >= 0 && which <= 3 && src, "bad argument");
    const ParamDesc& d = find_param(a, name);
    Stream& s = m->m->ctx->s;
    if (d.is_bias) dev_copy(s, a.base(which) + d.off, src, d.n_logical * sizeof(float));
    else pack_weight(s, d.ws, src, a.base(which) + d.off);
    if (which == 0) a.version += 1;
  });
}
int swn_model_param_get(swn_model* m, int net, int which, const char* name, float* dst) {
  return guard([&] {
    ParamArena& a = arena_of(m, net);
    REQUIRE(which >= 0 && which <= 3 && dst, "bad argument");
    const ParamDesc& d = find_param(a, name);
    Stream& s = m->m->ctx->s;
    if (d.is_bias) dev_copy(s, dst, a.base(which) + d.off, d.n_logical * sizeof(float));
    else unpack_weight(s, d.ws, a.base(which) + d.off, dst);
  });
}
int swn_model_optim_step_get(swn_model* m, int net, int* step) {
  return guard([&] { REQUIRE(step, "NULL"); *step = arena_of(m, net).step; });
}
int swn_model_optim_step_set(swn_model* m, int net, int step) {
  return guard([&] { arena_of(m, net).step = step; });
}

int swn_model_set_input(swn_model* m, int slot, const float* src, int n, int c, int h, int w) {
  return guard([&] { REQUIRE(m && src, "NULL argument"); m->m->set_input(slot, src, n, c, h, w); });
}
int swn_model_set_input_labels(swn_model* m, int slot, const int32_t* lab, int n, int h, int w) {
  return guard([&] { REQUIRE(m && lab, "NULL argument"); m->m->set_input_labels(slot, lab, n, h, w); });
}
int swn_model_get_output(swn_model* m, int slot, float* dst) {
  return guard([&] { REQUIRE(m && dst, "NULL argument"); m->m->get_output(slot, dst); });
}
int swn_model_get_tap_grad(swn_model* m, int net, const char* name, float* dst, int shape[4]) {
  return guard([&] {
    REQUIRE(m && name, "NULL argument");
    Net* n = m->m->net_for_taps(net);
    REQUIRE(n, "no such network");
    auto it = n->taps.find(name);
    if (it == n->taps.end()) throw Error(1, std::string("unknown tap ") + name);
    REQUIRE(it->second.has_grad, "tap carries no gradient");
    const TView& v = it->second.g;
    if (shape) { shape[0] = v.N; shape[1] = v.C; shape[2] = v.H; shape[3] = v.W; }
    if (dst) nhwc_to_nchw(m->m->ctx->s, v, dst, v.C);
  });
}
int swn_model_get_tap(swn_model* m, int net, const char* name, float* dst, int shape[4]) {
  return guard([&] {
    REQUIRE(m && name, "NULL argument");
    Net* n = m->m->net_for_taps(net);
    REQUIRE(n, "no such network");
    auto it = n->taps.find(name);
    if (it == n->taps.end()) throw Error(1, std::string("unknown tap ") + name);
    const TView& v = it->second.v;
    if (shape) { shape[0] = v.N; shape[1] = v.C; shape[2] = v.H; shape[3] = v.W; }
    if (dst) nhwc_to_nchw(m->m->ctx->s, v, dst, v.C);
  });
}
int swn_model_dropout_sites(swn_model* m, int net, int* count) {
  return guard([&] {
    REQUIRE(m && count, "NULL argument");
    Net* n = m->m->net_for_taps(net);
    REQUIRE(n, "no such network");
    *count = (int)n->drop_sites.size();
  });
}
int swn_model_dropout_mask(swn_model* m, int net, int site, uint64_t seed, float* dst, int shape[4], float* p) {
  return guard([&] {
    REQUIRE(m, "NULL argument");
    Net* n = m->m->net_for_taps(net);
    REQUIRE(n, "no such network");
    REQUIRE(site >= 0 && site < (int)n->drop_sites.size(), "dropout site out of range");
    const Net::DropSite& d = n->drop_sites[site];
    if (shape) { shape[0] = d.N; shape[1] = d.C; shape[2] = d.H; shape[3] = d.W; }
    if (p) *p = d.p;
    if (dst) dropout_mask(m->m->ctx->s, d.N, d.H, d.W, d.C, d.p, Net::drop_seed(seed, d.salt), dst);
  });
}
int swn_model_act_sites(swn_model* m, int net, int* count) {
  return guard([&] {
    REQUIRE(m && count, "NULL argument");
    Net* n = m->m->net_for_patterns(net);
    REQUIRE(n, "no such network");
    *count = (int)n->act_sites.size();
  });
}
int swn_model_act_pattern(swn_model* m, int net, int site, uint8_t* dst, int shape[4], int* kind) {
  return guard([&] {
    REQUIRE(m, "NULL argument");
    Net* n = m->m->net_for_patterns(net);
    REQUIRE(n, "no such network");
    REQUIRE(site >= 0 && site < (int)n->act_sites.size(), "activation site out of range");
    const Net::ActSite& a = n->act_sites[site];
    if (shape) { shape[0] = a.y.N; shape[1] = a.y.C; shape[2] = a.y.H; shape[3] = a.y.W; }
    if (kind) *kind = a.kind;
    if (dst) {
      if (a.kind == 2) pool_pattern(m->m->ctx->s, a.x, a.y, dst);
      else act_pattern(m->m->ctx->s, a.y, dst);
    }
  });
}
int swn_model_set_style_context(swn_model* m, const float* all_out, const float* all_tgt, int n_total, int n0) {
  return guard([&] { REQUIRE(m && all_out && all_tgt, "NULL argument"); m->m->set_style_context(all_out, all_tgt, n_total, n0); });
}
int swn_model_set_gp_random(swn_model* m, const float* alpha, const float* beta) {
  return guard([&] { REQUIRE(m, "NULL argument"); m->m->set_gp_random(alpha, beta); });
}
int swn_model_discriminate(swn_model* m, const float* x, float* pred) {
  return guard([&] { REQUIRE(m && x && pred, "NULL argument"); m->m->discriminate(x, pred); });
}
int swn_model_perceptual(swn_model* m, const float* output, const float* target, int use_style, float* out2,
                         float content_w, float style_w, float* d_output) {
  return guard([&] {
    REQUIRE(m && output && target && out2, "NULL argument");
    m->m->perceptual(output, target, use_style, out2, content_w, style_w, d_output);
  });
}
int swn_pipeline_create(swn_model* warp, swn_model* texture, swn_pipeline** out) {
  return guard([&] {
    REQUIRE(warp && texture && out, "NULL argument");
    auto h = std::make_unique<swn_pipeline>();
    h->keep = warp->keep;
    h->p = std::make_unique<Pipeline>(*warp->m, *texture->m);
    *out = h.release();
  });
}
int swn_pipeline_destroy(swn_pipeline* p) {
  return guard([&] { delete p; });
}
int swn_pipeline_run(swn_pipeline* p, int use_graph, int* graph_replayed) {
  return guard([&] {
    REQUIRE(p, "NULL argument");
    const bool had = p->p->graph_captured();
    p->p->run(use_graph != 0);
    if (graph_replayed) *graph_replayed = (use_graph && had) ? 1 : 0;
  });
}
int swn_pipeline_labels(swn_pipeline* p, int32_t** dev_labels) {
  return guard([&] { REQUIRE(p && dev_labels, "NULL argument"); *dev_labels = p->p->labels(); });
}
int swn_model_forward(swn_model* m, int training, uint64_t seed) {
  return guard([&] { REQUIRE(m, "NULL"); m->m->forward(training != 0, seed); });
}
int swn_model_backward_D(swn_model* m, float lf, float lr) {
  return guard([&] { REQUIRE(m && m->m->is_train, "model was not created for training"); m->m->backward_D(lf, lr); });
}
int swn_model_backward_G(swn_model* m, float lr) {
  return guard([&] { REQUIRE(m && m->m->is_train, "model was not created for training"); m->m->backward_G(lr); });
}
int swn_model_backward_G_parts(swn_model* m, int* nparts) {
  return guard([&] { REQUIRE(m && nparts, "NULL argument"); *nparts = m->m->backward_G_parts(); });
}
int swn_model_backward_G_part(swn_model* m, float lr, int part, size_t* off, size_t* count) {
  return guard([&] {
    REQUIRE(m && m->m->is_train && part >= 0 && part < m->m->backward_G_parts(), "bad argument");
    m->m->backward_G_part(lr, part, off, count);
  });
}
int swn_model_optimizer_step(swn_model* m, int net) {
  return guard([&] {
    REQUIRE(m && m->m->is_train, "model was not created for training");
    REQUIRE(net == 0 || net == 1, "net must be 0 (G) or 1 (D)");
    m->m->optimizer_step(net);
  });
}
int swn_model_optimizer_step_range(swn_model* m, int net, size_t off, size_t count, int first) {
  return guard([&] {
    REQUIRE(m && m->m->is_train, "model was not created for training");
    REQUIRE(net == 0 || net == 1, "net must be 0 (G) or 1 (D)");
    m->m->optimizer_step_range(net, off, count, first);
  });
}
int swn_model_step(swn_model* m, const float labels[3], int training, uint64_t seed) {
  return guard([&] {
    REQUIRE(m && labels && m->m->is_train, "model was not created for training");
    m->m->step(labels, training != 0, seed);
  });
}
int swn_model_step_dp(swn_model* m, const float labels[3], int training, uint64_t seed, int after_forward) {
  return guard([&] {
    REQUIRE(m && labels && m->m->is_train, "model was not created for training");
    m->m->step_dp(labels, training != 0, seed, after_forward != 0);
  });
}
int swn_model_step_captured(swn_model* m, const float labels[3], int training, uint64_t seed) {
  return guard([&] {
    REQUIRE(m && labels && m->m->is_train, "model was not created for training");
    m->m->step_captured(labels, training != 0, seed);
  });
}
int swn_model_get_losses(swn_model* m, float* host_out, int n) {
  return guard([&] {
    REQUIRE(m && host_out && n > 0 && n <= L_COUNT, "bad argument");
    dev_download(m->m->ctx->s, host_out, m->m->losses, n * sizeof(float));
  });
}
int swn_model_grad_arena(swn_model* m, int net, float** p, size_t* count) {
  return guard([&] { ParamArena& a = arena_of(m, net); REQUIRE(p && count, "NULL"); *p = a.g; *count = a.n; });
}
int swn_model_weight_arena(swn_model* m, int net, float** p, size_t* count) {
  return guard([&] { ParamArena& a = arena_of(m, net); REQUIRE(p && count, "NULL"); *p = a.w; *count = a.n; a.version += 1; });
}

int swn_model_arena(swn_model* m, int net, int which, float** p, size_t* count) {
  return guard([&] {
    ParamArena& a = arena_of(m, net);
    REQUIRE(p && count && which >= 0 && which <= 3, "bad argument");
    *p = a.base(which); *count = a.n;
    if (which == 0) a.version += 1;
  });
}

// ---- operator level -----------------------------------------------------------------------
int swn_op_roi_align(swn_ctx* ctx, const float* tex, int b, int c, int h, int w, const float* rois, int r, int ph,
                     int pw, float* out) {
  return guard([&] {
    REQUIRE(ctx && tex && rois && out, "NULL argument");
    Ctx tmp(ctx->c->s);
    ParamArena A; Net net(tmp, A);
    Var t = net.alloc_var(b, h, w, round_up(c, 4), false);
    Var o = net.alloc_var(b, ph, pw, round_up(r * c, 4), false);
    nchw_to_nhwc(tmp.s, tex, b, c, h, w, t.v);
    roi_align_fwd(tmp.s, t.v, c, rois, r, o.v);
    nhwc_to_nchw(tmp.s, o.v, out, r * c);
    stream_sync(tmp.s);
  });
}
int swn_op_roi_align_indices(swn_ctx* ctx, const float* rois, int k, int h, int w, int ph, int pw, int32_t* idx,
                             uint8_t* valid) {
  return guard([&] {
    REQUIRE(ctx && rois && idx && valid, "NULL argument");
    roi_align_indices(ctx->c->s, rois, k, h, w, ph, pw, idx, valid);
    stream_sync(ctx->c->s);
  });
}
int swn_op_decode_labels(swn_ctx* ctx, const float* x, int b, int c, int h, int w, uint8_t* rgb) {
  return guard([&] {
    REQUIRE(ctx && x && rgb, "NULL argument");
    Ctx tmp(ctx->c->s);
    ParamArena A; Net net(tmp, A);
    Var t = net.alloc_var(b, h, w, round_up(c, 4), false);
    nchw_to_nhwc(tmp.s, x, b, c, h, w, t.v);
    decode_labels(tmp.s, t.v, c, rgb);
    stream_sync(tmp.s);
  });
}
int swn_op_argmax_labels(swn_ctx* ctx, const float* x, int b, int c, int h, int w, int32_t* labels) {
  return guard([&] {
    REQUIRE(ctx && x && labels, "NULL argument");
    Ctx tmp(ctx->c->s);
    ParamArena A; Net net(tmp, A);
    Var t = net.alloc_var(b, h, w, round_up(c, 4), false);
    nchw_to_nhwc(tmp.s, x, b, c, h, w, t.v);
    argmax_labels(tmp.s, t.v, c, labels);
    stream_sync(tmp.s);
  });
}
int swn_op_labels_to_onehot(swn_ctx* ctx, const int32_t* labels, int b, int c, int h, int w, float* out) {
  return guard([&] {
    REQUIRE(ctx && labels && out, "NULL argument");
    Ctx tmp(ctx->c->s);
    ParamArena A; Net net(tmp, A);
    Var t = net.alloc_var(b, h, w, round_up(c, 4), false);
    labels_to_onehot(tmp.s, labels, t.v, c);
    nhwc_to_nchw(tmp.s, t.v, out, c);
    stream_sync(tmp.s);
  });
}

int swn_op_conv(swn_ctx* ctx, int kind, int transposed, int what, int naive, float* x, int n, int ci, int h, int w,
                float* wgt, int co, const float* bias, int act, float* y) {
  return guard([&] {
    REQUIRE(ctx && x && wgt && y, "NULL argument");
    REQUIRE(kind >= 0 && kind <= 5 && what >= 0 && what <= 2, "bad kind/what");
    REQUIRE(what == 0 || act == ACT_NONE, "backward entry points take the gradient of the pre-activation output");
    // (set before the net is built: with the checkers forced no pre-cut-only operands are planned)
    struct Restore { ~Restore() { conv_force_naive(0); } } restore;
    conv_force_naive(naive);
    Ctx tmp(ctx->c->s);
    ParamArena A;
    Net net(tmp, A);
    const int Cip = round_up(ci, 4), Cop = round_up(co, 4);
    int Ho, Wo;
    if (transposed) { Ho = 2 * h; Wo = 2 * w; }
    else if (kind == CK_K4S2) { Ho = h / 2; Wo = w / 2; }
    else if (kind == CK_K4S1) { Ho = h - 1; Wo = w - 1; }
    else if (kind == CK_TAIL_UP) { Ho = 2 * h; Wo = 2 * w; }
    else { Ho = h; Wo = w; }
    Var xv = net.alloc_var(n, h, w, Cip, true);
    Var yv = net.alloc_var(n, Ho, Wo, Cop, true);
    if (transposed) { REQUIRE(Cip == ci, "transposed conv needs Ci % 4 == 0"); net.convT("l", xv, yv, co, bias != nullptr); }
    else net.conv("l", xv, yv, (ConvKind)kind, ci, co, bias != nullptr, act);
    A.allocate(tmp);
    net.finalize({});
    Stream& s = tmp.s;
    const ParamDesc& wd = A.params[A.index.at("l.weight")];
    if (what != 2) nchw_to_nhwc(s, x, n, ci, h, w, xv.v);
    if (what != 1) pack_weight(s, wd.ws, wgt, A.w + wd.off);
    if (bias) dev_copy(s, A.w + A.params[A.index.at("l.bias")].off, bias, co * sizeof(float));
    if (what == 0) {
      net.forward();
      nhwc_to_nchw(s, yv.v, y, co);
    } else {
      nchw_to_nhwc(s, y, n, co, Ho, Wo, yv.g);
      if (what == 1) {
        net.refresh_dgrad();        // a layer that is not the net's first also forms its input gradient: from operands that exist
        net.backward(true, false);
        unpack_weight(s, wd.ws, A.g + wd.off, wgt);
      } else {
        A.version += 1;
        net.refresh_dgrad();
        net.backward(false, true);
        nhwc_to_nchw(s, xv.g, x, ci);
      }
    }
    stream_sync(s);
  });
}

int swn_op_instance_norm_act(swn_ctx* ctx, const float* x, int n, int c, int h, int w, int act, float* y) {
  return guard([&] {
    REQUIRE(ctx && x && y && c % 4 == 0, "bad argument (C must be a multiple of 4)");
    Ctx tmp(ctx->c->s);
    ParamArena A; Net net(tmp, A);
    Var xv = net.alloc_var(n, h, w, c, true), yv = net.alloc_var(n, h, w, c, true);
    net.norm_act(xv, yv, true, act, 0.f);
    net.finalize({});
    nchw_to_nhwc(tmp.s, x, n, c, h, w, xv.v);
    net.forward();
    nhwc_to_nchw(tmp.s, yv.v, y, c);
    stream_sync(tmp.s);
  });
}
int swn_op_instance_norm_act_bwd(swn_ctx* ctx, const float* x, const float* dy, int n, int c, int h, int w, int act,
                                 float* dx) {
  return guard([&] {
    REQUIRE(ctx && x && dy && dx && c % 4 == 0, "bad argument (C must be a multiple of 4)");
    Ctx tmp(ctx->c->s);
    ParamArena A; Net net(tmp, A);
    Var xv = net.alloc_var(n, h, w, c, true), yv = net.alloc_var(n, h, w, c, true);
    net.norm_act(xv, yv, true, act, 0.f);
    net.finalize({});
    nchw_to_nhwc(tmp.s, x, n, c, h, w, xv.v);
    net.forward();
    nchw_to_nhwc(tmp.s, dy, n, c, h, w, yv.g);
    net.backward(false, true);
    nhwc_to_nchw(tmp.s, xv.g, dx, c);
    stream_sync(tmp.s);
  });
}
int swn_op_affine_gather(swn_ctx* ctx, const float* src, float* dst, int b, int c, int h, int w, const double* maps,
                         int nmaps) {
  return guard([&] {
    REQUIRE(ctx && src && dst && maps, "NULL argument");
    REQUIRE(src != dst, "affine_gather is out of place");
    affine_gather(ctx->c->s, src, dst, b, c, h, w, maps, nmaps);
  });
}
int swn_op_gan_loss(swn_ctx* ctx, int gan_mode, const float* pred, int n, int c, int h, int w, float label,
                    int target_is_real, float grad_scale, float* loss_out, float* dpred) {
  return guard([&] {
    REQUIRE(ctx && pred && loss_out, "NULL argument");
    REQUIRE(gan_mode >= 0 && gan_mode <= 2, "gan mode not implemented");
    Ctx tmp(ctx->c->s);
    ParamArena A; Net net(tmp, A);
    // the prediction map (N,C,H,W) is reduced over all of its elements: view it as N*C single-channel images
    Var pv = net.alloc_var(n * c, h, w, 4, true);
    nchw_to_nhwc(tmp.s, pred, n * c, 1, h, w, pv.v);
    float* lo = static_cast<float*>(tmp.alloc(sizeof(float)));
    const TView* dp = dpred ? &pv.g : nullptr;
    if (gan_mode == 0) bce_logits_loss(tmp.s, pv.v, label, grad_scale, lo, dp);
    else if (gan_mode == 1) lsgan_loss(tmp.s, pv.v, label, grad_scale, lo, dp);
    else wgan_loss(tmp.s, pv.v, target_is_real ? -1.f : 1.f, grad_scale, lo, dp);
    dev_copy(tmp.s, loss_out, lo, sizeof(float));
    if (dpred) nhwc_to_nchw(tmp.s, pv.g, dpred, 1);
    stream_sync(tmp.s);
  });
}
int swn_op_norm_act_bwd2(swn_ctx* ctx, const float* x, const float* gy, const float* u, int n, int c, int h, int w, int act,
                         float* uy, float* ax) {
  return guard([&] {
    REQUIRE(ctx && x && gy && u && uy && ax && c % 4 == 0, "bad argument (C must be a multiple of 4)");
    Ctx tmp(ctx->c->s);
    ParamArena A; Net net(tmp, A);
    Var xv = net.alloc_var(n, h, w, c, true), yv = net.alloc_var(n, h, w, c, true), uv = net.alloc_var(n, h, w, c, true);
    net.norm_act(xv, yv, true, act, 0.f);
    net.finalize({});
    nchw_to_nhwc(tmp.s, x, n, c, h, w, xv.v);
    nchw_to_nhwc(tmp.s, gy, n, c, h, w, yv.g);
    nchw_to_nhwc(tmp.s, u, n, c, h, w, uv.v);
    net.forward();                                  // InstanceNorm statistics of x
    // the statistics live with the op: fetch them through a second-order call on the same buffers
    NormActBwd2Args b2;
    b2.u = uv.v; b2.gy = yv.g; b2.x = xv.v; b2.stats = net.last_stats; b2.uy = uv.g; b2.ax = xv.g; b2.act = act;
    norm_act_bwd2(tmp.s, b2);
    nhwc_to_nchw(tmp.s, uv.g, uy, c);
    nhwc_to_nchw(tmp.s, xv.g, ax, c);
    stream_sync(tmp.s);
  });
}
int swn_op_norm_act_dropout(swn_ctx* ctx, const float* x, const float* dy, int n, int c, int h, int w, int norm, int act,
                            float p, uint64_t seed, float* y, float* mask, float* dx) {
  return guard([&] {
    REQUIRE(ctx && x && y && c % 4 == 0, "bad argument (C must be a multiple of 4)");
    REQUIRE(p >= 0.f && p < 1.f, "dropout probability must be in [0, 1)");
    Ctx tmp(ctx->c->s);
    ParamArena A; Net net(tmp, A);
    Var xv = net.alloc_var(n, h, w, c, true), yv = net.alloc_var(n, h, w, c, true);
    net.norm_act(xv, yv, norm != 0, act, p);
    net.finalize({});
    net.training = true; net.seed = seed;
    nchw_to_nhwc(tmp.s, x, n, c, h, w, xv.v);
    net.forward();
    nhwc_to_nchw(tmp.s, yv.v, y, c);
    if (mask) {
      if (p > 0.f) {
        const Net::DropSite& d = net.drop_sites.at(0);
        dropout_mask(tmp.s, d.N, d.H, d.W, d.C, d.p, Net::drop_seed(seed, d.salt), mask);
      } else {
        REQUIRE(false, "mask requested with p == 0");
      }
    }
    if (dy && dx) {
      nchw_to_nhwc(tmp.s, dy, n, c, h, w, yv.g);
      net.backward(false, true);
      nhwc_to_nchw(tmp.s, xv.g, dx, c);
    }
    stream_sync(tmp.s);
  });
}
int swn_op_adamw(swn_ctx* ctx, float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2,
                 float eps, float wd, int step) {
  return guard([&] {
    REQUIRE(ctx && p && g && m && v, "NULL argument");
    AdamWArgs a{p, g, m, v, n, lr, b1, b2, eps, wd, step};
    adamw_step(ctx->c->s, a);
    stream_sync(ctx->c->s);
  });
}

}  // extern "C"
