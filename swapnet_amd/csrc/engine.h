// swapnet_amd -- native execution engine: parameter arenas, a small tape of fused ops over
// NHWC views, and the network builders for the SwapNet hot path.  Host C++ only; every
// device action goes through ops.h.
#pragma once
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "ops.h"

namespace swn {

struct Ctx {
  Stream s;
  // Side stream for weight-gradient work (wgrad GEMMs, their Winograd transforms, bias gradients, slab
  // sums).  In backward only the input-gradient chain is sequential; a layer's weight gradient just has
  // to be done before the optimizer step.  Forked per layer after dY is final (event on `s`), joined at
  // the end of every backward range: the HBM-bound kernels of one stream overlap the MFMA-bound GEMMs
  // of the other and kernel tails fill.  Own split / partial workspace.
  Stream side;
  bool has_side = false, side_dirty = false;
  bool side_enabled = true;   // runtime switch (swn_ctx_set_overlap): off = everything in order on `s`
  bool use_side() const { return has_side && side_enabled; }
  std::vector<void*> fork_events;
  size_t fork_i = 0;
  void* join_event = nullptr;
  void* owned_side_stream = nullptr;
  void enable_side(int device);
  Stream& fork_side();        // side stream, ordered after everything enqueued on `s` so far
  void join_side();           // `s` continues after the side stream's work (no-op if nothing was forked)
  // Device allocations are owned by the context, or -- while an AllocScope is open -- by the object that opened it
  // (a Model: everything it allocates, at construction or lazily, is released by its destructor).
  typedef std::vector<std::pair<void*, size_t>> AllocList;
  AllocList allocs;
  AllocList* sink = nullptr;
  void release(AllocList& list);          // frees the list's buffers (device-synchronising) and empties it
  size_t bytes_allocated = 0;
  // ---- library-owned gradient exchange (swn_ctx_attach_comm, Model::step_dp) ---------------------------
  // The caller hands over an all-reduce entry point with ncclAllReduce's signature and the communicator it runs on (RCCL's
  // own symbol from the library the process already holds: no second copy is linked here).  The exchange gets a stream of
  // its own; ordering against the compute streams is by explicit events, recorded and waited for by this library -- nothing
  // depends on which stream the host framework considers current.
  typedef int (*AllReduceFn)(const void* sendbuf, void* recvbuf, size_t count, int dtype, int op, void* comm, void* stream);
  AllReduceFn comm_fn = nullptr;
  void* comm_handle = nullptr;
  int comm_world = 1;
  int device_index = 0;
  Stream comm_stream;                // (the main stream itself on the host simulator: everything is in order there)
  void* owned_comm_stream = nullptr;
  std::vector<void*> comm_events;
  size_t comm_ev_i = 0;
  void* comm_join_event = nullptr;
  bool comm_dirty = false;
  void attach_comm(AllReduceFn fn, void* comm, int world);     // fn == NULL detaches
  Stream& comm_fork();               // the exchange stream, ordered after everything enqueued on `s` so far
  void comm_join();                  // `s` continues after the exchange stream's work (no-op if nothing was forked)
  void all_reduce_sum(Stream& on, float* buf, size_t count);   // in place, SUM over the communicator's ranks
  // discriminators.define_D(..., n_layers_D) (modules/discriminators.py:45-88, base_gan.py:147): stride-2 levels of the PatchGAN of
  // every model created on this context afterwards (3 = the reference's "basic" 70x70 PatchGAN)
  int patchgan_layers = 3;
  explicit Ctx(void* stream, size_t ws_bytes);
  explicit Ctx(const Stream& shared);      // borrows stream + workspace of another context
  bool owns_ws = true;
  ~Ctx();
  void* alloc(size_t bytes);
  Ctx(const Ctx&) = delete;
  Ctx& operator=(const Ctx&) = delete;
};

struct AllocScope {              // RAII: allocations made while it lives go to `list`
  Ctx& c; Ctx::AllocList* prev;
  AllocScope(Ctx& ctx, Ctx::AllocList& list) : c(ctx), prev(ctx.sink) { c.sink = &list; }
  ~AllocScope() { c.sink = prev; }
  AllocScope(const AllocScope&) = delete;
  AllocScope& operator=(const AllocScope&) = delete;
};

// activation + its gradient (same geometry).  gbase = start of the gradient allocation
// (identifies the buffer for the accumulate planner).
struct Var {
  TView v, g;
  float* vbase = nullptr;      // start of the value allocation (identifies the buffer for the amax-slot registry)
  float* gbase = nullptr;
  bool has_grad = false;
  Var slice(int c0, int c) const;
  Var batch(int n0, int n) const;
};

struct ParamDesc {
  std::string name;
  bool is_bias = false;
  WShape ws{};
  size_t off = 0, elems = 0;   // floats in the arena (elems multiple of 4)
  int n_logical = 0;           // bias length
  std::vector<int32_t> cimap;
};

// One contiguous arena per network: weights, grads and both Adam moments share offsets, so
// AdamW is a single launch and the data-parallel gradient exchange is one flat buffer.
struct ParamArena {
  std::vector<ParamDesc> params;
  std::map<std::string, int> index;
  size_t n = 0;
  float *w = nullptr, *g = nullptr, *m = nullptr, *v = nullptr;
  int step = 0;
  int version = 1;        // bumped whenever the weights change (dgrad operands follow it)
  bool frozen = false;    // after allocate(): builders may only re-bind existing params (shared nets)
  int add_weight(const std::string& name, int kind, int Co, int Ci, int KH, int KW, int Cip,
                 const std::vector<int32_t>* cimap);
  int add_bias(const std::string& name, int n);
  void allocate(Ctx& c);
  float* base(int which) const { return which == 0 ? w : which == 1 ? g : which == 2 ? m : v; }
};

enum ConvKind { CK_K4S2 = 0, CK_K3S1_REFLECT, CK_K4S1, CK_K3S1_ZERO, CK_TAIL_UP, CK_K1S1 /* 1x1, stride 1, no padding: PixelDiscriminator */ };

class Net;
struct Op {
  std::string label;
  std::function<void(Net&)> fwd;
  std::function<void(Net&, Op&, bool wgrad, bool igrad)> bwd;
  std::vector<Var> grad_targets;   // gradients this op writes in bwd, in execution order
  std::vector<int> acc;            // planned: 0 overwrite / 1 accumulate
  bool reads_net_input = false;
  size_t param_off = (size_t)-1;   // arena offset of this op's weight (ops without parameters: -1)
  std::function<void(Net&)> repack;   // refresh the derived operands (dgrad re-packs, Winograd filters, pre-cut panels) from the arena
  // with a side stream the refresh of all ops is started there at the top of forward(); `ready` is recorded behind this op's
  // share, so the op waits for its own operands only (Net::need)
  void* ready = nullptr;
  bool ready_pending = false;
  int repack_index = -1;    // position among the net's operand-carrying ops (Net::prefetch_dgrad)
};

class Net {
 public:
  Net(Ctx& c, ParamArena& a) : ctx(c), arena(a) {}
  ~Net() {
    if (refresh_event) event_destroy(refresh_event);
    for (auto& op : ops) if (op->ready) event_destroy(op->ready);
  }
  Ctx& ctx;
  ParamArena& arena;
  std::vector<std::unique_ptr<Op>> ops;
  std::map<std::string, Var> taps;
  bool training = false;
  uint64_t seed = 0;
  const uint64_t* seed_dev = nullptr;   // captured step: the step seed in device memory (NormActArgs::seed_base); else NULL
  float* dg = nullptr;      // dgrad operands of this net's convs
  size_t dg_n = 0;
  // weight operands pre-cut into bf16 planes for the round-3 ring kernel (ops.h conv_precut): forward and input-gradient
  // panels of every conv whose launches take it
  uint16_t* dgp = nullptr;
  size_t dgp_n = 0;
  size_t reserve_dgp(size_t elems) { const size_t off = dgp_n; dgp_n += (elems + 7) / 8 * 8; return off; }
  // amax slots (ops.h ConvFwdArgs::x_amax): one per tensor that feeds a two-plane GEMM and is written by a kernel of ours that
  // can fold its maximum in for free (the Winograd-domain planes).  All slots of the net are zeroed at the top of forward();
  // producers of forward AND backward tensors fold into them during the step.
  float* amax = nullptr;
  size_t amax_n = 0;
  size_t reserve_slot() { const size_t off = amax_n; amax_n += AMAX_SLOT; return off; }
  // scale exponents of Winograd planes stored in pair form (ops.h wino_input_transform): one device int per plane tensor,
  // written by the transform, read by the GEMMs
  int* kscale = nullptr;
  size_t kscale_n = 0;
  size_t reserve_k() { return kscale_n++; }
  // Registry of the amax slots of whole BUFFERS (activations, keyed by Var::vbase; gradients, by the pointer the backward
  // kernel writes): every op that writes into a buffer announces itself with note_writer(base, folds), folds = it leaves max |v|
  // of everything it writes in the buffer's slot.  A consumer may take its operand's scale from the slot only if EVERY writer
  // folds (slot_if_complete, asked at run time); the slot of a concatenation buffer then bounds each of its slices.  Buffers
  // filled from outside the tape (network inputs) get an external slot from the model (set_external_slot), valid across steps.
  struct BufSlot { size_t off = 0; bool complete = true; };
  std::map<const float*, BufSlot> buf_slots;
  std::map<const float*, const float*> ext_slots;
  size_t note_writer(const float* base, bool folds) {
    auto it = buf_slots.find(base);
    if (it == buf_slots.end()) it = buf_slots.emplace(base, BufSlot{reserve_slot(), true}).first;
    if (!folds) it->second.complete = false;
    else fold_writer_ops.push_back(ops.size());      // (called while the op is being built: its index once pushed)
    return it->second.off;
  }
  // Conv + InstanceNorm fusion (modules/layers.py:12-24): a direct conv whose launch can leave the statistics' partial sums of its
  // raw output behind (ops.h conv_fwd_stat_chunk) offers them here, keyed by the output buffer; the norm_act that normalises that
  // buffer switches the offer on (`use`) and reads `partial` instead of taking a statistics pass of its own
  struct StatLink { double* partial = nullptr; int chunks = 0; std::shared_ptr<bool> use; };
  std::map<const float*, StatLink> conv_stats;
  std::vector<size_t> fold_writer_ops;      // tape positions of the folding writers: forward_from() must not skip one (its slot would stay 0)
  void set_external_slot(const float* base, const float* slot) { ext_slots[base] = slot; }
  const float* slot_if_complete(const float* base) const {
    auto e = ext_slots.find(base);
    auto it = buf_slots.find(base);
    if (e != ext_slots.end()) return (it == buf_slots.end()) ? e->second : nullptr;    // (also written on the tape: no single slot)
    if (it == buf_slots.end() || !it->second.complete || !amax) return nullptr;
    return amax + it->second.off;
  }
  // per-layer buffers for the transformed output gradient when a layer's weight gradient (side stream) and input gradient (main
  // stream) multiply by the SAME planes: transformed once on the main stream, read by both (engine.cpp shared_dy)
  bool share_dy() const;
  int dg_version = 0;       // arena.version the operands were derived from
  // set by the model for nets whose weight gradients are taken (G, the 2B discriminator instance): the
  // Winograd-transformed input of every such conv is kept from forward for its weight gradient
  bool keep_wino_inputs = false;
  // strided Winograd F(4x4,2x2) for this net's k4 s2 convs / transposed convs (engine.cpp s2_wino_wanted).  Off for the texture
  // generator: the 8-level pix2pix U-Net back-propagates through InstanceNorms over 64 ... 4 pixels, which amplify the (2.5x
  // larger) round-off of the Winograd form by an order of magnitude -- its gradients moved from 1.8e-5 to 1e-4 of the pinned
  // float64 oracle with it on (tools/bisect_run.sh); WarpModule and PatchGAN stay at 1.2e-5 / 6e-6.  SWN_WINO_S2=2 forces it.
  bool s2_wino = true;
  // Winograd scratch shared by all 3x3 layers of the net (V/M planes, dU): sized in finalize()
  size_t wsV_need = 0, wsM_need = 0, wsU_need = 0;
  float *wsV = nullptr, *wsM = nullptr, *wsU = nullptr;
  float* wsM2 = nullptr;     // dY-transform planes of the weight gradient when it runs on the side stream
  // stream for this layer's weight-gradient work: the side stream (forked now) when the context has
  // one and the net keeps its Winograd inputs (so the work touches no scratch of the main stream)
  Stream& wgrad_stream() { return (ctx.use_side() && keep_wino_inputs) ? ctx.fork_side() : ctx.s; }
  float* wgrad_planes(const Stream& sw) const { return (&sw == &ctx.side && wsM2) ? wsM2 : wsM; }
  std::vector<std::pair<Op*, size_t>> dg_layout;
  // dropout sites in forward order (one per norm_act with drop_p > 0): what swn_model_dropout_mask exports
  float* last_stats = nullptr;        // (mean, rstd) buffer of the most recently built norm_act op (op-level entry points)
  // gradient buffers whose per-image column sums the InstanceNorm backward leaves behind (ops.h NormActBwdArgs::colsum): the
  // bias gradient of the conv that produced the normalised tensor is then a sum over N values per channel
  std::map<const float*, std::pair<double*, int>> colsums;
  void bias_grad_of(Stream& s, const TView& dy, float* db) {
    auto it = colsums.find(dy.p);
    if (it != colsums.end() && it->second.second == dy.N) bias_grad_from_colsums(s, it->second.first, dy.N, dy.C, db);
    else bias_grad(s, dy, db);
  }
  struct DropSite { uint64_t salt; int N, H, W, C; float p; };
  std::vector<DropSite> drop_sites;
  static uint64_t drop_seed(uint64_t seed, uint64_t salt) { return seed * 0x9E3779B1ull + salt; }
  // piecewise-linear ops in forward order: what swn_model_act_pattern exports (ops.h act_pattern / pool_pattern).
  // kind 1 = LeakyReLU / ReLU (fused into a conv, a norm_act or standalone): sign of the output y; kind 2 = MaxPool2d(2,2)
  struct ActSite { int kind; TView y, x; };
  std::vector<ActSite> act_sites;
  void note_act(int actf, const TView& y) { if (actf == ACT_LRELU || actf == ACT_RELU) act_sites.push_back({1, y, TView()}); }

  Var alloc_var(int N, int H, int W, int C, bool need_grad);
  // layers
  // Ci = logical input channels of the reference layer (<= x.v.C, the padded buffer channels)
  // dgrad_C > 0 (k4 s2 layers reading a network input): the input gradient is formed for the first dgrad_C buffer channels only
  void conv(const std::string& name, const Var& x, const Var& y, ConvKind kind, int Ci, int Co, bool bias, int act,
            const std::vector<int32_t>* cimap = nullptr, bool x_is_input = false, int dgrad_C = 0);
  void convT(const std::string& name, const Var& x, const Var& y, int Co, bool bias);
  void norm_act(const Var& raw, const Var& y, bool norm, int act, float drop_p, const Var* residual = nullptr);
  void act(const Var& x, const Var& y, int act);
  void upsample(const Var& x, const Var& y, int factor);
  void maxpool(const Var& x, const Var& y);
  void affine(const Var& x, const Var& y, float alpha, float shift);
  // free-form op (RoIAlign gather, loss taps): bwd receives the planned accumulate flags
  void custom(const std::string& label, std::function<void(Net&)> fwd,
              std::function<void(Net&, const std::vector<int>& acc)> bwd, const std::vector<Var>& grad_targets);
  // bookkeeping
  void finalize(const std::vector<Var>& pre_initialised_grads);
  void forward();
  void forward_from(int op_begin);           // ops [op_begin, end): the caller has filled the inputs of op_begin itself
  void backward(bool wgrad, bool igrad);
  // ops [begin,end) in reverse; join = the main stream waits for the range's weight gradients (false: the caller orders them)
  void backward_range(bool wgrad, bool igrad, int op_begin, int op_end, bool join = true);
  // first op whose parameters start at or after `frac` of the arena (ops are registered in arena order)
  int split_point(double frac, size_t* arena_off) const;
  void refresh_dgrad();     // no-op when the operands are current (waits for a prefetch in flight)
  void need(Op* op);        // forward ops: wait for THIS op's share of a prefetch in flight (else refresh_dgrad())
  // with a side stream: start the refresh there (forward() calls it, so the HBM-bound re-packs /
  // filter transforms overlap the first layers); refresh_dgrad() is the wait point
  void prefetch_dgrad();
  static int prefetch_mode();
  void prefetch_issue(size_t upto);      // re-packs of the first `upto` operand-carrying ops, on the second stream
  void prefetch_finish();                // whatever the pass did not ask for
  std::vector<size_t> repack_ops_;       // ops that carry a re-pack, in op order
  size_t repack_next_ = 0;               // how many of them the current refresh has issued
  bool refresh_open_ = false;            // a refresh is being issued piecewise
  void* refresh_event = nullptr;
  bool refresh_pending = false;

 private:
  size_t reserve_dg(Op* op, size_t elems);
  bool finalized_ = false;
};

// ---- network builders (reference layouts in the .cpp) ----------------------------------
void build_warp_generator(Net& net, const Var& body, const Var& cloth, const Var& out, float dropout, int body_channels = 3,
                          int cloth_channels = 19);
// in_grad_channels > 0: only the first that many channels of the conditional input need a gradient (the generator's output)
Var build_patchgan(Net& net, const Var& x, int n_layers, const std::vector<int32_t>& cimap, int in_grad_channels = 0);
void build_texture_generator(Net& net, const Var& tex, const float* rois_dev, int num_roi, const Var& cloth_cat,
                             const Var& unet_in, const Var& out, int img_size, int cloth_channels = 19);
// img: the image buffer (>= 4 channels, RGB in the first three; 16 channels put conv1_1 on the ring kernel) -- its gradient, if any,
// is formed for the first 4 channels only
std::vector<Var> build_vgg16_slices(Net& net, const Var& img);
// buffer channels of a first-layer input with Cp (multiple of 4) channels: the next multiple of 16 when that at most doubles it
// (nets.cpp; SWN_FIRST_RING=0: Cp)
int ring_pad(int Cp);
bool first_ring_on();

// ---- gradient penalty (gp.cpp): second-order pass through PatchGAN for --gan_mode wgan-gp / dragan-gp / dragan-lp ----
class GradPenalty {
 public:
  GradPenalty(Ctx& c, ParamArena& arenaD, int B, int H, int W, int n_layers = 3);
  ~GradPenalty();
  // real / fake: the conditioned (B, H, W, 24) halves of the discriminator's input buffer.  gp_mode 1 wgan-gp,
  // 2 dragan-gp, 3 dragan-lp.  Adds grad_scale * lambda_gp * d gp / d theta to the discriminator's gradient arena and
  // writes lambda_gp * gp to loss_gp_out (device).  alpha (B floats) / beta ((B,H,W,24) view) NULL = library RNG.
  void run(const TView& real, const TView& fake, int gp_mode, float grad_scale, float lambda_gp, const float* alpha,
           const TView* beta, uint64_t seed, float* loss_gp_out);
  TView beta_buffer() const { return beta_; }
  float* alpha_buffer() const { return alpha_; }
 private:
  struct Layer {
    WShape ws{};
    size_t woff = 0, boff = (size_t)-1;
    int kind = 0;                  // 0: k4 s2 p1, 1: k4 s1 p1
    int Cin = 0, Co = 0, Cop = 0;  // input buffer channels, logical / padded output channels
    bool norm = false;
    float* stats = nullptr;
    float* dg = nullptr;           // input-gradient operand (repack_dgrad), refreshed every run
    Var raw, h;                    // conv output (l = 1..n_layers) and activation; .g = first-backward gradients
    TView u_raw, u_h, a_raw, a_h, tmp, gr0;
  };
  void conv(int l, const TView& x, const TView& y, bool bias, int act);
  void dgrad(int l, const TView& dy, const TView& dx);
  void wgrad(int l, const TView& x, const TView& dy, float* arena);
  Ctx& ctx_;
  ParamArena& A_;
  int B_;
  int nl_ = 3, NL_ = 5;                // stride-2 levels of the PatchGAN; its convs (n_layers + 2)
  int Cd_ = 24, Cd_logical_ = 22;      // D input buffer channels / the reference's channel count
 
  std::unique_ptr<Net> net_;
  std::vector<Layer> L_;
  Var xh_;
  TView u0_, beta_;
  float *gA_ = nullptr, *gB_ = nullptr, *alpha_ = nullptr, *half_std_ = nullptr, *tmp_loss_ = nullptr;
};

// ---- trainers: the fused optimize_parameters() of models/{warp,texture}_model.py -----
struct Hyper {
  float lr = 1e-4f, d_lr = 4e-4f, weight_decay = 0.f, d_weight_decay = 0.01f, b1 = 0.9f, b2 = 0.999f;
  float lambda_gan = 1.f, lambda_ce = 100.f, lambda_l1 = 10.f, lambda_content = 20.f, lambda_style = 1e-8f;
  int gan_mode = 0;          // 0 vanilla (BCE), 1 lsgan, 2 wgan
  int warp_mode_ce_only = 0; // --warp_mode ce
  float grad_scale = 1.f;    // multiplies every loss gradient (1/world_size under data parallelism)
  float d_b1 = 0.9f, d_b2 = 0.999f;   // optimizer_D's betas
  int gp_mode = 0;           // 0 none, 1 wgan-gp, 2 dragan-gp, 3 dragan-lp (modules/loss.py:133-184)
  float lambda_gp = 10.f;
};

enum LossSlot {
  L_D = 0, L_D_REAL, L_D_FAKE, L_G, L_G_GAN, L_G_CE, L_G_L1, L_G_CONTENT, L_G_STYLE, L_D_GP, L_TMP0, L_TMP1, L_TMP2, L_TMP3,
  L_TMP4, L_TMP5, L_COUNT = 32
};

class Model {
 public:
  // share != NULL: this model uses `share`'s parameter arenas -- weights, gradients, both Adam moments, step counters -- and owns
  // only its activations and derived operands: the model of another batch size (an epoch's last partial batch, batch-1 inference
  // beside training) without a second copy of the training state or copies between the two (swn_model_create_shared).  The
  // sharer must outlive this model (the C ABI's handles see to that).
  explicit Model(Model* share = nullptr)
      : share_(share), arenaG(share ? share->arenaG : ownG_), arenaD(share ? share->arenaD : ownD_) {}
  virtual ~Model();                // derived destructors run first (nets, graphs), then the buffers go
  Model* share_ = nullptr;
  ParamArena ownG_, ownD_;
  Ctx::AllocList owned_allocs;     // every device buffer this model allocated (AllocScope in its entry points)
  Ctx* ctx = nullptr;
  int B = 0, H = 0, W = 0;
  bool is_train = true;
  ParamArena &arenaG, &arenaD;
  std::unique_ptr<Net> G, D2, D1;
  float* losses = nullptr;       // device [L_COUNT]
  Hyper hyper;
  virtual void set_input(int slot, const float* dev_nchw, int N, int C, int Hh, int Ww) = 0;
  virtual void set_input_labels(int slot, const int32_t* dev_labels, int N, int Hh, int Ww) = 0;
  virtual void get_output(int slot, float* dev_nchw) = 0;
  virtual TView output_view() = 0;              // NHWC view of self.fakes (logical channels: cloth_channels warp / 3 texture)
  virtual int output_channels() const = 0;
  virtual void forward(bool training, uint64_t seed) = 0;
  virtual void backward_D(float label_fake, float label_real) = 0;
  virtual void backward_G(float label_real) = 0;
  // backward_G in two parts for gradient-exchange overlap: part 0 = everything down to the split
  // op (its arena range [off, n) is final on return), part 1 = the rest ([0, off)).
  virtual void backward_G_head(float label_real) = 0;
  int backward_G_parts() const;
  void backward_G_part(float label_real, int part, size_t* ready_off, size_t* ready_count, bool join = true);
  void optimizer_step(int net);
  // AdamW on the arena range [off, off + count) only (data parallel: a bucket is stepped as soon as its all-reduce
  // has landed, under the back-propagation of the next bucket).  first != 0 opens a new optimizer step (advances
  // the bias-correction counter); the ranges of one step must tile the arena.
  void optimizer_step_range(int net, size_t off, size_t count, int first);
  void optimizer_step_range_on(Stream& s, int net, size_t off, size_t count, int first);
  // generator backward pass with each bucket's AdamW enqueued on the side stream behind the bucket's weight gradients, under the
  // back-propagation of the earlier layers on the main stream (Model::step with a side stream; SWN_STREAM_ADAMW=0: one launch
  // over the whole arena after the pass).  Same arithmetic per element: results are bit-identical.
  void backward_G_streamed(float label_real);
  void step(const float labels[3], bool training, uint64_t seed);
  // The data-parallel step with the exchange owned by the library (Ctx::attach_comm): forward; backward_D, all-reduce of D's
  // gradient arena, optimizer_D; backward_G bucket by bucket (backward_G_part), each bucket's all-reduce on the exchange
  // stream as soon as its gradients are final, its AdamW enqueued on the SAME stream behind the all-reduce -- the main
  // stream never waits for a collective before the end of the step, where it joins the exchange stream.  Every loss gradient
  // is pre-scaled by hyper.grad_scale = 1 / world, so SUM is the mean.  after_forward: the caller has run forward() already
  // (texture stage with the style term: the global-batch style context is set between the two).  With one rank attached the
  // collectives are identities and the result equals step()'s.
  void step_dp(const float labels[3], bool training, uint64_t seed, bool after_forward);
  // The same step recorded ONCE into a hipGraph (per value of `training`) and replayed: every per-step scalar -- the three
  // smooth labels (modules/loss.py:77-104), the dropout seed, the bias corrections of both AdamW steps -- lives in a small
  // device block (StepParams) that is uploaded in stream order before each launch, so the recorded launch sequence (two
  // streams, ~620 kernels) is identical for every step.  train.py:74's loss read-back stays the only synchronisation.
  // First call per mode runs eagerly (kernel attributes, lazy buffers), the second records, later ones replay.  Not for the
  // gradient-penalty modes (their alpha / beta draws are host-seeded per step) nor under data parallelism (the exchange
  // runs through torch.distributed between the phases): those keep step().  Results are bit-identical to step().
  void step_captured(const float labels[3], bool training, uint64_t seed);
  struct StepParams { float labels[4]; uint64_t seed; float schedG[2]; float schedD[2]; };
  StepParams* sp_dev = nullptr;
  bool indirect = false;          // the phases read labels / seed / AdamW schedule from sp_dev
  const float* label_dev(int i) const { return indirect ? reinterpret_cast<const float*>(sp_dev) + i : nullptr; }
  void invalidate_step_graphs() { for (void*& g : step_graph_) { graph_destroy(g); g = nullptr; } }
  void* step_graph_[2] = {nullptr, nullptr};
  int step_graph_key_[2] = {0, 0};      // (two streams?, AdamW placement) the graph was recorded under
  int step_warm_[2] = {0, 0};
  void* cap_stream_ = nullptr;
  // NLayerDiscriminator.forward (modules/discriminators.py:134-136) as a standalone call: x = the conditioned input
  // in the REFERENCE's channel order (B, 22, H, W) NCHW on the device, pred = (B, 1, H/8-2, W/8-2).  Runs on a
  // private PatchGAN instance bound to the same weights (created on first use), so the model's own buffers
  // (self.fakes, the conditioned batch) are left untouched.
  void discriminate(const float* x_nchw, float* pred_nchw);
  // PerceptualLoss.forward (modules/losses/perceptual.py:49-66) as a standalone call (texture model): writes
  // (content, style) = (sum_k MSE of the normalised VGG slice features, 5 x MSE of the image Grams) to the two
  // device floats at `out2`; d_output (optional, NCHW) receives content_w * d(content) + style_w * d(style).
  virtual void perceptual(const float* output_nchw, const float* target_nchw, int use_style, float* out2,
                          float content_w, float style_w, float* d_output_nchw);
 public:
  // one-shot random inputs of the next gradient-penalty pass (host-provided for seeded parity; see capi)
  void set_gp_random(const float* alpha_dev, const float* beta_nchw_dev);
  virtual bool supports_gradient_penalty() const { return false; }
  // Data parallelism, texture stage: the style term's image Gram couples ALL samples of the global batch
  // (modules/losses/perceptual.py:6-10,58-63).  The host all-gathers the generated and target images of every rank
  // and hands them over ((n_total, 3, H, W) NCHW, this rank's samples start at n0); the next backward_G then evaluates
  // the Gram over the global batch and back-propagates into the local samples -- identical to the one-process
  // big-batch step.  One-shot (consumed by the next backward_G).
  virtual void set_style_context(const float* all_out_nchw, const float* all_tgt_nchw, int n_total, int n0);
 protected:
  std::unique_ptr<GradPenalty> gp_;
  bool gp_alpha_set_ = false, gp_beta_set_ = false;
  void run_gradient_penalty(const TView& real, const TView& fake);   // called by backward_D after D2's backward
  std::unique_ptr<Net> D3_;       // discriminate(): own input buffer + activations, shared (frozen) arenaD
  Var d3_in_, d3_pred_;
  std::vector<int32_t> d_cimap_;  // buffer channel -> reference channel of the conditional D input (set by the model)
  int d_layers_ = 3;              // PatchGAN depth the model was built with (Ctx::patchgan_layers at construction)
 public:
  int patchgan_layers() const { return d_layers_; }
  ParamArena& arena(int net) { return net == 0 ? arenaG : arenaD; }
  virtual ParamArena* arena_ptr(int net) {
    if (net == 0) return &arenaG;
    if (net == 1 && is_train) return &arenaD;
    return nullptr;
  }
  Net* net_for_taps(int net) { return net == 0 ? G.get() : D2.get(); }
  // swn_model_act_pattern: 0 = generator, 1 = discriminator of the D step (batch [fake | real]), 2 = discriminator as
  // re-evaluated in the G step (fake only, updated weights), 3 = VGG16 on the generated image (texture model)
  virtual Net* net_for_patterns(int net) { return net == 0 ? G.get() : net == 1 ? D2.get() : net == 2 ? D1.get() : nullptr; }
};

// ---------------------------------------------------------------------------------------------------------------
// inference.py's two stages (inference.py:94-126,140-149,169-180) as one device-resident sequence:
//   warp forward -> argmax over the 19 cloth channels (what compress_and_save_cloth stores, data_utils.py:322) ->
//   one-hot expansion straight into the texture model's cloth inputs (to_onehot_tensor, data_utils.py:330-343) ->
//   texture forward.
// The sequence has no host decisions in it, so it is captured once into a hipGraph and replayed: at batch size 1
// the ~150 launches are otherwise launch-latency bound.
// ---------------------------------------------------------------------------------------------------------------
class Pipeline {
 public:
  Pipeline(Model& warp, Model& texture);
  ~Pipeline();
  void run(bool use_graph);
  int32_t* labels() const { return labels_; }      // device (B, H, W): the hand-off label map of the last run
  bool graph_captured() const { return exec_ != nullptr; }
 private:
  void enqueue();
  Model& warp_;
  Model& tex_;
  int32_t* labels_ = nullptr;
  Ctx::AllocList owned_;
  Ctx* ctx_ = nullptr;
  void* exec_ = nullptr;
  void* cap_stream_ = nullptr;     // capture happens on a private stream: the caller's may be the legacy default stream,
                                   // which cannot be captured; the instantiated graph is launched on the caller's stream
  bool warmed_ = false;
};

// body_channels / cloth_channels: --body_representation / --cloth_representation / --body_channels / --cloth_channels
// of the reference (models/warp_model.py:49-55, options/base_options.py:75-105); defaults 3 (rgb) / 19 (labels)
Model* create_warp_model(Ctx& ctx, int B, int H, int W, bool is_train, float dropout, int body_channels = 3, int cloth_channels = 19,
                         Model* share = nullptr);
Model* create_texture_model(Ctx& ctx, int B, int H, int W, bool is_train, int num_roi, int cloth_channels = 19, Model* share = nullptr);

}  // namespace swn
