// swapnet_amd -- network builders and the warp-stage model.
#include <cmath>
#include <cstdlib>

#include "engine.h"

namespace swn {

// ---------------------------------------------------------------------------------------
// WarpModule (modules/swapnet_modules.py:22-151).  Channel concatenations are buffers that
// the producers write into directly (zero-copy torch.cat):
//   cat3 [B,H/2 ,W/2 ,192] = dual_u3(64)  | body_d1(64)  | cloth_d1(64)     (:151 input of upsample_and_pad)
//   cat2 [B,H/4 ,W/4 ,384] = dual_u2(128) | body_d2(128) | cloth_d2(128)
//   cat1 [B,H/8 ,W/8 ,768] = dual_u1(256) | body_d3(256) | cloth_d3(256)
//   bc   [B,H/16,W/16,1024]= body_d4(512) | cloth_u2(512)                    (:131)
// ---------------------------------------------------------------------------------------
static void down_block(Net& n, const std::string& name, const Var& x, const Var& y, int Ci, int Co, bool normalize,
                       float dropout, bool x_is_input = false) {
  // UNetDown (modules/layers.py:12-24): Conv k4s2p1 (no bias) -> [IN] -> LeakyReLU(0.2) -> [Dropout]
  if (!normalize && dropout == 0.f) {
    n.conv(name, x, y, CK_K4S2, Ci, Co, false, ACT_LRELU, nullptr, x_is_input);
  } else {
    Var raw = n.alloc_var(x.v.N, x.v.H / 2, x.v.W / 2, round_up(Co, 4), true);
    n.conv(name, x, raw, CK_K4S2, Ci, Co, false, ACT_NONE, nullptr, x_is_input);
    n.norm_act(raw, y, normalize, ACT_LRELU, dropout);
  }
}
static void up_block(Net& n, const std::string& name, const Var& x, const Var& y, int Co, bool bias, float dropout) {
  // UNetUp (modules/layers.py:27-44): ConvT k4s2p1 -> IN -> ReLU -> [Dropout]
  Var raw = n.alloc_var(x.v.N, x.v.H * 2, x.v.W * 2, round_up(Co, 4), true);
  n.convT(name, x, raw, Co, bias);
  n.norm_act(raw, y, true, ACT_RELU, dropout);
}

void build_warp_generator(Net& n, const Var& body, const Var& cloth, const Var& out, float dropout, int Cb, int Cc) {
  const int B = body.v.N, H = body.v.H, W = body.v.W;
  if (H % 64 || W % 64) throw Error(1, "WarpModule needs H and W to be multiples of 64");
  Var cat3 = n.alloc_var(B, H / 2, W / 2, 192, true);
  Var cat2 = n.alloc_var(B, H / 4, W / 4, 384, true);
  Var cat1 = n.alloc_var(B, H / 8, W / 8, 768, true);
  Var bc = n.alloc_var(B, H / 16, W / 16, 1024, true);
  Var body_d1 = cat3.slice(64, 64), cloth_d1 = cat3.slice(128, 64);
  Var body_d2 = cat2.slice(128, 128), cloth_d2 = cat2.slice(256, 128);
  Var body_d3 = cat1.slice(256, 256), cloth_d3 = cat1.slice(512, 256);
  Var body_d4 = bc.slice(0, 512), cloth_u2 = bc.slice(512, 512);
  // body encoder (:34-37)
  down_block(n, "body_down1.model.0", body, body_d1, Cb, 64, false, 0.f, true);
  down_block(n, "body_down2.model.0", body_d1, body_d2, 64, 128, true, 0.f);
  down_block(n, "body_down3.model.0", body_d2, body_d3, 128, 256, true, 0.f);
  down_block(n, "body_down4.model.0", body_d3, body_d4, 256, 512, true, dropout);
  // cloth encoder (:42-51)
  Var cloth_d4 = n.alloc_var(B, H / 16, W / 16, 512, true);
  Var cloth_d5 = n.alloc_var(B, H / 32, W / 32, 1024, true);
  Var cloth_d6 = n.alloc_var(B, H / 64, W / 64, 1024, true);
  Var cloth_u1 = n.alloc_var(B, H / 32, W / 32, 1024, true);
  down_block(n, "cloth_down1.model.0", cloth, cloth_d1, Cc, 64, false, 0.f, true);
  down_block(n, "cloth_down2.model.0", cloth_d1, cloth_d2, 64, 128, true, 0.f);
  down_block(n, "cloth_down3.model.0", cloth_d2, cloth_d3, 128, 256, true, 0.f);
  down_block(n, "cloth_down4.model.0", cloth_d3, cloth_d4, 256, 512, true, 0.f);
  down_block(n, "cloth_down5.model.0", cloth_d4, cloth_d5, 512, 1024, true, dropout);
  down_block(n, "cloth_down6.model.0", cloth_d5, cloth_d6, 1024, 1024, false, dropout);
  up_block(n, "cloth_up1.model.0", cloth_d6, cloth_u1, 1024, false, 0.f);
  up_block(n, "cloth_up2.model.0", cloth_u1, cloth_u2, 512, false, 0.f);
  // residual blocks (:56-62; modules/layers.py:126-144)
  Var r = bc;
  for (int i = 0; i < 4; ++i) {
    const std::string p = "resblocks." + std::to_string(i) + ".conv_block.";
    Var raw1 = n.alloc_var(B, H / 16, W / 16, 1024, true);
    Var h = n.alloc_var(B, H / 16, W / 16, 1024, true);
    Var raw2 = n.alloc_var(B, H / 16, W / 16, 1024, true);
    Var rn = n.alloc_var(B, H / 16, W / 16, 1024, true);
    n.conv(p + "1", r, raw1, CK_K3S1_REFLECT, 1024, 1024, true, ACT_NONE);
    n.norm_act(raw1, h, true, ACT_RELU, dropout);
    n.conv(p + "6", h, raw2, CK_K3S1_REFLECT, 1024, 1024, true, ACT_NONE);
    n.norm_act(raw2, rn, true, ACT_NONE, 0.f, &r);
    n.taps["res" + std::to_string(i)] = rn;
    n.taps["res" + std::to_string(i) + "_h"] = h;
    r = rn;
  }
  // dual decoder (:72-76)
  up_block(n, "dual_up1.model.0", r, cat1.slice(0, 256), 256, false, 0.f);
  up_block(n, "dual_up2.model.0", cat1, cat2.slice(0, 128), 128, false, 0.f);
  up_block(n, "dual_up3.model.0", cat2, cat3.slice(0, 64), 64, false, 0.f);
  // upsample_and_pad (:85-90): Upsample x2 + ZeroPad(1,0,1,0) + Conv k4 p1 (bias) + Tanh
  n.conv("upsample_and_pad.2", cat3, out, CK_TAIL_UP, 192, Cc, true, ACT_TANH);
  n.taps["body_d1"] = body_d1; n.taps["body_d2"] = body_d2; n.taps["body_d3"] = body_d3; n.taps["body_d4"] = body_d4;
  n.taps["cloth_d1"] = cloth_d1; n.taps["cloth_d2"] = cloth_d2; n.taps["cloth_d3"] = cloth_d3;
  n.taps["cloth_d4"] = cloth_d4; n.taps["cloth_d5"] = cloth_d5; n.taps["cloth_d6"] = cloth_d6;
  n.taps["cloth_u1"] = cloth_u1; n.taps["cloth_u2"] = cloth_u2;
  n.taps["dual_u1"] = cat1; n.taps["dual_u2"] = cat2; n.taps["dual_u3"] = cat3;
  n.taps["fakes"] = out;
}

// ---------------------------------------------------------------------------------------
// NLayerDiscriminator under instance norm (modules/discriminators.py:91-136): all convs
// carry a bias (:103-106).  Returns the 1-channel prediction map (C padded to 4).
// ---------------------------------------------------------------------------------------
Var build_patchgan(Net& n, const Var& x, int n_layers, const std::vector<int32_t>& cimap, int in_grad_channels) {
  const int N = x.v.N;
  int ci = 0;
  for (int v : cimap) ci += v >= 0;
  const int ndf = 64;
  if (n_layers == 0) {
    // PixelDiscriminator under instance norm (modules/discriminators.py:139-170; --discriminator pixel, models/base_gan.py:61-65): three
    // 1x1 convs, all with a bias (use_bias is true under InstanceNorm2d, :152-155), a prediction per PIXEL.  state_dict keys net.{0,2,5}.
    Var a = n.alloc_var(N, x.v.H, x.v.W, ndf, true);
    n.conv("net.0", x, a, CK_K1S1, ci, ndf, true, ACT_LRELU, &cimap, true, in_grad_channels);          // :158-159
    n.taps["d0"] = a;
    Var raw = n.alloc_var(N, x.v.H, x.v.W, ndf * 2, true);
    Var act = n.alloc_var(N, x.v.H, x.v.W, ndf * 2, true);
    n.conv("net.2", a, raw, CK_K1S1, ndf, ndf * 2, true, ACT_NONE);                                   // :160
    n.norm_act(raw, act, true, ACT_LRELU, 0.f);                                                        // :161-162
    n.taps["d1"] = act;
    Var pred = n.alloc_var(N, x.v.H, x.v.W, 4, true);
    n.conv("net.5", act, pred, CK_K1S1, ndf * 2, 1, true, ACT_NONE);                                   // :163
    n.taps["pred"] = pred;
    return pred;
  }
  if (n_layers < 1 || n_layers > 5) throw Error(1, "PatchGAN: n_layers_D must be in [1, 5] (0 = the 1x1 PixelDiscriminator)");
  if ((x.v.H >> n_layers) < 3 || (x.v.W >> n_layers) < 3)
    throw Error(1, "PatchGAN: the input is too small for " + std::to_string(n_layers) + " stride-2 levels followed by two 4x4 stride-1 convs");
  int H = x.v.H / 2, W = x.v.W / 2;
  Var a = n.alloc_var(N, H, W, ndf, true);
  n.conv("model.0", x, a, CK_K4S2, ci, ndf, true, ACT_LRELU, &cimap, true, in_grad_channels);       // :110
  n.taps["d0"] = a;
  int idx = 2, mult = 1;
  for (int l = 1; l < n_layers; ++l) {                                            // :113-120
    const int prev = mult;
    mult = std::min(1 << l, 8);
    H /= 2; W /= 2;
    Var raw = n.alloc_var(N, H, W, ndf * mult, true);
    Var act = n.alloc_var(N, H, W, ndf * mult, true);
    n.conv("model." + std::to_string(idx), a, raw, CK_K4S2, ndf * prev, ndf * mult, true, ACT_NONE);
    n.norm_act(raw, act, true, ACT_LRELU, 0.f);
    n.taps["d" + std::to_string(l)] = act;
    a = act; idx += 3;
  }
  const int prev = mult;
  mult = std::min(1 << n_layers, 8);
  H -= 1; W -= 1;
  Var raw = n.alloc_var(N, H, W, ndf * mult, true);
  Var act = n.alloc_var(N, H, W, ndf * mult, true);
  n.conv("model." + std::to_string(idx), a, raw, CK_K4S1, ndf * prev, ndf * mult, true, ACT_NONE);   // :124-128
  n.norm_act(raw, act, true, ACT_LRELU, 0.f);
  n.taps["d" + std::to_string(n_layers)] = act;
  idx += 3;
  H -= 1; W -= 1;
  Var pred = n.alloc_var(N, H, W, 4, true);
  n.conv("model." + std::to_string(idx), act, pred, CK_K4S1, ndf * mult, 1, true, ACT_NONE);         // :131
  n.taps["pred"] = pred;
  return pred;
}

// ---------------------------------------------------------------------------------------
// shared GAN plumbing
// ---------------------------------------------------------------------------------------
static void gan_loss_op(Stream& s, int mode, const TView& pred, float label, bool target_is_real, float gscale,
                        float* out, const TView* dpred, const float* label_dev = nullptr) {
  // GANLoss.__call__ (modules/loss.py:110-130)
  if (mode == 0) bce_logits_loss(s, pred, label, gscale, out, dpred, label_dev);
  else if (mode == 1) lsgan_loss(s, pred, label, gscale, out, dpred, label_dev);
  else wgan_loss(s, pred, target_is_real ? -1.f : 1.f, gscale, out, dpred);
}

// ---------------------------------------------------------------------------------------
// WarpModel (models/warp_model.py).  D is conditional on cat(bodys, cloth) (:115,119,157);
// the D input buffer orders it [cloth(19)+0 | body(3)+0] so the generator's tanh output and
// the CE logits live in an aligned 20-channel slice.  Dx holds 2B images: [0,B) conditioned
// fakes, [B,2B) conditioned targets -- the two D passes of backward_D run as one 2B batch
// (InstanceNorm is per-sample, so batching is exact).
// ---------------------------------------------------------------------------------------
// First-layer buffers on the ring kernel (round 4).  The convs that read network inputs (cloth_down1: 19 -> 64, PatchGAN model.0:
// 22 -> 64) ran on the register-staged f32-MFMA kernels because the LDS-DMA ring kernels walk the channels of a tap in 16-channel
// stages; their NHWC input buffers are now allocated with the channel count rounded up to a multiple of 16 where that costs at
// most 2x the channels (20 -> 32, 24 -> 32; the 3-channel body stays at 4: 16 would be 4x the work of a layer that is small
// anyway).  Pad channels hold zeros (allocation zero-fills, nothing writes them) and meet zero weight rows.  SWN_FIRST_RING=0
// keeps the round-3 layout (read when a model is built).
bool first_ring_on() {
  static const bool on = true;        // (SWN_FIRST_RING=0, the round-4 A/B switch, is gone: +0.6 ms/step without it)
  return on;
}
int ring_pad(int Cp) {
  const int r = round_up(Cp, 16);
  return (first_ring_on() && r <= 2 * Cp) ? r : Cp;
}

class WarpModel final : public Model {
 public:
  Var body, cloth, Dx, pred2, pred1;
  float dropout = 0.5f;
  int Cb = 3, Cc = 19, Cbp = 4, Ccp = 20;      // logical / padded channel counts of the body and cloth representations
  int CbB = 4, CcB = 20, CdB = 24;             // buffer channels of the body, cloth and conditional-D inputs (ring_pad)
  // amax slots of the buffers filled from outside the tape (engine.h ext_slots): taken when an input is handed over.  The
  // conditional-D buffer also receives the generator's tanh output each step: its slot is floored at 1 = sup |tanh|.
  float *slot_body = nullptr, *slot_cloth = nullptr, *slot_dx = nullptr;
  bool ce_first_ = false, ce_done_ = false;      // CE gradient written first, PatchGAN input gradient accumulated (see the constructor)
  float ce_scale_ = 0.f;

  WarpModel(Ctx& c, int B_, int H_, int W_, bool train, float drop, int body_channels, int cloth_channels, Model* share = nullptr)
      : Model(share) {
    ctx = &c; B = B_; H = H_; W = W_; is_train = train; dropout = drop;
    Cb = body_channels; Cc = cloth_channels; Cbp = round_up(Cb, 4); Ccp = round_up(Cc, 4);
    if (Cb < 1 || Cc < 2 || Cb > 64 || Cc > 64) throw Error(1, "WarpModel: body_channels in [1,64], cloth_channels in [2,64]");
    CbB = ring_pad(Cbp); CcB = ring_pad(Ccp); CdB = ring_pad(Ccp + Cbp);
    AllocScope mine(c, owned_allocs);
    G = std::make_unique<Net>(c, arenaG);
    G->keep_wino_inputs = train;
    body = G->alloc_var(B, H, W, CbB, false);
    cloth = G->alloc_var(B, H, W, CcB, false);
    Dx = G->alloc_var(train ? 2 * B : B, H, W, CdB, train);
    float* slots = static_cast<float*>(c.alloc(3 * AMAX_SLOT * sizeof(float)));
    slot_body = slots; slot_cloth = slots + AMAX_SLOT; slot_dx = slots + 2 * AMAX_SLOT;
    G->set_external_slot(body.vbase, slot_body);
    G->set_external_slot(cloth.vbase, slot_cloth);
    Var fake_slot = Dx.batch(0, B).slice(0, Ccp);
    build_warp_generator(*G, body, cloth, fake_slot, dropout, Cb, Cc);
    if (!arenaG.frozen) arenaG.allocate(c);          // (a sharing model binds the sharer's frozen arena: shapes were checked)
    G->finalize({fake_slot});
    losses = static_cast<float*>(c.alloc(L_COUNT * sizeof(float)));
    if (train) {
      std::vector<int32_t> cimap(CdB, -1);
      for (int i = 0; i < Cc; ++i) cimap[i] = Cb + i;     // cloth channels follow the body channels (warp_model.py:115)
      for (int i = 0; i < Cb; ++i) cimap[Ccp + i] = i;
      d_cimap_ = cimap; d_layers_ = c.patchgan_layers;
      D2 = std::make_unique<Net>(c, arenaD);
      D2->keep_wino_inputs = true;
      D2->set_external_slot(Dx.vbase, slot_dx);
      pred2 = build_patchgan(*D2, Dx, c.patchgan_layers, cimap);
      if (!arenaD.frozen) arenaD.allocate(c);
      D2->finalize({pred2});
      // second instance over the first B images, bound to the same (now frozen) arena
      D1 = std::make_unique<Net>(c, arenaD);
      D1->set_external_slot(Dx.vbase, slot_dx);
      pred1 = build_patchgan(*D1, Dx.batch(0, B), c.patchgan_layers, cimap, Ccp);      // d(fakes) only: the condition is data
      // Cross-entropy term first (round 6).  d loss_G / d fakes = the PatchGAN pass's input gradient + the CE gradient; the CE term
      // needs neither network's update, so it is taken EARLY -- behind the discriminator's backward pass, where the main stream
      // otherwise waits 0.6 ms for the first layer's weight gradient and AdamW(D) on the second stream -- and writes the buffer; the
      // PatchGAN's input gradient then accumulates into it (the planner is told the range is pre-written).  a + b = b + a: the sum is
      // bit-identical to the old order (GAN term written, CE accumulated).  Only where the first layer's gradient covers exactly
      // the generator's channels (the narrow input gradient); SWN_CE_EARLY=0 keeps the old order (read when a model is built).
      const bool ce_early_wanted = !(getenv("SWN_CE_EARLY") && atoi(getenv("SWN_CE_EARLY")) == 0);
      ce_first_ = false;
      if (ce_early_wanted)
        for (auto& op : D1->ops)
          if (!op->grad_targets.empty() && op->grad_targets[0].gbase == Dx.gbase) {
            ce_first_ = op->grad_targets.size() == 1 && op->grad_targets[0].g.C == Ccp && op->grad_targets[0].g.p == Dx.batch(0, B).g.p;
            break;
          }
      if (ce_first_) D1->finalize({pred1, Dx.batch(0, B).slice(0, Ccp)});
      else D1->finalize({pred1});
    }
  }
  void refresh_input_slots(int slot) {
    Stream& s = ctx->s;
    if (slot == 0) tensor_amax(s, body.v, slot_body);
    if (slot == 1) tensor_amax(s, cloth.v, slot_cloth);
    if (slot != 1) tensor_amax(s, Dx.v, slot_dx, 1.0f);       // bodys and targets live in it; the fakes are bounded by 1
  }
  void set_input(int slot, const float* src, int N, int C, int Hh, int Ww) override {
    if (N != B || Hh != H || Ww != W) throw Error(1, "set_input: shape mismatch with the model's (B,H,W)");
    Stream& s = ctx->s;
    if (slot == 0) {            // bodys (B,Cb,H,W)
      if (C != Cb) throw Error(1, "bodys must have " + std::to_string(Cb) + " channels");
      nchw_to_nhwc(s, src, N, C, H, W, body.v);
      nchw_to_nhwc(s, src, N, C, H, W, Dx.batch(0, B).v.slice(Ccp, Cbp));
      if (is_train) nchw_to_nhwc(s, src, N, C, H, W, Dx.batch(B, B).v.slice(Ccp, Cbp));
    } else if (slot == 1) {     // input_cloths (B,Cc,H,W)
      if (C != Cc) throw Error(1, "input_cloths must have " + std::to_string(Cc) + " channels");
      nchw_to_nhwc(s, src, N, C, H, W, cloth.v);
    } else if (slot == 2) {     // target_cloths
      if (!is_train) throw Error(1, "targets are only used in training");
      if (C != Cc) throw Error(1, "target_cloths must have " + std::to_string(Cc) + " channels");
      nchw_to_nhwc(s, src, N, C, H, W, Dx.batch(B, B).v.slice(0, Ccp));
    } else {
      throw Error(1, "set_input: unknown slot");
    }
    refresh_input_slots(slot);
  }
  void set_input_labels(int slot, const int32_t* lab, int N, int Hh, int Ww) override {
    if (N != B || Hh != H || Ww != W) throw Error(1, "set_input_labels: shape mismatch with the model's (B,H,W)");
    if (slot == 1) labels_to_onehot(ctx->s, lab, cloth.v, Cc);
    else if (slot == 2 && is_train) labels_to_onehot(ctx->s, lab, Dx.batch(B, B).v.slice(0, Ccp), Cc);
    else throw Error(1, "set_input_labels: slot has no label form");
    refresh_input_slots(slot);
  }
  void get_output(int slot, float* dst) override {
    if (slot != 0) throw Error(1, "get_output: unknown slot");
    nhwc_to_nchw(ctx->s, Dx.batch(0, B).v.slice(0, Ccp), dst, Cc);
  }
  TView output_view() override { return Dx.batch(0, B).v.slice(0, Ccp); }
  int output_channels() const override { return Cc; }
  bool supports_gradient_penalty() const override { return d_layers_ >= 1; }      // (the reverse-over-reverse pass of gp.cpp walks the PatchGAN)
  void forward(bool training, uint64_t seed) override {       // warp_model.py:106-107
    G->training = training; G->seed = seed;
    G->forward();
    ce_done_ = false;
  }
  // loss_G's cross-entropy term (warp_model.py:150-156): value into the loss slots, gradient WRITTEN to d(fakes)
  void ce_term() {
    Stream& s = ctx->s;
    TView dfakes = Dx.batch(0, B).g.slice(0, Ccp);
    ce_scale_ = hyper.lambda_ce * hyper.grad_scale;
    ce_argmax_loss(s, Dx.batch(0, B).v.slice(0, Ccp), Dx.batch(B, B).v.slice(0, Ccp), Cc, ce_scale_, losses + L_TMP1, &dfakes, 0);
    ce_done_ = true;
  }
  void backward_D(float label_fake, float label_real) override {     // warp_model.py:109-139
    Stream& s = ctx->s;
    D2->training = false;
    D2->refresh_dgrad();
    D2->forward();
    TView pf = pred2.batch(0, B).v, pr = pred2.batch(B, B).v;
    TView gf = pred2.batch(0, B).g, gr = pred2.batch(B, B).g;
    // loss_D = 0.5 * (loss_D_fake + loss_D_real); lambda_discriminator is ignored here (:123)
    gan_loss_op(s, hyper.gan_mode, pf, label_fake, false, 0.5f * hyper.grad_scale, losses + L_D_FAKE, &gf, label_dev(0));
    gan_loss_op(s, hyper.gan_mode, pr, label_real, true, 0.5f * hyper.grad_scale, losses + L_D_REAL, &gr, label_dev(1));
    scalar_axpby(s, losses + L_D_FAKE, 0.5f, losses + L_D_REAL, 0.5f, losses + L_D);
    D2->backward_range(true, false, 0, (int)D2->ops.size(), /*join=*/false);
    if (ce_first_ && !hyper.warp_mode_ce_only) ce_term();      // under the tail of the weight gradients on the second stream
    ctx->join_side();                                          // every D gradient is final for whatever the main stream does next
    if (hyper.gp_mode) run_gradient_penalty(Dx.batch(B, B).v, Dx.batch(0, B).v);     // warp_model.py:126-136
    else dev_memset(s, losses + L_D_GP, 0, sizeof(float));
  }
  void backward_G(float label_real) override {                        // warp_model.py:141-167
    backward_G_head(label_real);
    G->refresh_dgrad();
    G->backward(true, false);
  }
  void backward_G_head(float label_real) override {
    Stream& s = ctx->s;
    TView fakes = Dx.batch(0, B).v.slice(0, Ccp);
    TView dfakes = Dx.batch(0, B).g.slice(0, Ccp);
    TView targets = Dx.batch(B, B).v.slice(0, Ccp);
    if (!hyper.warp_mode_ce_only) {
      // (the early CE term stands if it was taken for this forward pass under the weights in force now)
      if (ce_first_ && !(ce_done_ && ce_scale_ == hyper.lambda_ce * hyper.grad_scale)) ce_term();
      D1->refresh_dgrad();
      D1->training = false;
      D1->forward();                                   // D was just updated (base_gan.py:199)
      gan_loss_op(s, hyper.gan_mode, pred1.v, label_real, true, hyper.lambda_gan * hyper.grad_scale, losses + L_TMP0, &pred1.g, label_dev(2));
      scalar_axpby(s, losses + L_TMP0, hyper.lambda_gan, nullptr, 0.f, losses + L_G_GAN);
      D1->backward(false, true);                       // D weight grads would be discarded (quirk 5); accumulates onto the CE term if that went first
      if (!ce_first_) ce_argmax_loss(s, fakes, targets, Cc, hyper.lambda_ce * hyper.grad_scale, losses + L_TMP1, &dfakes, 1);
      ce_done_ = false;                                // consumed: a second backward_G on the same forward pass takes it again
    } else {
      dev_memset(s, losses + L_G_GAN, 0, sizeof(float));
      ce_argmax_loss(s, fakes, targets, Cc, hyper.lambda_ce * hyper.grad_scale, losses + L_TMP1, &dfakes, 0);
    }
    scalar_axpby(s, losses + L_TMP1, hyper.lambda_ce, nullptr, 0.f, losses + L_G_CE);
    scalar_axpby(s, losses + L_G_GAN, 1.f, losses + L_G_CE, 1.f, losses + L_G);
  }
};

Model* create_warp_model(Ctx& ctx, int B, int H, int W, bool is_train, float dropout, int body_channels, int cloth_channels, Model* share) {
  if (share && !dynamic_cast<WarpModel*>(share)) throw Error(1, "shared model: the sharer is not a warp model");
  return new WarpModel(ctx, B, H, W, is_train, dropout, body_channels, cloth_channels, share);
}

}  // namespace swn
