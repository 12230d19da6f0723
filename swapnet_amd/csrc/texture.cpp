// swapnet_amd -- texture stage: TextureModule generator, VGG16 perceptual network and the
// TextureModel training step (reference: modules/swapnet_modules.py:154-260,
// modules/pix2pix_modules.py:113-262, modules/losses/perceptual.py, models/texture_model.py).
#include <cstdlib>
#include <cmath>

#include "engine.h"

namespace swn {

// ---------------------------------------------------------------------------------------
// TextureModule.forward (swapnet_modules.py:231-260):
//   rois -> RoIAlign(128x128) -> view(B, 36, 128, 128) -> UNetDown(36,36) -> nearest x(H/64)
//   -> cat cloth (36 + 19 = 55 channels, buffer padded to 56) -> UnetGenerator(55 -> 3).
// UnetGenerator under instance norm (pix2pix_modules.py:113-262).  With e_d the tensor entering
// block d (d = 1 .. depth-1, depth = num_downs), the in-place LeakyReLU (:220) makes every skip
// LeakyReLU(e_d), so block d returns C_d = [ l_d | u_d ] with l_d = lrelu(e_d):
//   e_1 = conv0(x)                              outermost down conv, bias, no norm (:225-231)
//   e_{d+1} = IN(conv_d(l_d))                   (:247-249)   [innermost: no norm, :232-238]
//   u_d = IN(convT_d(relu(C_{d+1})))  [+Dropout(0.5) for 4 <= d < depth-1]   (:239-254)
//   out = tanh(convT_0(relu(C_1)))              (:226-231)
// ---------------------------------------------------------------------------------------
static std::string unet_prefix(int d) {
  std::string p = "unet.model";
  for (int j = 0; j < d; ++j) p += j == 0 ? ".model.1" : ".model.3";
  return p;
}

void build_texture_generator(Net& n, const Var& tex, const float* rois_dev, int num_roi, const Var& cloth_slot,
                             const Var& unet_in, const Var& out, int img_size, int cloth_channels) {
  const int B = tex.v.N, H = tex.v.H, W = tex.v.W;
  const int depth = (int)std::lround(std::log2((double)img_size));
  if ((1 << depth) != img_size || H != img_size || W != img_size || depth < 6)
    throw Error(1, "TextureModule: crop_size must be a power of two >= 64 (U-Net depth = log2(size))");
  (void)cloth_slot;
  const int RC = num_roi * 3;                       // 36
  if (RC % 4) throw Error(1, "TextureModule: 3*num_roi must be a multiple of 4");
  // RoIAlign -> (B,128,128,36): channel = roi*3 + c  (swapnet_modules.py:234-240)
  Var pooled = n.alloc_var(B, 128, 128, RC, false);
  const TView texv = tex.v, pv = pooled.v;
  n.custom("roi_align", [=](Net& nn) { roi_align_fwd(nn.ctx.s, texv, 3, rois_dev, num_roi, pv); }, nullptr, {});
  n.taps["pooled"] = pooled;
  // encode = UNetDown(36, 36) (:170,242)
  Var enc_raw = n.alloc_var(B, 64, 64, RC, true);
  Var enc = n.alloc_var(B, 64, 64, RC, true);
  n.conv("encode.model.0", pooled, enc_raw, CK_K4S2, RC, RC, false, ACT_NONE, nullptr, true);
  n.norm_act(enc_raw, enc, true, ACT_LRELU, 0.f);
  n.taps["encoded"] = enc;
  // nearest upsample to the input size, written into the first 36 channels of the U-Net input (:244-258)
  n.upsample(enc, unet_in.slice(0, RC), H / 64);
  // ---- U-Net
  auto inner = [&](int d) { return d == 0 ? 64 : d == 1 ? 128 : d == 2 ? 256 : 512; };   // inner_nc of block d
  std::vector<Var> C(depth), R(depth);      // C_d = [l_d | u_d], R_d = relu(C_d)
  for (int d = 1; d < depth; ++d) {
    const int ch = inner(d - 1), hw = H >> d;
    C[d] = n.alloc_var(B, hw, hw, 2 * ch, true);
    R[d] = n.alloc_var(B, hw, hw, 2 * ch, true);
  }
  // down path
  // (the input gradient is only wanted for the pooled-texture channels: the cloth channels are data)
  n.conv(unet_prefix(0) + ".model.0", unet_in, C[1].slice(0, 64), CK_K4S2, RC + cloth_channels, 64, true, ACT_LRELU, nullptr, false, RC);
  Var innermost_mid;
  for (int d = 1; d < depth; ++d) {
    const int cin = inner(d - 1), cout = inner(d), hw = H >> (d + 1);
    const std::string name = unet_prefix(d) + ".model.1";
    if (d < depth - 1) {
      Var raw = n.alloc_var(B, hw, hw, cout, true);
      n.conv(name, C[d].slice(0, cin), raw, CK_K4S2, cin, cout, true, ACT_NONE);
      n.norm_act(raw, C[d + 1].slice(0, cout), true, ACT_LRELU, 0.f);
    } else {                                      // innermost: conv -> (uprelu) -> convT
      innermost_mid = n.alloc_var(B, hw, hw, cout, true);
      n.conv(name, C[d].slice(0, cin), innermost_mid, CK_K4S2, cin, cout, true, ACT_RELU);
    }
  }
  // up path
  for (int d = depth - 1; d >= 1; --d) {
    const int ch = inner(d - 1), hw = H >> d;
    const bool innermost = d == depth - 1;
    Var raw = n.alloc_var(B, hw, hw, ch, true);
    const std::string name = unet_prefix(d) + (innermost ? ".model.3" : ".model.5");
    n.convT(name, innermost ? innermost_mid : R[d + 1], raw, ch, true);
    const float drop = (d >= 4 && d < depth - 1) ? 0.5f : 0.f;
    n.norm_act(raw, C[d].slice(ch, ch), true, ACT_NONE, drop);
    n.act(C[d], R[d], ACT_RELU);
  }
  Var out_raw = n.alloc_var(B, H, W, 4, true);
  n.convT(unet_prefix(0) + ".model.3", R[1], out_raw, 3, true);
  n.act(out_raw, out, ACT_TANH);
  n.taps["fakes"] = out;
  n.taps["unet_in"] = unet_in;
}

// ---------------------------------------------------------------------------------------
// VGG16 features[0:30] split into the 5 slices of PerceptualLoss (perceptual.py:28-42).
// Parameter names follow PerceptualLoss.state_dict(): net.<slice>.<vgg index>.{weight,bias}.
// ---------------------------------------------------------------------------------------
std::vector<Var> build_vgg16_slices(Net& n, const Var& img) {
  static const int cfg[13] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
  static const int vidx[13] = {0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28};
  static const int slice_of[13] = {0, 0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4};
  const bool grad = img.has_grad;
  std::vector<Var> outs;
  Var x = img;
  int cin = 3, H = img.v.H, W = img.v.W;
  for (int i = 0; i < 13; ++i) {
    const bool first_of_slice = i > 0 && slice_of[i] != slice_of[i - 1];
    if (first_of_slice) {
      outs.push_back(x);
      H /= 2; W /= 2;
      Var p = n.alloc_var(x.v.N, H, W, x.v.C, grad);
      n.maxpool(x, p);
      x = p;
    }
    Var y = n.alloc_var(x.v.N, H, W, cfg[i], grad);
    n.conv("net." + std::to_string(slice_of[i]) + "." + std::to_string(vidx[i]), x, y, CK_K3S1_ZERO, cin, cfg[i], true,
           ACT_RELU, nullptr, false, i == 0 ? 4 : 0);
    x = y;
    cin = cfg[i];
  }
  outs.push_back(x);
  return outs;
}

// ---------------------------------------------------------------------------------------
// TextureModel (models/texture_model.py).  D input = cat(cloths, textures) (:138,142,164), held
// as [texture(3)+0 | cloth(19)+0] so the generator output / L1 / VGG input is an aligned slice.
// ---------------------------------------------------------------------------------------
class TextureModel final : public Model {
 public:
  Var tex, unet_in, Dx, pred2, pred1, vin_f, vin_t;
  float* rois = nullptr;
  int num_roi = 12;
  ParamArena ownV_;
  ParamArena& arenaV;
  std::unique_ptr<Net> VF, VT;
  Net* net_for_patterns(int net) override { return net == 3 ? VF.get() : Model::net_for_patterns(net); }
  std::vector<Var> feat_f, feat_t;

  ParamArena* arena_ptr(int net) override {
    if (net == 2) return is_train ? &arenaV : nullptr;
    return Model::arena_ptr(net);
  }

  int Cc = 19, Ccp = 20, RC = 36;       // cloth channels (logical / padded), ROI-pooled texture channels 3 * num_roi
  TextureModel(Ctx& c, int B_, int H_, int W_, bool train, int nroi, int cloth_channels, TextureModel* share = nullptr)
      : Model(share), arenaV(share ? share->arenaV : ownV_) {
    ctx = &c; B = B_; H = H_; W = W_; is_train = train; num_roi = nroi;
    Cc = cloth_channels; Ccp = round_up(Cc, 4); RC = 3 * num_roi;
    if (Cc < 1 || Cc > 64) throw Error(1, "TextureModel: cloth_channels in [1,64]");
    AllocScope mine(c, owned_allocs);
    G = std::make_unique<Net>(c, arenaG);
    G->keep_wino_inputs = train;
    G->s2_wino = getenv("SWN_WINO_S2") && atoi(getenv("SWN_WINO_S2")) == 2;
    tex = G->alloc_var(B, H, W, 4, false);
    // first-layer buffers padded to multiples of 16 channels (zero pads meeting zero weight rows): 56 -> 64, 24 -> 32 and, for
    // VGG16's conv1_1, 4 -> 16 put those layers on the ring kernels (nets.cpp ring_pad; SWN_FIRST_RING=0: the round-3 layout)
    unet_in = G->alloc_var(B, H, W, ring_pad(RC + Ccp), true);     // d(unet_in)[0:RC) feeds the encode branch
    const int CdB = ring_pad(4 + Ccp);
    Dx = G->alloc_var(train ? 2 * B : B, H, W, CdB, train);
    rois = static_cast<float*>(c.alloc((size_t)B * num_roi * 4 * sizeof(float)));
    Var fake_slot = Dx.batch(0, B).slice(0, 4);
    build_texture_generator(*G, tex, rois, num_roi, unet_in.slice(RC, Ccp), unet_in, fake_slot, H, Cc);
    if (!arenaG.frozen) arenaG.allocate(c);
    G->finalize({fake_slot});
    losses = static_cast<float*>(c.alloc(L_COUNT * sizeof(float)));
    if (!train) return;
    std::vector<int32_t> cimap(CdB, -1);
    for (int i = 0; i < 3; ++i) cimap[i] = Cc + i;       // textures follow the cloth channels (texture_model.py:135)
    for (int i = 0; i < Cc; ++i) cimap[4 + i] = i;
    d_cimap_ = cimap; d_layers_ = c.patchgan_layers;
    D2 = std::make_unique<Net>(c, arenaD);
    D2->keep_wino_inputs = true;
    pred2 = build_patchgan(*D2, Dx, c.patchgan_layers, cimap);
    if (!arenaD.frozen) arenaD.allocate(c);
    D2->finalize({pred2});
    D1 = std::make_unique<Net>(c, arenaD);
    pred1 = build_patchgan(*D1, Dx.batch(0, B), c.patchgan_layers, cimap, 4);        // d(fakes) only: the condition is data
    D1->finalize({pred1});
    // perceptual network: one instance with gradients (fakes), one without (targets, no_grad :52-53)
    VF = std::make_unique<Net>(c, arenaV);
    const int CvB = first_ring_on() ? 16 : 4;
    Var vbuf_f = VF->alloc_var(B, H, W, CvB, true);
    vin_f = vbuf_f.slice(0, 4);
    VF->affine(fake_slot, vin_f, 2.f, -1.f);              // x <- 2x - 1 (perceptual.py:70)
    feat_f = build_vgg16_slices(*VF, vbuf_f);
    if (!arenaV.frozen) arenaV.allocate(c);
    VT = std::make_unique<Net>(c, arenaV);
    Var vbuf_t = VT->alloc_var(B, H, W, CvB, false);
    vin_t = vbuf_t.slice(0, 4);
    VT->affine(Dx.batch(B, B).slice(0, 4), vin_t, 2.f, -1.f);
    feat_t = build_vgg16_slices(*VT, vbuf_t);
    VT->finalize({});
    // backward_G writes d(content)/d(feature_k) into every slice output and d(GAN + L1)/d(fakes)
    // into the fake slot BEFORE VF->backward(): register them as pre-initialised so the tape
    // (next slice's max-pool backward, the 2x-1 affine) accumulates on top instead of overwriting.
    std::vector<Var> pre(feat_f.begin(), feat_f.end());
    pre.push_back(fake_slot);
    VF->finalize(pre);
  }

  // PerceptualLoss.forward (modules/losses/perceptual.py:49-66) on caller-supplied images.  The VGG instances of
  // the training step are reused from their second op on (op 0 is the 2x-1 affine reading the model's own fake /
  // target slots, which stay untouched); raw images and the image gradient live in lazily allocated scratch.
  Var p_out_, p_tgt_;
  void perceptual(const float* output_nchw, const float* target_nchw, int use_style, float* out2, float content_w,
                  float style_w, float* d_output_nchw) override {
    if (!is_train) throw Error(1, "perceptual: the model was created without its loss networks (is_train = 0)");
    Stream& s = ctx->s;
    vt_done_ = false;                      // (the target-feature buffers are about to hold another image's)
    if (!p_out_.v.p) {
      AllocScope mine(*ctx, owned_allocs);
      p_out_ = G->alloc_var(B, H, W, 4, true);
      p_tgt_ = G->alloc_var(B, H, W, 4, false);
    }
    nchw_to_nhwc(s, output_nchw, B, 3, H, W, p_out_.v);
    nchw_to_nhwc(s, target_nchw, B, 3, H, W, p_tgt_.v);
    axpy(s, p_out_.v, vin_f.v, 2.f, 0, -1.f);            // x <- 2x - 1 (perceptual.py:70)
    axpy(s, p_tgt_.v, vin_t.v, 2.f, 0, -1.f);
    VF->refresh_dgrad();
    VT->forward_from(1);
    VF->forward_from(1);
    const bool want_grad = d_output_nchw != nullptr;
    dev_memset(s, losses + L_TMP4, 0, 2 * sizeof(float));
    for (int k = 0; k < 5; ++k) {
      normed_mse_loss(s, feat_f[k].v, feat_t[k].v, content_w, losses + L_TMP2, want_grad ? &feat_f[k].g : nullptr, 0);
      scalar_axpby(s, losses + L_TMP4, 1.f, losses + L_TMP2, 1.f, losses + L_TMP4);
    }
    if (want_grad) {
      VF->backward_range(false, true, 1, (int)VF->ops.size());      // d(content)/d(2x-1) -> vin_f.g
      axpy(s, vin_f.g, p_out_.g, 2.f, 0, 0.f);
    }
    if (use_style) {                                       // 5 identical image-Gram terms (perceptual.py:58-63)
      gram_style_loss(s, p_out_.v, p_tgt_.v, 3, 5.f * style_w, losses + L_TMP3, want_grad ? &p_out_.g : nullptr, 1);
      scalar_axpby(s, losses + L_TMP3, 5.f, nullptr, 0.f, losses + L_TMP5);
    }
    dev_copy(s, out2, losses + L_TMP4, 2 * sizeof(float));
    if (want_grad) nhwc_to_nchw(s, p_out_.g, d_output_nchw, 3);
  }

  Var g_out_, g_tgt_;            // global batches of the data-parallel style term (lazily sized)
  int style_total_ = 0, style_n0_ = 0;
  bool style_ctx_ = false;
  // VGG16 features of the TARGETS (perceptual.py:49-57) depend on nothing the step computes: taken early, behind the discriminator's
  // backward pass where the main stream otherwise waits for the last weight gradients and AdamW(D) on the second stream (round 6;
  // values only, nothing accumulates: bit-identical).  SWN_VT_EARLY=0: where the reference takes them, inside backward_G.
  bool vt_done_ = false;
  void set_style_context(const float* all_out, const float* all_tgt, int n_total, int n0) override {
    if (!is_train) throw Error(1, "set_style_context: training model required");
    if (n_total < B || n0 < 0 || n0 + B > n_total) throw Error(1, "set_style_context: local range outside the global batch");
    if (!g_out_.v.p || g_out_.v.N != n_total) {
      AllocScope mine(*ctx, owned_allocs);
      g_out_ = G->alloc_var(n_total, H, W, 4, false);
      g_tgt_ = G->alloc_var(n_total, H, W, 4, false);
    }
    nchw_to_nhwc(ctx->s, all_out, n_total, 3, H, W, g_out_.v);
    nchw_to_nhwc(ctx->s, all_tgt, n_total, 3, H, W, g_tgt_.v);
    style_total_ = n_total; style_n0_ = n0; style_ctx_ = true;
  }

  void set_input(int slot, const float* src, int N, int C, int Hh, int Ww) override {
    vt_done_ = false;
    Stream& s = ctx->s;
    if (slot == 1) {                                   // rois (B,R,4)
      if (N != B || C != num_roi || Hh != 4) throw Error(1, "rois must be (B, num_roi, 4)");
      dev_copy(s, rois, src, (size_t)B * num_roi * 4 * sizeof(float));
      return;
    }
    if (N != B || Hh != H || Ww != W) throw Error(1, "set_input: shape mismatch with the model's (B,H,W)");
    if (slot == 0) {                                   // input_textures
      if (C != 3) throw Error(1, "input_textures must have 3 channels");
      nchw_to_nhwc(s, src, N, C, H, W, tex.v);
    } else if (slot == 2) {                            // cloths
      if (C != Cc) throw Error(1, "cloths must have " + std::to_string(Cc) + " channels");
      nchw_to_nhwc(s, src, N, C, H, W, unet_in.v.slice(RC, Ccp));
      nchw_to_nhwc(s, src, N, C, H, W, Dx.batch(0, B).v.slice(4, Ccp));
      if (is_train) nchw_to_nhwc(s, src, N, C, H, W, Dx.batch(B, B).v.slice(4, Ccp));
    } else if (slot == 3) {                            // target_textures
      if (!is_train) throw Error(1, "targets are only used in training");
      if (C != 3) throw Error(1, "target_textures must have 3 channels");
      nchw_to_nhwc(s, src, N, C, H, W, Dx.batch(B, B).v.slice(0, 4));
    } else {
      throw Error(1, "set_input: unknown slot");
    }
  }
  void set_input_labels(int slot, const int32_t* lab, int N, int Hh, int Ww) override {
    vt_done_ = false;
    if (N != B || Hh != H || Ww != W) throw Error(1, "set_input_labels: shape mismatch with the model's (B,H,W)");
    if (slot != 2) throw Error(1, "set_input_labels: slot has no label form");
    labels_to_onehot(ctx->s, lab, unet_in.v.slice(RC, Ccp), Cc);
    labels_to_onehot(ctx->s, lab, Dx.batch(0, B).v.slice(4, Ccp), Cc);
    if (is_train) labels_to_onehot(ctx->s, lab, Dx.batch(B, B).v.slice(4, Ccp), Cc);
  }
  void get_output(int slot, float* dst) override {
    if (slot != 0) throw Error(1, "get_output: unknown slot");
    nhwc_to_nchw(ctx->s, Dx.batch(0, B).v.slice(0, 4), dst, 3);
  }
  TView output_view() override { return Dx.batch(0, B).v.slice(0, 4); }
  int output_channels() const override { return 3; }
  void forward(bool training, uint64_t seed) override {        // texture_model.py:121-125
    G->training = training; G->seed = seed;
    G->forward();
    vt_done_ = false;
  }
  void backward_D(float label_fake, float label_real) override {       // texture_model.py:127-155
    Stream& s = ctx->s;
    D2->refresh_dgrad();
    D2->forward();
    TView pf = pred2.batch(0, B).v, pr = pred2.batch(B, B).v;
    TView gf = pred2.batch(0, B).g, gr = pred2.batch(B, B).g;
    const float gs = 0.5f * hyper.grad_scale;
    if (hyper.gan_mode == 0) { bce_logits_loss(s, pf, label_fake, gs, losses + L_D_FAKE, &gf, label_dev(0)); bce_logits_loss(s, pr, label_real, gs, losses + L_D_REAL, &gr, label_dev(1)); }
    else if (hyper.gan_mode == 1) { lsgan_loss(s, pf, label_fake, gs, losses + L_D_FAKE, &gf, label_dev(0)); lsgan_loss(s, pr, label_real, gs, losses + L_D_REAL, &gr, label_dev(1)); }
    else { wgan_loss(s, pf, 1.f, gs, losses + L_D_FAKE, &gf); wgan_loss(s, pr, -1.f, gs, losses + L_D_REAL, &gr); }
    scalar_axpby(s, losses + L_D_FAKE, 0.5f, losses + L_D_REAL, 0.5f, losses + L_D);
    D2->backward_range(true, false, 0, (int)D2->ops.size(), /*join=*/false);
    static const bool vt_early = !(getenv("SWN_VT_EARLY") && atoi(getenv("SWN_VT_EARLY")) == 0);
    if (vt_early && VT && (hyper.lambda_content != 0.f || hyper.lambda_style != 0.f)) {
      VF->refresh_dgrad();                 // (the frozen VGG16's operands: derived once)
      VT->forward();
      vt_done_ = true;
    }
    ctx->join_side();                      // every D gradient is final for whatever the main stream does next
    // (texture_model.py:148-153 hands the UNconditioned 3-channel targets / fakes to the 22-channel discriminator in the
    // gradient-penalty modes, which raises in the reference; set_hyper rejects gp_mode for this model)
    dev_memset(s, losses + L_D_GP, 0, sizeof(float));
  }
  void backward_G(float label_real) override {                          // texture_model.py:157-180
    backward_G_head(label_real);
    G->refresh_dgrad();
    G->backward(true, false);
  }
  void backward_G_head(float label_real) override {
    Stream& s = ctx->s;
    const float gsc = hyper.grad_scale;
    TView fakes = Dx.batch(0, B).v.slice(0, 4), dfakes = Dx.batch(0, B).g.slice(0, 4);
    TView targets = Dx.batch(B, B).v.slice(0, 4);
    D1->refresh_dgrad();
    D1->forward();
    if (hyper.gan_mode == 0) bce_logits_loss(s, pred1.v, label_real, hyper.lambda_gan * gsc, losses + L_TMP0, &pred1.g, label_dev(2));
    else if (hyper.gan_mode == 1) lsgan_loss(s, pred1.v, label_real, hyper.lambda_gan * gsc, losses + L_TMP0, &pred1.g, label_dev(2));
    else wgan_loss(s, pred1.v, -1.f, hyper.lambda_gan * gsc, losses + L_TMP0, &pred1.g);
    scalar_axpby(s, losses + L_TMP0, hyper.lambda_gan, nullptr, 0.f, losses + L_G_GAN);
    D1->backward(false, true);                           // first writer of d(fakes)
    // L1 (:168-170)
    l1_loss(s, fakes, targets, 3, hyper.lambda_l1 * gsc, losses + L_TMP1, &dfakes, 1);
    scalar_axpby(s, losses + L_TMP1, hyper.lambda_l1, nullptr, 0.f, losses + L_G_L1);
    // perceptual (:171-176; perceptual.py:49-66)
    dev_memset(s, losses + L_G_CONTENT, 0, 2 * sizeof(float));
    if (hyper.lambda_content != 0.f || hyper.lambda_style != 0.f) {
      VF->refresh_dgrad();
      if (!vt_done_) VT->forward();
      vt_done_ = false;
      VF->forward();
      for (int k = 0; k < 5; ++k) {
        normed_mse_loss(s, feat_f[k].v, feat_t[k].v, hyper.lambda_content * gsc, losses + L_TMP2, &feat_f[k].g, 0);
        scalar_axpby(s, losses + L_G_CONTENT, 1.f, losses + L_TMP2, hyper.lambda_content, losses + L_G_CONTENT);
      }
      VF->backward(false, true);                         // accumulates 2 * d/d(2x-1) into d(fakes)
      if (hyper.lambda_style != 0.f) {
        // 5 identical image-Gram terms (perceptual.py:58-63)
        if (style_ctx_) {
          // data parallel: Gram over the gathered global batch, gradient for this rank's samples; the factor
          // n_total / B undoes grad_scale = 1 / world (the term is the same global value on every rank, and each rank
          // contributes exactly the rows it owns)
          gram_style_loss(s, g_out_.v, g_tgt_.v, 3, 5.f * hyper.lambda_style * gsc * (float)style_total_ / (float)B, losses + L_TMP3,
                          &dfakes, 1, style_n0_, B);
          style_ctx_ = false;
        } else {
          gram_style_loss(s, fakes, targets, 3, 5.f * hyper.lambda_style * gsc, losses + L_TMP3, &dfakes, 1);
        }
        scalar_axpby(s, losses + L_TMP3, 5.f * hyper.lambda_style, nullptr, 0.f, losses + L_G_STYLE);
      }
    }
    scalar_axpby(s, losses + L_G_GAN, 1.f, losses + L_G_L1, 1.f, losses + L_TMP4);
    scalar_axpby(s, losses + L_G_CONTENT, 1.f, losses + L_G_STYLE, 1.f, losses + L_TMP5);
    scalar_axpby(s, losses + L_TMP4, 1.f, losses + L_TMP5, 1.f, losses + L_G);
  }
};

Model* create_texture_model(Ctx& ctx, int B, int H, int W, bool is_train, int num_roi, int cloth_channels, Model* share) {
  TextureModel* sh = dynamic_cast<TextureModel*>(share);
  if (share && !sh) throw Error(1, "shared model: the sharer is not a texture model");
  return new TextureModel(ctx, B, H, W, is_train, num_roi, cloth_channels, sh);
}

}  // namespace swn
