// swapnet_amd -- gradient-penalty objectives of the discriminator step (modules/loss.py:133-184, called from
// models/warp_model.py:126-136): wgan-gp, dragan-gp, dragan-lp.  SURVEY.md 8(f) rank 4.
//
//   x_hat = a + alpha (b - a)          a = conditioned real batch, b = conditioned fakes (wgan) or a + 0.5 std(a) U[0,1)
//   g     = d sum(D(x_hat)) / d x_hat  (torch.autograd.grad(..., create_graph=True), loss.py:160-162)
//   gp    = mean_n (||g_n||_2 - 1)^2   (lp: max(0, . )^2);   loss_D += lambda_gp * gp
// and the step needs d gp / d theta_D: a derivative THROUGH the backward pass of D (reverse over reverse).  With the
// layers h_l = f_l(h_{l-1}; theta_l) and the first backward gbar_{l-1} = J_l^T gbar_l (gbar_L = 1, g = gbar_0):
//   up pass   (l = 1..L):  u_0 = d gp / d g;  u_l = J_l u_{l-1}  (the layer's linearisation applied to u);
//                          d theta_l += (d/d theta_l)[J_l^T gbar_l]^T u_{l-1};  a_{l-1} = (d/d h_{l-1})[J_l^T gbar_l]^T u_{l-1}
//   down pass (l = L..1):  an ordinary backward pass of the injected adjoints a_l.
// Layer by layer for PatchGAN (modules/discriminators.py:110-131): Conv is linear (J u = W * u without bias; its
// second-order weight term is the weight-gradient contraction of u_{l-1} with gbar_l; no activation adjoint);
// LeakyReLU is piecewise linear (mask only); InstanceNorm is the one non-linear layer: ops.h norm_act_bwd2 gives
// both J u and the injected adjoint.  Every contraction reuses the MFMA conv / wgrad kernels of the training step
// (direct form: this non-default mode does not use the Winograd / taps-on-N variants).
#include <cstring>

#include "engine.h"

namespace swn {

namespace {
TView flat_view(float* p, size_t n) {
  TView v; v.p = p; v.N = 1; v.H = 1; v.W = (int)(n / 4); v.C = 4; v.cs = 4; return v;
}
}  // namespace

// n_layers = the PatchGAN's stride-2 levels (define_D's n_layers_D, modules/discriminators.py:110-131): conv k of its
// nn.Sequential sits at index 0, then 2 + 3 (k - 1) -- [conv, norm, lrelu] triples behind [conv, lrelu]; layers 0 .. n_layers - 1
// are k4 s2, layer n_layers the k4 s1 conv with a norm, layer n_layers + 1 the k4 s1 prediction conv; norms on 1 .. n_layers.
GradPenalty::GradPenalty(Ctx& c, ParamArena& arenaD, int B, int H, int W, int n_layers)
    : ctx_(c), A_(arenaD), B_(B), nl_(n_layers), NL_(n_layers + 2) {
  if (n_layers < 1 || n_layers > 5) throw Error(1, "GradPenalty: n_layers_D must be in [1, 5]");
  net_ = std::make_unique<Net>(c, arenaD);
  Net& n = *net_;
  const int nl = nl_, NL = NL_;
  std::vector<std::string> names(NL);
  for (int l = 0; l < NL; ++l) names[l] = "model." + std::to_string(l == 0 ? 0 : 2 + 3 * (l - 1));
  int h = H, w = W;
  const ParamDesc& w0 = arenaD.params[arenaD.index.at("model.0.weight")];
  Cd_ = w0.ws.Cip;                        // channels of D's conditional input buffer (padded layout of the model)
  Cd_logical_ = w0.ws.Ci;
  xh_ = n.alloc_var(B, H, W, Cd_, true);
  u0_ = n.alloc_var(B, H, W, Cd_, false).v;
  L_.resize(NL);
  for (int l = 0; l < NL; ++l) {
    Layer& y = L_[l];
    const ParamDesc& wd = arenaD.params[arenaD.index.at(names[l] + ".weight")];
    const ParamDesc& bd = arenaD.params[arenaD.index.at(names[l] + ".bias")];
    y.ws = wd.ws; y.woff = wd.off; y.boff = bd.off;
    y.kind = l < nl ? 0 : 1;
    y.Cin = wd.ws.Cip; y.Co = wd.ws.Co; y.Cop = round_up(wd.ws.Co, 4);
    y.norm = l >= 1 && l <= nl;
    if (y.kind == 0) { h /= 2; w /= 2; } else { h -= 1; w -= 1; }
    y.h = n.alloc_var(B, h, w, y.Cop, true);
    if (y.norm) {
      y.raw = n.alloc_var(B, h, w, y.Cop, true);
      y.stats = static_cast<float*>(c.alloc((size_t)B * y.Cop * 2 * sizeof(float)));
      y.tmp = n.alloc_var(B, h, w, y.Cop, false).v;
    }
    if (l == 0) y.gr0 = n.alloc_var(B, h, w, y.Cop, false).v;
    if (l < NL - 1) {
      y.u_raw = n.alloc_var(B, h, w, y.Cop, false).v;
      y.u_h = n.alloc_var(B, h, w, y.Cop, false).v;
      y.a_raw = n.alloc_var(B, h, w, y.Cop, false).v;
      if (l < nl) y.a_h = n.alloc_var(B, h, w, y.Cop, false).v;
    }
    y.dg = static_cast<float*>(c.alloc(dgrad_elems(y.ws, y.kind == 0 ? 0 : 1, y.Cop, y.Cin) * sizeof(float)));
  }
  gA_ = static_cast<float*>(c.alloc(arenaD.n * sizeof(float)));
  gB_ = static_cast<float*>(c.alloc(arenaD.n * sizeof(float)));
  alpha_ = static_cast<float*>(c.alloc(round_up(B, 4) * sizeof(float)));
  half_std_ = static_cast<float*>(c.alloc(16));
  tmp_loss_ = static_cast<float*>(c.alloc(16));
  beta_ = n.alloc_var(B, H, W, Cd_, false).v;
}
GradPenalty::~GradPenalty() {}

void GradPenalty::conv(int l, const TView& x, const TView& y, bool bias, int act) {
  const Layer& L = L_[l];
  ConvFwdArgs a;
  a.x = x;
  a.g.KH = a.g.KW = 4; a.g.pad_t = a.g.pad_l = 1;
  a.g.stride = L.kind == 0 ? 2 : 1;
  a.g.Ho = y.H; a.g.Wo = y.W;
  a.w = A_.w + L.woff; a.Npad = L.ws.Npad;
  a.bias = bias ? A_.w + L.boff : nullptr;
  a.act = act; a.y = y; a.Cout = L.Co;
  conv_fwd(ctx_.s, a);
}
void GradPenalty::dgrad(int l, const TView& dy, const TView& dx) {
  const Layer& L = L_[l];
  ConvFwdArgs d;
  d.x = dy;
  if (L.kind == 0) {          // four sub-pixel phases of the transposed conv
    d.g.KH = d.g.KW = 2; d.g.stride = 1; d.g.pad_t = d.g.pad_l = 1; d.g.Ho = dy.H; d.g.Wo = dy.W;
    d.w = L.dg; d.w_bs = (size_t)4 * L.Cop * L.Cin; d.Npad = L.Cin; d.Cout = L.Cin;
    d.y = dx; d.om.ymul = d.om.xmul = 2; d.phases = 4;
  } else {
    d.g.KH = d.g.KW = 4; d.g.stride = 1; d.g.pad_t = d.g.pad_l = 2; d.g.Ho = dx.H; d.g.Wo = dx.W;
    d.w = L.dg; d.Npad = L.Cin; d.Cout = L.Cin; d.y = dx;
  }
  conv_fwd(ctx_.s, d);
}
void GradPenalty::wgrad(int l, const TView& x, const TView& dy, float* arena) {
  const Layer& L = L_[l];
  ConvWgradArgs a;
  a.x = x;
  a.g.KH = a.g.KW = 4; a.g.pad_t = a.g.pad_l = 1;
  a.g.stride = L.kind == 0 ? 2 : 1;
  a.g.Ho = dy.H; a.g.Wo = dy.W;
  a.dy = dy; a.dw = arena + L.woff; a.Npad = L.ws.Npad; a.Cout = L.Co;
  conv_wgrad(ctx_.s, a);
}

void GradPenalty::run(const TView& real, const TView& fake, int gp_mode, float grad_scale, float lambda_gp,
                      const float* alpha, const TView* beta, uint64_t seed, float* loss_gp_out) {
  Stream& s = ctx_.s;
  const bool dragan = gp_mode >= 2;
  const int lp = gp_mode == 3;
  // ---- random draws (host-provided for seeded parity with the reference's CPU RNG, else the library's counter RNG)
  if (!alpha) {
    TView av = flat_view(alpha_, (size_t)round_up(B_, 4));
    gp_uniform(s, av, 4, seed * 2 + 1);
    alpha = alpha_;
  }
  if (dragan) {
    // layout pads (cimap < 0) must stay exactly 0: x_hat feeds wgrad(0, x_hat, .), and a non-zero pad channel would put a
    // gradient on the pad rows of model.0.weight, which AdamW would then move off zero
    if (!beta) { gp_uniform(s, beta_, Cd_, seed * 2 + 2, L_[0].ws.cimap); beta = &beta_; }
    gp_half_std(s, real, (size_t)real.N * real.H * real.W * Cd_logical_, half_std_);
    gp_interpolate(s, real, nullptr, alpha, beta, half_std_, xh_.v);
  } else {
    gp_interpolate(s, real, &fake, alpha, nullptr, nullptr, xh_.v);
  }
  const int nl = nl_, P = NL_ - 1;          // P: the prediction conv
  for (int l = 0; l < NL_; ++l) repack_dgrad(s, L_[l].ws, L_[l].kind == 0 ? 0 : 1, L_[l].Cop, L_[l].Cin, A_.w + L_[l].woff, L_[l].dg);

  // ---- forward
  conv(0, xh_.v, L_[0].h.v, true, ACT_LRELU);
  for (int l = 1; l <= nl; ++l) {
    conv(l, L_[l - 1].h.v, L_[l].raw.v, true, ACT_NONE);
    NormActArgs a;
    a.x = L_[l].raw.v; a.y = L_[l].h.v; a.stats = L_[l].stats; a.norm = 1; a.act = ACT_LRELU;
    norm_act_fwd(s, a);
  }
  conv(P, L_[nl].h.v, L_[P].h.v, true, ACT_NONE);
  // ---- first backward with grad_outputs = ones (loss.py:160-162)
  const TView pred = L_[P].h.v, gpred = L_[P].h.g;
  wgan_loss(s, pred, 1.f, (float)pred.pixels(), tmp_loss_, &gpred);          // d(sum pred)/d pred = 1 on channel 0
  dgrad(P, gpred, L_[nl].h.g);
  for (int l = nl; l >= 1; --l) {
    NormActBwdArgs b;
    b.dy = L_[l].h.g; b.x = L_[l].raw.v; b.stats = L_[l].stats; b.dx = L_[l].raw.g; b.norm = 1; b.act = ACT_LRELU;
    norm_act_bwd(s, b);
    dgrad(l, L_[l].raw.g, L_[l - 1].h.g);
  }
  act_bwd(s, L_[0].h.g, L_[0].h.v, L_[0].gr0, ACT_LRELU, 0);
  dgrad(0, L_[0].gr0, xh_.g);
  // ---- penalty value and u_0 = lambda_gp * grad_scale * d gp / d g
  gp_penalty(s, xh_.g, lp, lambda_gp * grad_scale, tmp_loss_, u0_);
  scalar_axpby(s, tmp_loss_, lambda_gp, nullptr, 0.f, loss_gp_out);
  // ---- up pass
  dev_memset(s, gA_, 0, A_.n * sizeof(float));
  dev_memset(s, gB_, 0, A_.n * sizeof(float));
  conv(0, u0_, L_[0].u_raw, false, ACT_NONE);
  wgrad(0, u0_, L_[0].gr0, gA_);
  act_bwd(s, L_[0].u_raw, L_[0].h.v, L_[0].u_h, ACT_LRELU, 0);
  for (int l = 1; l <= nl; ++l) {
    conv(l, L_[l - 1].u_h, L_[l].u_raw, false, ACT_NONE);
    wgrad(l, L_[l - 1].u_h, L_[l].raw.g, gA_);
    NormActBwd2Args b2;
    b2.u = L_[l].u_raw; b2.gy = L_[l].h.g; b2.x = L_[l].raw.v; b2.stats = L_[l].stats;
    b2.uy = L_[l].u_h; b2.ax = L_[l].a_raw; b2.act = ACT_LRELU;
    norm_act_bwd2(s, b2);
  }
  wgrad(P, L_[nl].u_h, gpred, gA_);
  // ---- down pass of the injected adjoints
  for (int l = nl; l >= 1; --l) {
    wgrad(l, L_[l - 1].h.v, L_[l].a_raw, gB_);
    bias_grad(s, L_[l].a_raw, gB_ + L_[l].boff);
    dgrad(l, L_[l].a_raw, L_[l - 1].a_h);
    if (l - 1 >= 1) {
      NormActBwdArgs b;
      b.dy = L_[l - 1].a_h; b.x = L_[l - 1].raw.v; b.stats = L_[l - 1].stats; b.dx = L_[l - 1].tmp; b.norm = 1; b.act = ACT_LRELU;
      norm_act_bwd(s, b);
      axpy(s, L_[l - 1].tmp, L_[l - 1].a_raw, 1.f, 1);
    } else {
      act_bwd(s, L_[0].a_h, L_[0].h.v, L_[0].a_raw, ACT_LRELU, 0);
    }
  }
  wgrad(0, xh_.v, L_[0].a_raw, gB_);
  bias_grad(s, L_[0].a_raw, gB_ + L_[0].boff);
  // ---- add both contributions to the discriminator's gradient arena
  axpy(s, flat_view(gA_, A_.n), flat_view(A_.g, A_.n), 1.f, 1);
  axpy(s, flat_view(gB_, A_.n), flat_view(A_.g, A_.n), 1.f, 1);
}

}  // namespace swn
