// swapnet_amd -- device op launchers.  The engine (engine.cpp) is written against this
// header only.  The product implementation is the set of HIP translation units in this
// directory (conv_gemm.hip, norm_act.hip, losses.hip, optim.hip, gather.hip).
// tests/hostsim/hostsim_ops.cpp implements the same signatures with plain loops so that
// the engine's graph / backward / packing logic can be checked in CI without a GPU; it is
// never part of the shipped library.
#pragma once
#include "common.h"

namespace swn {

struct WShape;

struct Stream {
  void* handle = nullptr;   // hipStream_t
  char* ws = nullptr;       // scratch for split-K slabs / reduction partials (device)
  size_t ws_bytes = 0;
};

// ---- stream ordering (the weight-gradient side stream of engine.cpp) ------------------------------
void* event_create();
void event_destroy(void* ev);
void event_record(void* ev, Stream& s);
void stream_wait_event(Stream& s, void* ev);      // work enqueued on s afterwards starts after the recorded point

// ---- hipGraph capture of a launch sequence (two-stage inference, pipeline.cpp) ---------------------------------
// graph_begin puts the stream into capture mode: everything enqueued until graph_end is recorded instead of executed
// (no allocation, synchronisation or host read-back in between).  graph_end returns an executable graph handle (NULL on
// the host simulator, which has no graphs: callers then simply run the sequence eagerly).
void* stream_create_current();        // a new non-blocking stream on the calling thread's current device (NULL on the simulator)
void graph_begin(Stream& s);
void* graph_end(Stream& s);
void graph_abort(Stream& s);          // ends a capture that failed half way: status ignored, any partial graph destroyed
void graph_launch(void* exec, Stream& s);
void graph_destroy(void* exec);

// ---- memory -------------------------------------------------------------------------
void* dev_alloc(size_t bytes);            // zero-filled
void dev_free(void* p);
void dev_memset(Stream& s, void* p, int v, size_t bytes);     // a kernel launch (a recordable KERNEL node, not a memset node: device.hip)
void dev_copy(Stream& s, void* dst, const void* src, size_t bytes);        // device->device
void dev_upload(Stream& s, void* dst, const void* src, size_t bytes);      // host->device
// <= 64 bytes (a multiple of 4) of host data into device memory, stream-ordered and WITHOUT a host synchronisation: the bytes travel
// as the arguments of a one-thread kernel (device.hip) -- per-step parameter blocks of a recorded step
void dev_store_small(Stream& s, void* dst, const void* src, size_t bytes);
void dev_download(Stream& s, void* dst, const void* src, size_t bytes);    // device->host (syncs)
void stream_sync(Stream& s);
void* stream_create(int device);          // selects the device, returns a new stream handle
void stream_destroy(void* handle);
void device_check(int device);            // throws unless `device` is a usable HIP device; makes it current
int is_device_build();                    // 1: HIP library, 0: CI host simulator
void conv_force_naive(int on);            // route conv_fwd/conv_wgrad to the naive checkers (tests)
// optional per-launch timing of the implicit-GEMM kernels (bench.py's roofline leg): when on,
// every conv_fwd / conv_wgrad launch is bracketed by HIP events recorded on the launch stream.
void prof_enable(int on);
void prof_reset();
// one line per kernel variant: "<name> <launches> <total_ms> <total_flops>\n"; returns bytes written
int prof_report(char* buf, int len);
// swapnet_hip.h swn_probe_mfma: register-only fp16 MFMA loop, in-kernel clock (conv_gemm.hip); synchronises the stream
void probe_mfma(Stream& s, int zeros, int iters, float* out4);
// Routing trace (swn_route_trace / swn_route_report, engine.cpp): which kernel family / algorithmic form every layer of a model
// takes under the current environment.  While on, the engine labels each tape op as it runs it (phase f = forward, b = backward,
// r = derived-operand refresh) and the launchers note the kernel they picked (implicit-GEMM launches with their M, N, K, batch
// and split schedule; the Winograd transforms by form).  The report lists the distinct (label, phase, kernel) triples in first-
// use order: tests compare the list of the configuration bench.py times with the one their own run takes.
void route_enable(int on);
bool route_on();
void route_label(const char* label, char phase);
void route_note(const char* kernel);
int route_report(char* buf, int len);      // returns the bytes the full report needs

// ---- implicit-GEMM convolution (MFMA) -----------------------------------------------
// y[map(m)][co] (=|+=) act( sum_k A[m][k] * w[k][co] + bias[co] ),  A = gather(x)
struct ConvFwdArgs {
  TView x;
  Gather g;
  const float* w = nullptr;   // [K = KH*KW*x.C][Npad]
  int Npad = 0;
  const float* bias = nullptr;
  int act = ACT_NONE;
  int accumulate = 0;         // 1: y += result (act must be NONE)
  TView y;                    // y.C = Cout (logical), may be a slice
  OutMap om;
  int Cout = 0;               // valid output channels (<= Npad)
  // batched mode (Winograd: the 16 / 36 plane GEMMs in one launch): element strides between batches
  int batch = 1;
  size_t x_bs = 0, w_bs = 0, y_bs = 0;
  // sub-pixel phase mode (k4s2 transposed convs and the dgrad of k4s2 convs): with phases = 4 one
  // launch covers the four phases ph = 2a + b of a stride-2 scatter: phase ph gathers with
  // pad_t = g.pad_t - a, pad_l = g.pad_l - b, writes outputs (2oy + a, 2ox + b) (om.ymul = om.xmul = 2)
  // and uses the weight panel w + ph * w_bs.  Excludes batch > 1.
  int phases = 0;
  // folded tail conv (tail_fold_weights below), four phases fused: w = the folded block, g = the 3x3
  // union gather (KH = KW = 3, stride 1, pad 1 on the un-upsampled input), om.ymul = om.xmul = 2; phase
  // (a,b) uses taps u < 2+a, v < 2+b and writes outputs (2oy + a, 2ox + b).
  int tail4 = 0;
  // optional: the weight operand pre-cut into bf16 planes (conv_precut below) for the column tile `wpc_bn`; `w` stays valid
  // (the launch falls back to it when it does not take the pre-cut kernel).  wpc_bs = elements between batch / phase panels.
  const uint16_t* wpc = nullptr;
  int wpc_bn = 0;
  size_t wpc_bs = 0;
  // optional: an amax slot of the A operand (AMAX_SLOT floats, device; their maximum >= max |x| over everything the launch
  // gathers).  The two-plane kernels scale the operand by a power of two from it; without a slot the launch takes the amax itself
  // with a pass over x.  Producers that write the operand fill the slot for free (amax_out of the transforms below).
  const float* x_amax = nullptr;
  // optional: the A operand is stored in PAIR form (see wino_input_transform): *x_pair_k = the exponent its producer scaled it by
  const int* x_pair_k = nullptr;
  // optional: an amax slot the launch folds max |y| of everything it stores into -- in the epilogue of the pre-cut ring kernel
  // (and the reduce kernel of its split tiles), by a pass over the output view behind every other kernel family
  float* y_amax = nullptr;
  // optional (north_star's Conv + InstanceNorm fusion, modules/layers.py:12-24): the launch leaves the InstanceNorm statistics'
  // partial sums of its OUTPUT behind -- stat_partial[(m / chunk) * y.C + c][2] = (sum, sum of squares) in fp64 over the `chunk`
  // consecutive output rows m of one image, chunk = conv_fwd_stat_chunk(...) -- from the accumulators in the epilogue, so that
  // norm_act_fwd needs no statistics pass over the tensor (NormActArgs::partial_in).  Requires act = NONE, accumulate = 0, one launch
  // (no batch / phases) and Ho * Wo a multiple of the chunk.
  double* stat_partial = nullptr;
};
// rows per statistics partial if a forward conv (K = taps x xC) over xC input channels into Npad columns, nimg images of HoWo output
// pixels each, can emit them -- the 128 x 128 pre-cut ring kernel, HoWo a multiple of its 128 rows, and a launch the planner runs in
// whole tiles anyway (no K split to give up) -- else 0 (always 0 on the host simulator)
int conv_fwd_stat_chunk(int xC, int Npad, int HoWo, int nimg, int K);
constexpr int AMAX_SLOT = 256;
// 256 partial maxima of |x| over a view into `slot` (overwrites all AMAX_SLOT entries): the amax of a tensor no kernel of ours
// produced (network inputs)
void tensor_amax(Stream& s, const TView& x, float* slot, float floor = 0.f);
void conv_fwd(Stream& s, const ConvFwdArgs& a);
// Pre-cut weight operand of the LDS-DMA ring kernel (conv_gemm.hip conv_fwd_pc_kernel): x = hi + mid + lo, three bf16 planes by
// truncation (exact), laid out in MFMA operand order per 16-k stage and column tile.  conv_precut_tile: the column tile (64 /
// 128) a forward-type launch over xC input channels into Npad columns takes, 0 = it would not use a pre-cut operand (host
// simulator, narrow / first-layer launches, SWN_PRECUT=0): callers then need not produce one.
int conv_precut_tile(int xC, int Npad);
size_t conv_precut_elems(int K, int Npad, int bn);       // uint16 elements of one [K][Npad] panel
// amax_io (two-plane form, optional): a layer's forward and input-gradient operands are permutations (transposes, flips, leading
// channels) of ONE parameter tensor, so the 256 partial maxima taken for the first bound the second.  *amax_io == NULL on entry: the
// launch takes its own and leaves the pointer; != NULL: it uses them -- valid only back to back on one stream (the partials live in the
// stream scratch until the next pass) and for sources whose values are a subset of what the partials were taken over.
void conv_precut(Stream& s, const float* w, int K, int Npad, int bn, int batch, size_t w_bs, uint16_t* out, const float** amax_io = nullptr);
// Plane format of the pre-cut operands: 3 = three bf16 planes (six MFMAs per product), 2 = two fp16 planes of the operand times a
// power of two chosen from its amax (three MFMAs; conv_gemm.hip "two fp16 planes").  In the two-plane form a panel carries a
// 16-byte trailer with the scale exponent (counted by conv_precut_elems) and a producer first takes the 256 partial maxima of its
// SOURCE tensor ([batch][rows][C] floats, dense rows, batch stride bs) with conv_precut_amax (NULL in the three-plane form).
int conv_precut_planes();
const float* conv_precut_amax(Stream& s, const float* src, size_t rows, int C, int batch, size_t bs);

// dw[k][co] = sum_m A[m][k] * dy[map(m)][co]      (dw: [K][Npad], overwritten)
struct ConvWgradArgs {
  TView x;
  Gather g;
  TView dy;
  OutMap om;
  float* dw = nullptr;
  int Npad = 0;
  int Cout = 0;
  int batch = 1;
  size_t x_bs = 0, dy_bs = 0, dw_bs = 0;
  int phases = 0;             // as in ConvFwdArgs; phase ph reads dy at (2oy + a, 2ox + b), writes dw + ph * dw_bs
  int tail4 = 0;              // as in ConvFwdArgs; dw = the folded-gradient block (layout of tail_fold_weights)
  const float* x_amax = nullptr;      // amax slots of the two operands (as ConvFwdArgs::x_amax)
  const float* dy_amax = nullptr;
  const int* x_pair_k = nullptr;      // operands stored in pair form (as ConvFwdArgs::x_pair_k): scale exponents of their producers
  const int* dy_pair_k = nullptr;
};
// true when the library's transforms / GEMMs implement the pair form (the device build with two-plane operands; SWN_PAIR=0 disables)
bool wino_pair_planes();
// would a batched plane GEMM of these dimensions read pair-form operands? (forward-type: A = planes with xC channels into Npad
// columns; weight-gradient: T rows reduced, K x Npad outputs).  False on the host simulator.
bool conv_fwd_takes_pairs(int xC, int Npad);
bool conv_wgrad_takes_pairs(size_t T, int K, int Npad);
void conv_wgrad(Stream& s, const ConvWgradArgs& a);

// reference implementations of the two launches above (one thread per output element,
// no MFMA): used by tests to cross-check the tiled kernels at sizes the CPU cannot reach.
void conv_fwd_naive(Stream& s, const ConvFwdArgs& a);
void conv_wgrad_naive(Stream& s, const ConvWgradArgs& a);

// db[c] = sum over pixels of dy[.,c]   (db overwritten)
void bias_grad(Stream& s, const TView& dy, float* db);

// dx[i] (+)= sum_{u: reflect(u-1)==i} dxpad[u]   -- folds the (H+2)x(W+2) gradient of a
// ReflectionPad2d(1) back onto the HxW tensor.
void reflect_fold(Stream& s, const TView& dxpad, const TView& dx, int accumulate);

// ---- Winograd F(m x m, r x r) transforms (stride-1 convolutions; see wino.hip) -------------------
// supported (m, r): (2,3), (4,3) for the 3x3 convs, (3,4) for PatchGAN's k4 s1 conv.
// tile (n,ty,tx) covers outputs (m*ty..m*ty+m-1, m*tx..) and input rows m*ty-pad .. m*ty-pad+m+r-2;
// P = (m+r-1)^2 transform planes (16 for F(2,3): 2.25x fewer multiplies; 36 for F(4,3) / F(3,4): 4x fewer)
// amax_out (optional, here and below): an amax slot (AMAX_SLOT floats, zeroed by the caller before the first producer of a
// tensor runs) into which the kernel folds max |v| over everything it writes: entry blockIdx % AMAX_SLOT, an atomic max on
// the bit pattern (non-negative floats order like unsigned integers, so the slot's maximum is exact and order-independent)
// in_amax / kscale_out (optional, the 6-point and strided forms): write the planes in PAIR form for the two-plane GEMMs -- each
// element the 32-bit word {h | l << 16} of fp16 planes of (x 2^k), k derived from the amax slot `in_amax` of the transform's INPUT
// through the transform's gain bound (wino.hip) and published in *kscale_out (device int) for the GEMM (ConvFwdArgs::x_pair_k)
void wino_input_transform(Stream& s, int m, int r, const TView& x, int pad, int pad_mode, int Th, int Tw, float* V,
                          float* amax_out = nullptr, const float* in_amax = nullptr, int* kscale_out = nullptr);        // V[P][T][x.C]
// mode 0: U[P][Cip][Npad] for the forward conv; mode 1: U[P][Npad][Cip] (flipped, transposed) for the transposed-conv form of
// dgrad; mode 2: U[P][Npad][Cip] = mode 0 with the channel axes swapped, the operand of the adjoint form (wino_input_adjoint)
void wino_filter_transform(Stream& s, int m, int r, const WShape& w, int mode, const float* packed, float* U);
// the same transform (modes 0 and 2, 6-point forms) written straight into the pre-cut operand layout of the ring kernel
// (conv_precut's, one panel of `panel_elems` = conv_precut_elems(K, N, bn) uint16 per Winograd plane): no fp32 U at all
void wino_filter_transform_pc(Stream& s, int m, int r, const WShape& w, int mode, const float* packed, int bn, uint16_t* out,
                              size_t panel_elems, const float** amax_io = nullptr);       // amax_io: as conv_precut's
// Input gradient in the forward tiling: dV[P][T][C] = dM U^T (dM = wino_dy_transform of dY, T = dx.N * Th * Tw forward tiles).
// dx (+)= sum over tiles of the patches BT^T dV_t BT placed where wino_input_transform(pad, pad_mode, Th, Tw) gathered them
// (reflected / dropped exactly like the forward gather).  dV is overwritten (scratch).
void wino_input_adjoint(Stream& s, int m, int r, float* dV, int C, int pad, int pad_mode, int Th, int Tw, const TView& dx,
                        int accumulate);
void wino_output_transform(Stream& s, int m, int r, const float* M, int Cm, int Th, int Tw, const float* bias, int act,
                           const TView& y, int Cout, int accumulate, float* amax_out = nullptr);    // amax_out: 6-point forms only                                      // M[P][T][Cm]
void wino_dy_transform(Stream& s, int m, int r, const TView& dy, int Th, int Tw, float* dM, float* amax_out = nullptr,
                       const float* in_amax = nullptr, int* kscale_out = nullptr);                                     // dM[P][T][dy.C]
// ---- the folded tail conv (tail_fold_weights below) in Winograd form: its four sub-pixel phases are (2+a)x(2+b)-tap stride-1
// convolutions over the same input, i.e. four F(4x4,3x3) convolutions sharing ONE wino_input_transform(4, 3, x, pad 1, zero);
// their filters sit side by side on the N axis (N = 4 * Npad) of one batched GEMM.  Th, Tw = tiles of 4x4 INPUT positions.
void tailw_filter_transform(Stream& s, const WShape& w, const float* folded, float* U);        // U[36][Cip][4 * Npad]
void tailw_filter_grad(Stream& s, const WShape& w, const float* dU, float* dfolded);          // dU[36][Cip][4 * Npad] -> folded layout
// y (2H x 2W, Npad channels): y[2 i + a][2 j + b] = act(A^T M_ab A + bias),  M[36][T][4 * Npad]
void tailw_output_transform(Stream& s, const float* M, int Th, int Tw, int Npad, const float* bias, int act, const TView& y, int Cout);
void tailw_dy_transform(Stream& s, const TView& dy, int Th, int Tw, int Npad, float* dM, float* amax_out = nullptr,
                        const float* in_amax = nullptr, int* kscale_out = nullptr);                        // dM[36][T][4 * Npad]
// ---- strided Winograd F(4x4, 2x2): the k4 s2 p1 convolutions and their transposes as four polyphase 2x2 stride-1 convolutions
// sharing one batched GEMM (wino.hip).  "fine" = the 2H x 2W side, "coarse" = the H x W side; tiles = 4x4 coarse pixels.
// (m, r) = (4, 2) is accepted by wino_output_transform (coarse = A^T M A) and wino_dy_transform (dM = A coarse A^T).
void wino_s2_input_transform(Stream& s, const TView& fine, int Th, int Tw, float* V, float* amax_out = nullptr,
                             const float* in_amax = nullptr, int* kscale_out = nullptr);   // V[25][T][4 * fine.C], channel (2s+t)*C + c
// fine (+)= [bias +] adjoint of the transform above applied to dV[25][T][4 * Cf] (overwritten: scratch)
void wino_s2_input_adjoint(Stream& s, float* dV, int Cf, int Th, int Tw, const TView& fine, const float* bias, int accumulate);
// w: the layer's WShape (WK_CONV: fine = input, coarse = output; WK_CONVT: fine = output, coarse = input).
// mode 0: U[25][4 * Cf][Cc] (fine -> coarse GEMM); mode 1: U[25][Cc][4 * Cf] (coarse -> fine GEMM); Cf, Cc = padded counts
void wino_s2_filter_transform(Stream& s, const WShape& w, int mode, const float* packed, float* U);
void wino_s2_filter_grad(Stream& s, const WShape& w, const float* dU, float* dpacked);       // dU[25][4 * Cf][Cc]
void wino_filter_grad(Stream& s, int m, int r, const WShape& w, const float* dU, float* dpacked);          // dU[P][Cip][Npad]

// ---- InstanceNorm / activation / dropout -------------------------------------------
struct NormActArgs {
  TView x;                    // raw conv output (dense)
  TView y;                    // destination (may be a channel slice)
  float* stats = nullptr;     // [N][C][2] (mean, rstd); written when norm
  int norm = 1;
  int act = ACT_NONE;
  float drop_p = 0.f;         // 0 => no dropout
  uint64_t seed = 0;          // dropout stream for this call (ignored if drop_p==0)
  const TView* residual = nullptr;   // y = ... + residual
  float* amax_out = nullptr;  // optional amax slot of y (see wino_input_transform)
  // captured training step (engine.h Model::step_captured): the step seed is read from device memory, so that one recorded
  // launch sequence serves every step -- the kernels use *seed_base * 0x9E3779B1 + salt (= Net::drop_seed) instead of `seed`
  const uint64_t* seed_base = nullptr;
  uint64_t salt = 0;
  // optional: the statistics' partial sums as the producing conv's epilogue left them (ConvFwdArgs::stat_partial):
  // [N][partial_chunks][x.C][2] fp64.  The statistics pass over x is skipped: finalize + apply only.
  const double* partial_in = nullptr;
  int partial_chunks = 0;
};
void norm_act_fwd(Stream& s, const NormActArgs& a);

struct NormActBwdArgs {
  TView dy;                   // grad wrt y (view, same geometry as y)
  TView x;                    // raw conv output saved by forward (norm) or y itself (!norm)
  const float* stats = nullptr;
  TView dx;                   // grad wrt raw conv output (dense), overwritten
  int norm = 1;
  int act = ACT_NONE;
  float drop_p = 0.f;
  uint64_t seed = 0;
  // optional [N][C] (fp64): per-image column sums of dx -- the bias gradient of the conv that produced x, for free while the
  // slab is in registers.  Written only by launches for which norm_act_bwd_emits_colsum(H * W, C) holds.
  double* colsum = nullptr;
  float* amax_out = nullptr;  // optional amax slot of dx
  const uint64_t* seed_base = nullptr;   // as NormActArgs
  uint64_t salt = 0;
};
void norm_act_bwd(Stream& s, const NormActBwdArgs& a);
bool norm_act_bwd_emits_colsum(int HW, int C);
// db[c] = sum_n partial[n][c]   (fixed order)
void bias_grad_from_colsums(Stream& s, const double* partial, int N, int C, float* db);

// The keep/scale factor norm_act_fwd / norm_act_bwd apply at a dropout site, written out as an NCHW tensor
// (N,C,H,W): element (n,c,h,w) = drop_scale(seed, ((n*H*W + h*W + w)*C + c), p) -- 0 or 1/(1-p).
// Diagnostic export (swn_model_dropout_mask): parity tests hand it to the oracle so that a train-mode
// step is compared value for value.
void dropout_mask(Stream& s, int N, int H, int W, int C, float p, uint64_t seed, float* out_nchw);

// The branch every piecewise-linear op took, written out as NCHW bytes (diagnostic export, swn_model_act_pattern).
// LeakyReLU / ReLU: 1 where the activation OUTPUT y is > 0 -- the side act_bwd / norm_act_bwd differentiate on.
// MaxPool2d(2,2): the window position (2*kh + kw) the backward pass routes the gradient to (first maximum in scan
// order).  An fp32 evaluation in another summation order puts pre-activations within round-off of zero on the other
// side; each such flip changes the gradient by O(1) of that element, which is what bounds any fp32-vs-fp64 gradient
// comparison at ~1e-3.  Parity tests replay these patterns in the float64 oracle and compare at ~1e-5 instead.
void act_pattern(Stream& s, const TView& y, uint8_t* out_nchw);
void pool_pattern(Stream& s, const TView& x, const TView& y, uint8_t* out_nchw);

// y = act(x) elementwise on views; bwd: dx (+)= dy * act'  (derivative expressed through the
// activation OUTPUT y: lrelu y>0?1:.2, relu y>0, tanh 1-y^2)
void act_fwd(Stream& s, const TView& x, const TView& y, int act, float* amax_out = nullptr);     // amax_out: fold max|y| into the slot
void act_bwd(Stream& s, const TView& dy, const TView& y, const TView& dx, int act, int accumulate, float* amax_out = nullptr);
// dst (+)= alpha * src + shift   (views)
void axpy(Stream& s, const TView& src, const TView& dst, float alpha, int accumulate, float shift = 0.f, float* amax_out = nullptr);

// ---- resampling / gather ---------------------------------------------------------------
void upsample_nearest_fwd(Stream& s, const TView& x, const TView& y, int factor, float* amax_out = nullptr);
void upsample_nearest_bwd(Stream& s, const TView& dy, const TView& dx, int factor, int accumulate);
void maxpool2_fwd(Stream& s, const TView& x, const TView& y, float* amax_out = nullptr);
void maxpool2_bwd(Stream& s, const TView& dy, const TView& x, const TView& y, const TView& dx, int accumulate);
// legacy RoIAlign (torchvision 0.4.0), sampling_ratio 1, spatial_scale 1.  tex: (B,H,W,C);
// rois: device float [B*R][4] = x1,y1,x2,y2 (batch index = k / R); out: (B,PH,PW,R*C) with
// channel index r*C + c (the reference's view(B,-1,PH,PW), swapnet_modules.py:237-240).
void roi_align_fwd(Stream& s, const TView& tex, int C, const float* rois, int R, const TView& out);
// bit-exact debug dump of the integer part: idx[k][ph][pw][4] = yl,yh,xl,xh ; valid[k][ph][pw]
void roi_align_indices(Stream& s, const float* rois, int K, int H, int W, int PH, int PW,
                       int32_t* idx, uint8_t* valid);

// Per-channel geometric augmentation of the warp dataloader (datasets/warp_dataset.py:131-137 ->
// datasets/data_utils.py:346-361: every one of the 19 cloth channels of every sample goes through its own random
// PIL transform chain) as one device gather.  src / dst: (B, C, H, W) fp32 NCHW.  maps: device doubles
// [B*C][nmaps][9] = { kind, c0..c7 }, applied to a channel in index order (map 0 first), each with PIL's NEAREST rule:
//   kind 0 -- identity;
//   kind 1 -- affine in Pillow's 16.16 fixed point (Geometry.c affine_fixed): c0..c5 = the six FIXed integers
//             a0,a1,a2',a3,a4,a5' (a2', a5' already include the half-pixel terms): xin = (a2' + a1 y + a0 x) >> 16,
//             yin = (a5' + a4 y + a3 x) >> 16 -- flips and RandomAffine, bit-identical to Image.transform(AFFINE, NEAREST);
//   kind 2 -- perspective (Geometry.c perspective_transform): xin = floor((c0 xc + c1 yc + c2) / (c6 xc + c7 yc + 1)),
//             xc = x + .5 (same for y with c3..c5); -1 when negative.
// A source coordinate outside the image at ANY step yields 0 (PIL's fill), like the sequential PIL calls.
void affine_gather(Stream& s, const float* src, float* dst, int B, int C, int H, int W, const double* maps, int nmaps);

// layout conversion at the Python boundary (NCHW fp32 <-> NHWC view)
void nchw_to_nhwc(Stream& s, const float* src, int N, int C, int H, int W, const TView& dst);
void nhwc_to_nchw(Stream& s, const TView& src, float* dst, int C);
// integer work
void decode_labels(Stream& s, const TView& x, int C, uint8_t* rgb_nchw);            // util/decode_labels.py
void argmax_labels(Stream& s, const TView& x, int C, int32_t* labels);              // data_utils.py:322
void labels_to_onehot(Stream& s, const int32_t* labels, const TView& y, int C);     // data_utils.py:330-343

// ---- losses: each writes the plain MEAN loss into *loss_out (device float) and, if a grad
// view is given, scale * d(mean loss)/dx into it (scale carries lambda and the 0.5 of loss_D).
// BCEWithLogits(pred, label) mean over N*H*W of channel 0;  dpred = scale*(sigmoid-t)/numel
// label_dev (optional): the label is read from device memory instead (captured training step)
void bce_logits_loss(Stream& s, const TView& pred, float label, float scale, float* loss_out,
                     const TView* dpred, const float* label_dev = nullptr);
void lsgan_loss(Stream& s, const TView& pred, float label, float scale, float* loss_out, const TView* dpred,
                const float* label_dev = nullptr);
void wgan_loss(Stream& s, const TView& pred, float sign, float scale, float* loss_out, const TView* dpred);
// CrossEntropy(logits, argmax_c(target)) mean over pixels; dlogits (+)= scale*(softmax-onehot)/P
void ce_argmax_loss(Stream& s, const TView& logits, const TView& target, int C, float scale,
                    float* loss_out, const TView* dlogits, int accumulate);
void l1_loss(Stream& s, const TView& a, const TView& b, int C, float scale, float* loss_out,
             const TView* da, int accumulate);
// content term of PerceptualLoss for one VGG slice: MSE of channel-L2-normalised features
void normed_mse_loss(Stream& s, const TView& f, const TView& t, float scale, float* loss_out,
                     const TView* df, int accumulate);
// style term: scale * MSE(Gram(a), Gram(b)), Gram over (N*C) x (H*W) of the raw images
// Under data parallelism a, b are the GLOBAL batches (all ranks' images, gathered by the host) and the gradient is
// produced for the nloc samples starting at n0 only (da: a view of those samples); n0 = 0, nloc < 0: whole batch.
void gram_style_loss(Stream& s, const TView& a, const TView& b, int C, float scale, float* loss_out,
                     const TView* da, int accumulate, int n0 = 0, int nloc = -1);
// ---- gradient penalty (modules/loss.py:133-184; wgan-gp / dragan-gp / dragan-lp), see gp.cpp ---------------------
// x_hat = a + alpha[n] * (b - a) per sample n.  b = the second view (wgan: the conditioned fakes) or, when b is NULL
// (dragan), a + half_std[0] * beta with beta ~ U[0,1) the same shape as a (NHWC view; pad channels 0).
void gp_interpolate(Stream& s, const TView& a, const TView* b, const float* alpha, const TView* beta, const float* half_std,
                    const TView& out);
// out[0] = 0.5 * unbiased std of the `numel` logical elements of view a (pad channels hold zeros and add nothing)
void gp_half_std(Stream& s, const TView& a, size_t numel, float* out);
// per-sample L2 norm of g over all its elements; penalty = mean_n (norm_n - 1)^2 (lp != 0: max(0, norm_n - 1)^2);
// loss_out[0] = penalty, u = scale * d(penalty)/dg   (same geometry as g)
void gp_penalty(Stream& s, const TView& g, int lp, float scale, float* loss_out, const TView& u);
// fills a view with U[0,1) from the library's counter RNG: beta of dragan.  Channels c >= Clog and, when a device channel map
// is given (WShape::cimap of the layer that reads the buffer), channels with cimap[c] < 0 are layout pads and stay 0 -- a
// non-zero pad would reach the pad rows of the first conv's weight gradient.
void gp_uniform(Stream& s, const TView& v, int Clog, uint64_t seed, const int32_t* cimap = nullptr);
// Second-order step through y = act(InstanceNorm(x)) (reverse over reverse).  The first backward computed
// gx = IN'(x)^T (act'(xh) * gy).  Given u = adjoint of gx:  uy = act'(xh) * IN'(x) u   (adjoint of gy; IN' is symmetric)
// and ax = (d gx / d x)^T u (adjoint of the raw activation x):
//   ax = -rstd^2 [ xh (<u gm> - <u><gm> - <u xh><gm xh>) + <gm xh> (u - <u> - xh <u xh>) + <u xh> (gm - <gm> - xh <gm xh>) ],
//   gm = act'(xh) * gy, <.> = mean over the pixels of one (n, c) plane.  All reductions in fp64.
struct NormActBwd2Args {
  TView u, gy, x;             // adjoint of gx; gradient w.r.t. y from the first backward; raw activation
  const float* stats = nullptr;
  TView uy, ax;               // outputs (overwritten)
  int act = ACT_NONE;
};
void norm_act_bwd2(Stream& s, const NormActBwd2Args& a);

// out[0] = a[0]*ca + b[0]*cb (device scalars; used to combine loss terms without a sync)
void scalar_axpby(Stream& s, const float* a, float ca, const float* b, float cb, float* out);

// ---- optimizer / parameter layout ------------------------------------------------------
struct AdamWArgs {
  float* p; const float* g; float* m; float* v; size_t n;
  float lr, beta1, beta2, eps, weight_decay;
  int step;   // 1-based
  // optional (captured training step): {lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step)} of THIS step in device memory, as
  // adamw_schedule computes them; `step` is then ignored
  const float* sched_dev = nullptr;
};
void adamw_step(Stream& s, const AdamWArgs& a);
void adamw_schedule(float lr, float beta1, float beta2, int step, float out[2]);

enum WKind : int { WK_CONV = 0, WK_CONVT = 1 };
// Packed weight layouts (all [K][Npad], row-major):
//  conv  (Co,Ci,KH,KW) : k = (kh*KW+kw)*Cip + ci , n = co
//  convT (Ci,Co,4,4)   : 4 phase blocks p=(a*2+b) of k = (dy*2+dx)*Cip + ci , n = co,
//                        tap (ky,kx) = (3-a-2dy, 3-b-2dx)
struct WShape {
  int kind, Co, Ci, KH, KW, Cip, Npad;
  // optional device table [Cip]: buffer channel -> reference input channel (or -1 = zero pad);
  // used where the NHWC buffer orders a torch.cat differently (D's conditional input).
  const int32_t* cimap = nullptr;
};
size_t packed_elems(const WShape& w);
void pack_weight(Stream& s, const WShape& w, const float* nchw, float* packed);
void unpack_weight(Stream& s, const WShape& w, const float* packed, float* nchw);
// dgrad operand derived from the forward-packed weights.  mode:
//  0 conv stride-2 k4 : 4 phase blocks [ (dy*2+dx)*Cop + co ][ci]        (Ndg = Ci)
//  1 conv stride-1    : [ ((KH-1-kh)*KW + (KW-1-kw))*Cop + co ][ci]
//  2 convT k4 s2      : [ (ky*4+kx)*Cop + co ][ci]
//  3 tail (x2 nearest upsample + ZeroPad(1,0,1,0) + conv k4 p1): 5x5 stride-2 effective
//    kernel  [ (r*5+c)*Cop + co ][ci] = sum_{a-ky+3=r, b-kx+3=c; a,b in {0,1}} W[ky][kx]
void repack_dgrad(Stream& s, const WShape& w, int mode, int Cop, int Ndgpad, const float* packed, float* dg);
size_t dgrad_elems(const WShape& w, int mode, int Cop, int Ndgpad);
// Tail conv (x2 nearest upsample + ZeroPad(1,0,1,0) + conv k4 p1, swapnet_modules.py:85-90) folded
// onto the un-upsampled input: output phase (a,b) = (y&1, x&1) is a (2+a)x(2+b) conv with
// pre-summed taps  r(a,ky) = a ? (ky+1)>>1 : ky>>1  (25 instead of 64 taps per 2x2 outputs).
// Layout: 4 phase blocks p = a*2+b, each [ (r*(2+b)+c)*Cip + ci ][Npad], at tail_fold_offset(p).
size_t tail_fold_offset(const WShape& w, int phase);       // in floats; phase 4 = total size
void tail_fold_weights(Stream& s, const WShape& w, const float* packed, float* folded);
// inverse for the weight gradient: dW[ky][kx] = sum over the 4 phases of dWfold[p][r(a,ky)][c(b,kx)]
void tail_unfold_wgrad(Stream& s, const WShape& w, const float* dfolded, float* dpacked);

// ---- PatchGAN's 1-channel k4 s1 p1 head conv (modules/discriminators.py:131) as "taps on the N axis" ----
// An implicit GEMM with N = 1 reads every input element 16 times (once per tap) for one MAC each: it is
// bound by the load path.  Instead Z[q][t] = sum_c x[q][c] w[t][c] (a 1x1 conv, N = 16 taps, x read
// once), then y[n,oy,ox] = b + sum_{kh,kw} Z[n, oy-1+kh, ox-1+kw][4 kh + kw].  Backward uses the adjoint
// gather dZ[n,iy,ix][4 kh + kw] = dY[n, iy+1-kh, ix+1-kw], dW = X^T dZ, dX = dZ W.
// wt: [Cip][16], wt2: [16][Cip] (both derived from the packed [(t*Cip + c)][Npad] weight, column 0)
void head_pack(Stream& s, const WShape& w, const float* packed, float* wt, float* wt2);
void head_unpack_grad(Stream& s, const WShape& w, const float* dwt, float* dpacked);     // dpacked column 0; pads zeroed
void head_gather(Stream& s, const TView& z, const float* bias, const TView& y);          // z.C = 16, y = (N, H-1, W-1, 4)
void head_scatter(Stream& s, const TView& dy, const TView& dz);

}  // namespace swn
