"""Thin Python handles over the C-ABI objects (swn_ctx / swn_model).

torch is used here only for device memory (tensors whose raw pointers are handed to the
library) and, in swapnet_amd.parallel, for torch.distributed.  No arithmetic of the hot path
happens in torch.
"""
import ctypes as C
import os
from collections import OrderedDict

import torch

from . import _C

NET_G, NET_D, NET_VGG = 0, 1, 2
W_WEIGHT, W_GRAD, W_EXP_AVG, W_EXP_AVG_SQ = 0, 1, 2, 3


class Context:
    """One per process / GPU (models/base_model.py:36-40 picks the device in the reference)."""

    def __init__(self, device=None, lib=None, workspace_mb=512, use_torch_stream=True):
        self.lib = lib or _C.lib()
        if self.lib.is_device:
            if not torch.cuda.is_available():
                raise _C.SwapnetHipError("no HIP device visible; swapnet_amd has no CPU path")
            self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
            torch.cuda.set_device(self.device)
            # enqueue on torch's current stream: H2D copies, allocator reuse of the temporaries whose
            # pointers we borrow, and RCCL collectives are then ordered with our kernels for free
            self.torch_stream = torch.cuda.current_stream(self.device)
            stream = C.c_void_p(self.torch_stream.cuda_stream)
            if os.environ.get("SWAPNET_OWN_STREAM") == "1":     # (experiment: the context's main stream is its own, not torch's current one)
                use_torch_stream = False
            dev_index, create = self.device.index, int(not use_torch_stream)
        else:                       # CI host simulator (tests only)
            self.device = torch.device("cpu")
            stream, dev_index, create = None, 0, 0
        h = C.c_void_p()
        self.lib.call("swn_ctx_create", dev_index, stream, create, C.c_size_t(workspace_mb << 20), C.byref(h))
        self.handle = h
        self._copy_stream = None

    def upload(self, tensor, dtype, key=None):
        """Host batch -> device on a dedicated copy stream: `set_input` is called right after
        `optimize_parameters` returned (train.py:62-64) while that step's kernels are still running, so the
        H2D transfer (335 MB/step for one-hot cloths at bs 32) hides under them instead of queueing
        behind them.  With `key` the destination is one of two persistent staging buffers per input slot
        (no allocator traffic -- a fresh hipMalloc per step synchronises the device); the copy stream
        waits for the kernel that last read the buffer it is about to overwrite (`consumed`), the main
        stream for the copy.  Returns (device tensor, token for `consumed`)."""
        t = tensor.detach()
        if self.device.type != "cuda" or t.is_cuda:
            return t.to(device=self.device, dtype=dtype).contiguous(), None
        t = t.to(dtype=dtype).contiguous()
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
            self._staging = {}
        cs, main = self._copy_stream, torch.cuda.current_stream(self.device)
        if key is None:
            with torch.cuda.stream(cs):
                d = t.to(self.device, non_blocking=True)
            main.wait_stream(cs)
            d.record_stream(main)
            return d, None
        st = self._staging.setdefault(key, {"bufs": [None, None], "events": [None, None], "i": 0})
        i = st["i"]
        st["i"] ^= 1
        buf = st["bufs"][i]
        if buf is None or buf.shape != t.shape or buf.dtype != dtype:
            buf = torch.empty(t.shape, dtype=dtype, device=self.device)
            st["bufs"][i], st["events"][i] = buf, None
        if st["events"][i] is not None:
            cs.wait_event(st["events"][i])
        with torch.cuda.stream(cs):
            buf.copy_(t, non_blocking=True)
        main.wait_stream(cs)
        return buf, (key, i)

    def consumed(self, token):
        """The kernel reading an `upload(..., key=)` buffer has been enqueued on the current stream."""
        if token is None:
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._staging[token[0]]["events"][token[1]] = ev

    def set_overlap(self, on):
        """Side stream on/off (results identical; off gives un-overlapped per-kernel timings)."""
        self.lib.call("swn_ctx_set_overlap", self.handle, int(bool(on)))

    def sync(self):
        self.lib.call("swn_ctx_sync", self.handle)

    def attach_comm(self, fn_ptr, comm, world):
        """swn_ctx_attach_comm: hand the library an all-reduce entry point (address of a function with ncclAllReduce's signature)
        and the communicator to call it on; fn_ptr None detaches.  See swapnet_amd.parallel.NativeComm."""
        self.lib.call("swn_ctx_attach_comm", self.handle, C.c_void_p(fn_ptr) if fn_ptr else None,
                      comm if comm is not None else None, int(world))

    def route_trace(self, on):
        """Start (clears the log) / stop recording which kernel every layer launches (swn_route_trace)."""
        self.lib.call("swn_route_trace", int(bool(on)))

    def route_report(self):
        """The distinct "<layer> <phase> <kernel[shape, schedule]>" lines recorded since route_trace(True), in first-use order."""
        need = self.lib.dll.swn_route_report(None, 0)
        buf = C.create_string_buffer(need + 16)
        self.lib.dll.swn_route_report(buf, need + 16)
        return [l for l in buf.value.decode().split("\n") if l]

    def bytes_allocated(self):
        n = C.c_size_t()
        self.lib.call("swn_ctx_bytes_allocated", self.handle, C.byref(n))
        return n.value

    def close(self):
        if getattr(self, "handle", None):
            comm = getattr(self, "native_comm", None)       # the communicator the library's exchange runs on dies with its context,
            if comm is not None:                            # before it: parallel.NativeComm registers itself here
                self.native_comm = None
                comm.close()
            self.lib.call("swn_ctx_destroy", self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=None, lib=None):
    key = (id(lib) if lib is not None else 0, str(device))
    if key not in _default_ctx:
        _default_ctx[key] = Context(device=device, lib=lib)
    return _default_ctx[key]


class NativeModel:
    """swn_model handle: generator (+ discriminator, optimizers) living in library-owned
    NHWC arenas.  Tensors cross the boundary as NCHW fp32, exactly as the reference holds them."""

    def __init__(self, ctx, kind, batch, height, width, is_train=True, dropout=0.5, num_roi=12, body_channels=3,
                 cloth_channels=19, n_layers_D=3, share=None):
        """share = another NativeModel of the same kind: the new model uses ITS parameter arenas (weights, gradients, Adam
        moments, step counters) and owns only activations -- the model of another batch size on the same training state
        (swn_model_create_shared)."""
        self.ctx, self.lib, self.kind = ctx, ctx.lib, kind
        self.B, self.H, self.W, self.is_train = batch, height, width, is_train
        self.body_channels, self.cloth_channels = body_channels, cloth_channels
        self.n_layers_D = int(n_layers_D)
        self.share = share
        if share is not None:
            if share.kind != kind:
                raise ValueError("a sharing model must be of its sharer's kind")
            h = C.c_void_p()
            self.lib.call("swn_model_create_shared", share.handle, batch, height, width, C.byref(h))
            self.is_train, self.body_channels, self.cloth_channels = share.is_train, share.body_channels, share.cloth_channels
            self.n_layers_D, self.out_channels = share.n_layers_D, share.out_channels
            self.handle = h
            self._keep = []
            return
        # a context-level option read at model construction (define_D's n_layers_D, base_gan.py:147): set it for this model,
        # whatever an earlier model on the same context asked for
        self.lib.call("swn_ctx_set_patchgan_layers", ctx.handle, self.n_layers_D)
        h = C.c_void_p()
        if kind == "warp":
            self.lib.call("swn_warp_model_create_ex", ctx.handle, batch, height, width, int(is_train),
                          C.c_float(dropout), body_channels, cloth_channels, C.byref(h))
            self.out_channels = cloth_channels
        elif kind == "texture":
            self.lib.call("swn_texture_model_create_ex", ctx.handle, batch, height, width, int(is_train), num_roi,
                          cloth_channels, C.byref(h))
            self.out_channels = 3
        else:
            raise ValueError("unknown model kind " + kind)
        self.handle = h
        self._keep = []

    # ---- parameters ---------------------------------------------------------------------
    def param_infos(self, net):
        n = C.c_int()
        self.lib.call("swn_model_param_count", self.handle, net, C.byref(n))
        out = OrderedDict()
        buf = C.create_string_buffer(256)
        for i in range(n.value):
            shape = (C.c_int * 4)()
            nd = C.c_int()
            self.lib.call("swn_model_param_info", self.handle, net, i, buf, 256, C.byref(shape), C.byref(nd))
            out[buf.value.decode()] = tuple(shape[: nd.value])
        return out

    def _dev(self, t):
        return t.detach().to(device=self.ctx.device, dtype=torch.float32).contiguous()

    def set_param(self, net, name, tensor, which=W_WEIGHT):
        t = self._dev(tensor)
        self.lib.call("swn_model_param_set", self.handle, net, which, name.encode(), _C.ptr(t))
        self._sync_if_needed()

    def get_param(self, net, name, shape, which=W_WEIGHT):
        t = torch.empty(shape, dtype=torch.float32, device=self.ctx.device)
        self.lib.call("swn_model_param_get", self.handle, net, which, name.encode(), _C.ptr(t))
        self._sync_if_needed()
        return t

    def _sync_if_needed(self):
        # the library runs on torch's current stream, so torch frees/reuses are ordered; a
        # private stream would need an explicit sync before temporaries die
        pass

    def load_state_dict(self, net, sd, which=W_WEIGHT, strict=True):
        infos = self.param_infos(net)
        missing = [k for k in infos if k not in sd]
        unexpected = [k for k in sd if k not in infos]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {unexpected}")
        for k, shape in infos.items():
            if k in sd:
                if tuple(sd[k].shape) != tuple(shape):
                    raise RuntimeError(f"size mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(shape)}")
                self.set_param(net, k, sd[k], which)
        self.ctx.sync()

    def state_dict(self, net, which=W_WEIGHT, to_cpu=False):
        out = OrderedDict()
        for k, shape in self.param_infos(net).items():
            t = self.get_param(net, k, shape, which)
            out[k] = t.cpu() if to_cpu else t
        self.ctx.sync()
        return out

    def optim_step_count(self, net, value=None):
        if value is None:
            n = C.c_int()
            self.lib.call("swn_model_optim_step_get", self.handle, net, C.byref(n))
            return n.value
        self.lib.call("swn_model_optim_step_set", self.handle, net, int(value))

    def set_hyper(self, **kw):
        h = _C.SwnHyper(lr=1e-4, d_lr=4e-4, weight_decay=0.0, d_weight_decay=0.01, b1=0.9, b2=0.999,
                        lambda_gan=1.0, lambda_ce=100.0, lambda_l1=10.0, lambda_content=20.0,
                        lambda_style=1e-8, gan_mode=0, warp_mode_ce=0, grad_scale=1.0, d_b1=-1.0, d_b2=-1.0,
                        gp_mode=0, lambda_gp=10.0)
        for k, v in kw.items():
            setattr(h, k, v)
        self.lib.call("swn_model_set_hyper", self.handle, C.byref(h))

    # ---- data / step ----------------------------------------------------------------------
    def set_input(self, slot, tensor):
        t, token = self.ctx.upload(tensor, torch.float32, key=(id(self), "in", slot))
        if t.dim() == 3:                       # rois (B,R,4)
            n, c, h, w = t.shape[0], t.shape[1], t.shape[2], 1
        else:
            n, c, h, w = t.shape
        self.lib.call("swn_model_set_input", self.handle, slot, _C.ptr(t), n, c, h, w)
        self.ctx.consumed(token)
        self._keep = [t]

    def set_input_labels(self, slot, labels):
        """Integer cloth label map (B,H,W) -> one-hot expansion on the device."""
        t, token = self.ctx.upload(labels, torch.int32, key=(id(self), "lab", slot))
        n, h, w = t.shape
        self.lib.call("swn_model_set_input_labels", self.handle, slot, _C.ptr(t), n, h, w)
        self.ctx.consumed(token)
        self._keep_labels = t

    def forward(self, training=False, seed=0):
        self.lib.call("swn_model_forward", self.handle, int(training), C.c_uint64(seed))

    def dropout_masks(self, net=NET_G, seed=0):
        """The keep/scale factors (0 or 1/(1-p)) a training-mode pass with `seed` applies at every dropout
        site of `net`, in forward order, as NCHW tensors (diagnostic export for the parity tests)."""
        n = C.c_int()
        self.lib.call("swn_model_dropout_sites", self.handle, net, C.byref(n))
        out = []
        for i in range(n.value):
            shape, p = (C.c_int * 4)(), C.c_float()
            self.lib.call("swn_model_dropout_mask", self.handle, net, i, C.c_uint64(seed), None, C.byref(shape), C.byref(p))
            t = torch.empty(tuple(shape), dtype=torch.float32, device=self.ctx.device)
            self.lib.call("swn_model_dropout_mask", self.handle, net, i, C.c_uint64(seed), _C.ptr(t), C.byref(shape), C.byref(p))
            out.append((t, p.value))
        self.ctx.sync()
        return out

    def act_patterns(self, net=NET_G):
        """The branch every LeakyReLU / ReLU / MaxPool of `net` took in the last pass, in forward order: a list of
        (kind, uint8 NCHW tensor on the CPU), kind 1 = output > 0, kind 2 = arg-max window position (diagnostic export:
        the parity tests replay the pattern in the float64 oracle).  net: 0 G, 1 D of the D step ([fake | real]),
        2 D inside the G step, 3 VGG16 on the generated image (texture)."""
        n = C.c_int()
        self.lib.call("swn_model_act_sites", self.handle, net, C.byref(n))
        out = []
        for i in range(n.value):
            shape, kind = (C.c_int * 4)(), C.c_int()
            self.lib.call("swn_model_act_pattern", self.handle, net, i, None, C.byref(shape), C.byref(kind))
            t = torch.empty(tuple(shape), dtype=torch.uint8, device=self.ctx.device)
            self.lib.call("swn_model_act_pattern", self.handle, net, i, _C.ptr(t), C.byref(shape), C.byref(kind))
            self.ctx.sync()
            out.append((kind.value, t.cpu()))
        return out

    def set_style_context(self, all_out, all_tgt, n0):
        """Global (all ranks') generated / target images for the style term of the next backward_G (data parallel)."""
        o = all_out.detach().to(device=self.ctx.device, dtype=torch.float32).contiguous()
        t = all_tgt.detach().to(device=self.ctx.device, dtype=torch.float32).contiguous()
        self.lib.call("swn_model_set_style_context", self.handle, _C.ptr(o), _C.ptr(t), int(o.shape[0]), int(n0))
        self._style_keep = (o, t)

    def set_gp_random(self, alpha=None, beta=None):
        """alpha (B,) / (B,1,1,1) and beta (B,C_D,H,W): the gradient-penalty draws of the next backward_D (one-shot)."""
        a = None if alpha is None else alpha.detach().reshape(-1).to(device=self.ctx.device, dtype=torch.float32).contiguous()
        b = None if beta is None else beta.detach().to(device=self.ctx.device, dtype=torch.float32).contiguous()
        self.lib.call("swn_model_set_gp_random", self.handle, _C.ptr(a), _C.ptr(b))
        self._gp_keep = (a, b)

    def discriminate(self, x):
        """NLayerDiscriminator.forward on a conditioned input in the reference's channel order (B,C_D,H,W),
        C_D = body + cloth channels (warp, 22 by default) / texture + cloth channels (texture)."""
        xd = x.detach().to(device=self.ctx.device, dtype=torch.float32).contiguous()
        cd = self.cloth_channels + (self.body_channels if self.kind == "warp" else 3)
        if tuple(xd.shape) != (self.B, cd, self.H, self.W):
            raise ValueError("discriminator input must be (%d, %d, %d, %d), got %s" % (self.B, cd, self.H, self.W, tuple(xd.shape)))
        k = self.n_layers_D                     # n stride-2 levels, then two 4x4 stride-1 convs with padding 1 (-1 pixel each)
        shape = (self.B, 1, (self.H >> k) - 2, (self.W >> k) - 2) if k > 0 else (self.B, 1, self.H, self.W)     # 0: PixelDiscriminator
        pred = torch.empty(shape, dtype=torch.float32, device=self.ctx.device)
        self.lib.call("swn_model_discriminate", self.handle, _C.ptr(xd), _C.ptr(pred))
        self.ctx.sync()
        return pred

    def perceptual(self, output, target, use_style=True, content_w=1.0, style_w=1.0, want_grad=False):
        """PerceptualLoss.forward: returns (losses[2] device tensor = content, style ; d_output or None)."""
        o = output.detach().to(device=self.ctx.device, dtype=torch.float32).contiguous()
        t = target.detach().to(device=self.ctx.device, dtype=torch.float32).contiguous()
        if tuple(o.shape) != (self.B, 3, self.H, self.W) or o.shape != t.shape:
            raise ValueError("perceptual loss inputs must both be (%d, 3, %d, %d)" % (self.B, self.H, self.W))
        out2 = torch.empty(2, dtype=torch.float32, device=self.ctx.device)
        d = torch.empty_like(o) if want_grad else None
        self.lib.call("swn_model_perceptual", self.handle, _C.ptr(o), _C.ptr(t), int(bool(use_style)), _C.ptr(out2),
                      C.c_float(content_w), C.c_float(style_w), _C.ptr(d))
        self.ctx.sync()
        return out2, d

    def output(self, slot=0):
        t = torch.empty((self.B, self.out_channels, self.H, self.W), dtype=torch.float32, device=self.ctx.device)
        self.lib.call("swn_model_get_output", self.handle, slot, _C.ptr(t))
        self.ctx.sync()
        return t

    def tap(self, net, name):
        shape = (C.c_int * 4)()
        self.lib.call("swn_model_get_tap", self.handle, net, name.encode(), None, C.byref(shape))
        t = torch.empty(tuple(shape), dtype=torch.float32, device=self.ctx.device)
        self.lib.call("swn_model_get_tap", self.handle, net, name.encode(), _C.ptr(t), C.byref(shape))
        self.ctx.sync()
        return t

    def tap_grad(self, net, name):
        """Gradient w.r.t. a named activation after the last backward pass (diagnostics)."""
        shape = (C.c_int * 4)()
        self.lib.call("swn_model_get_tap_grad", self.handle, net, name.encode(), None, C.byref(shape))
        t = torch.empty(tuple(shape), dtype=torch.float32, device=self.ctx.device)
        self.lib.call("swn_model_get_tap_grad", self.handle, net, name.encode(), _C.ptr(t), C.byref(shape))
        self.ctx.sync()
        return t

    def backward_D(self, label_fake, label_real):
        self.lib.call("swn_model_backward_D", self.handle, C.c_float(label_fake), C.c_float(label_real))

    def backward_G(self, label_real):
        self.lib.call("swn_model_backward_G", self.handle, C.c_float(label_real))

    def backward_G_parts(self):
        """Number of buckets backward_G_part splits the generator backward into."""
        n = C.c_int()
        self.lib.call("swn_model_backward_G_parts", self.handle, C.byref(n))
        return n.value

    def backward_G_part(self, label_real, part):
        """Returns (offset, count) of the generator gradient-arena range that is final after this part."""
        off, cnt = C.c_size_t(), C.c_size_t()
        self.lib.call("swn_model_backward_G_part", self.handle, C.c_float(label_real), part, C.byref(off), C.byref(cnt))
        return off.value, cnt.value

    def optimizer_step(self, net):
        self.lib.call("swn_model_optimizer_step", self.handle, net)

    def optimizer_step_range(self, net, off, count, first):
        self.lib.call("swn_model_optimizer_step_range", self.handle, net, C.c_size_t(off), C.c_size_t(count), int(bool(first)))

    def step(self, labels, training=True, seed=0, captured=False):
        """One G+D optimize_parameters step.  captured=True: the hipGraph form (swn_model_step_captured) -- first call eager,
        second records, later ones replay; bit-identical results."""
        arr = (C.c_float * 3)(*[float(x) for x in labels])
        self.lib.call("swn_model_step_captured" if captured else "swn_model_step", self.handle, C.byref(arr), int(training),
                      C.c_uint64(seed))

    def step_dp(self, labels, training=True, seed=0, after_forward=False):
        """One data-parallel G+D step with the exchange owned by the library (swn_model_step_dp; the context needs a communicator:
        Context.attach_comm / parallel.NativeComm)."""
        arr = (C.c_float * 3)(*[float(x) for x in labels])
        self.lib.call("swn_model_step_dp", self.handle, C.byref(arr), int(training), C.c_uint64(seed), int(bool(after_forward)))

    def losses(self):
        buf = (C.c_float * len(_C.LOSS_NAMES))()
        self.lib.call("swn_model_get_losses", self.handle, C.cast(buf, C.POINTER(C.c_float)), len(_C.LOSS_NAMES))
        return OrderedDict(zip(_C.LOSS_NAMES, [float(x) for x in buf]))

    def grad_arena(self, net):
        """Flat fp32 view of the net's gradient arena as a torch tensor (zero-copy on GPU)."""
        p, n = C.c_void_p(), C.c_size_t()
        self.lib.call("swn_model_grad_arena", self.handle, net, C.byref(p), C.byref(n))
        return _wrap_pointer(p.value, n.value, self.ctx.device)

    def weight_arena(self, net):
        p, n = C.c_void_p(), C.c_size_t()
        self.lib.call("swn_model_weight_arena", self.handle, net, C.byref(p), C.byref(n))
        return _wrap_pointer(p.value, n.value, self.ctx.device)

    def arena(self, net, which):
        p, n = C.c_void_p(), C.c_size_t()
        self.lib.call("swn_model_arena", self.handle, net, which, C.byref(p), C.byref(n))
        return _wrap_pointer(p.value, n.value, self.ctx.device)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.call("swn_model_destroy", self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def op_gan_loss(ctx, pred, gan_mode, label, target_is_real, grad_scale=1.0, want_grad=True):
    """GANLoss value (+ gradient w.r.t. the prediction map) through swn_op_gan_loss.  Returns (loss[1], dpred)."""
    p = pred.detach().to(device=ctx.device, dtype=torch.float32).contiguous()
    shape = tuple(p.shape)
    p4 = p.reshape((shape + (1, 1, 1, 1))[:4]) if p.dim() < 4 else p
    n, c, h, w = p4.shape
    loss = torch.empty(1, dtype=torch.float32, device=ctx.device)
    d = torch.empty_like(p4) if want_grad else None
    ctx.lib.call("swn_op_gan_loss", ctx.handle, int(gan_mode), _C.ptr(p4), n, c, h, w, C.c_float(label),
                 int(bool(target_is_real)), C.c_float(grad_scale), _C.ptr(loss), _C.ptr(d))
    return loss, (d.reshape(shape) if d is not None else None)


def op_norm_act_bwd2(ctx, x, gy, u, act=1):
    """(uy, ax) of swn_op_norm_act_bwd2: the second-order step through act(InstanceNorm(x))."""
    d = [t.to(device=ctx.device, dtype=torch.float32).contiguous() for t in (x, gy, u)]
    uy, ax = torch.empty_like(d[0]), torch.empty_like(d[0])
    n, c, h, w = d[0].shape
    ctx.lib.call("swn_op_norm_act_bwd2", ctx.handle, _C.ptr(d[0]), _C.ptr(d[1]), _C.ptr(d[2]), n, c, h, w, int(act),
                 _C.ptr(uy), _C.ptr(ax))
    return uy, ax


def op_norm_act_dropout(ctx, x, dy=None, norm=True, act=2, p=0.5, seed=0):
    """[InstanceNorm] -> activation -> Dropout(p), training mode, forward (+ backward when dy is given) through
    swn_op_norm_act_dropout.  Returns (y, mask, dx): mask holds the factor (0 or 1/(1-p)) both passes applied."""
    xd = x.to(device=ctx.device, dtype=torch.float32).contiguous()
    dyd = None if dy is None else dy.to(device=ctx.device, dtype=torch.float32).contiguous()
    y, mask = torch.empty_like(xd), torch.empty_like(xd)
    dx = None if dy is None else torch.empty_like(xd)
    n, c, h, w = xd.shape
    ctx.lib.call("swn_op_norm_act_dropout", ctx.handle, _C.ptr(xd), _C.ptr(dyd), n, c, h, w, int(norm), int(act),
                 C.c_float(p), C.c_uint64(seed), _C.ptr(y), _C.ptr(mask), _C.ptr(dx))
    return y, mask, dx


class _ArrayIface:
    def __init__(self, ptr, n, cuda, typestr="<f4"):
        d = dict(shape=(n,), typestr=typestr, data=(ptr, False), version=2 if cuda else 3)
        if cuda:
            self.__cuda_array_interface__ = d
        else:
            self.__array_interface__ = d


def _wrap_pointer(ptr, n, device, typestr="<f4"):
    if device.type == "cuda":
        return torch.as_tensor(_ArrayIface(ptr, n, True, typestr), device=device)
    import numpy as np
    return torch.from_numpy(np.asarray(_ArrayIface(ptr, n, False, typestr)))


class NativePipeline:
    """swn_pipeline handle: warp forward -> argmax labels -> one-hot -> texture forward on the device, optionally
    replayed as a hipGraph (see include/swapnet_hip.h)."""

    def __init__(self, warp_model, texture_model):
        self.warp, self.texture = warp_model, texture_model
        self.lib, self.ctx = warp_model.lib, warp_model.ctx
        h = C.c_void_p()
        self.lib.call("swn_pipeline_create", warp_model.handle, texture_model.handle, C.byref(h))
        self.handle = h

    def run(self, use_graph=True):
        """Returns True when the call was a graph replay (False: eager run, incl. the capturing first call)."""
        replayed = C.c_int()
        self.lib.call("swn_pipeline_run", self.handle, int(bool(use_graph)), C.byref(replayed))
        return bool(replayed.value)

    def labels(self):
        p = C.c_void_p()
        self.lib.call("swn_pipeline_labels", self.handle, C.byref(p))
        n = self.warp.B * self.warp.H * self.warp.W
        self.ctx.sync()
        return _wrap_pointer(p.value, n, self.ctx.device, "<i4").reshape(self.warp.B, self.warp.H, self.warp.W).clone()

    def close(self):
        if getattr(self, "handle", None):
            self.lib.call("swn_pipeline_destroy", self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
