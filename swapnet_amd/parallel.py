"""Data parallelism: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm)
over xGMI.  The hot path shards over the batch axis: InstanceNorm and RoIAlign are per-sample and
every loss is a batch mean, so summing per-rank gradients pre-scaled by 1/world
reproduces the single-process big-batch step -- with ONE exception that needs data, not just gradients:
the texture stage's style term.  `gram_matrix` views the batch as (B*C, H*W)
(modules/losses/perceptual.py:6-10), so the Gram and its MSE couple the samples of ALL ranks
(SURVEY.md 8(e) caveat 1; not negligible: with the default lambda_style = 1e-8 a per-rank Gram moves
generator gradient tensors by >10 %).  `gather_style_context` therefore all-gathers the 3-channel
generated / target images (12.6 MB per rank) and the library evaluates the Gram over the global
batch, back-propagating into the local samples: tests/test_data_parallel.py checks that the
two-rank step equals the big-batch step with the term on and off.  The only other exchange is the
gradient of each optimizer's arena, which is ONE flat fp32 buffer per network (G 550 MB, D 11 MB) ->
bucketed all-reduces, no per-tensor launches.  The reference has no multi-GPU code at all; this is
new design."""
import ctypes as C
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment.  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # SWAPNET_DIST_BACKEND=gloo lets the N>1 code path be smoke-tested on a single-GPU box
        backend = backend or os.environ.get("SWAPNET_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def launched_data_parallel():
    """Opt-in data parallelism for the reference's UNCHANGED train.py: start it under torchrun
    (`python -m torch.distributed.run --nproc-per-node N train.py ...`, i.e. WORLD_SIZE > 1 in the environment) and
    every model it creates trains data-parallel -- rank r drives GPU LOCAL_RANK, the weights start from rank 0's, each
    rank shuffles its own way through the dataset, gradients are averaged over RCCL.  SWAPNET_DATA_PARALLEL=0 opts out
    (e.g. N independent runs under one launcher)."""
    return int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("SWAPNET_DATA_PARALLEL", "1") != "0"


def launch_device(gpu_id):
    """The device of this rank under launched_data_parallel(): torchrun's LOCAL_RANK (one process per GPU)."""
    return int(os.environ.get("SWAPNET_FORCE_DEVICE", os.environ.get("LOCAL_RANK", gpu_id if gpu_id is not None else 0)))


def broadcast_floats(values, src=0):
    """Rank `src`'s host scalars on every rank (the smooth-label draws of GANLoss: ranks that shuffle differently have
    consumed their RNGs differently, and all of them must train against the same targets, SURVEY.md 8(e) caveat 2)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return list(values)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.broadcast(t, src=src)
    return [float(v) for v in t.cpu()]


class GradExchange:
    """Averages a flat gradient arena across ranks.  `begin()` launches the collective on
    RCCL's own stream (it depends on the kernels already enqueued on the compute stream);
    `finish()` makes the compute stream wait for it -- work enqueued between the two calls
    (the other network's forward/backward) overlaps with the transfer."""

    def __init__(self, world=None, force=False, wire=None):
        """force=True issues the collectives even at world size 1 (a 1-rank RCCL all-reduce is an identity that still
        exercises pointer wrapping of the arena slices and the stream ordering with the library's side stream).
        wire = "bf16" (or SWAPNET_GRAD_WIRE=bf16; BASELINE.json C4 / C5 "bf16", SURVEY 8(d): 275 MB instead of 550 MB per generator
        exchange): a bucket travels as bfloat16 -- rounded to nearest on the device, summed by the collective in bfloat16, widened back
        into the fp32 gradient arena, AdamW and the master weights stay fp32.  An OPTION, never the parity configuration: the reference
        has no reduced-precision path, and the rounding (8 mantissa bits per gradient element, before the sum) is a property of this
        wire format, bounded by tests/test_data_parallel.py against the fp32 exchange."""
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.force = force and dist.is_initialized()
        self.wire = (wire or os.environ.get("SWAPNET_GRAD_WIRE", "f32")).lower()
        if self.wire not in ("f32", "bf16"):
            raise ValueError("GradExchange wire format must be f32 or bf16, got %r" % self.wire)
        self.bytes_sent = 0                     # what this rank handed to the collectives (the bench line reports it per step)
        self._pending = []

    def begin(self, flat):
        if self.world <= 1 and not self.force:
            return
        if self.wire == "bf16":
            buf = flat.to(torch.bfloat16)       # (torch elementwise kernel on the current stream, behind the bucket's gradients)
            self.bytes_sent += buf.numel() * 2
            self._pending.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True), flat, buf))
            return
        self.bytes_sent += flat.numel() * 4
        self._pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, None))

    @staticmethod
    def _land(work, flat, buf):
        work.wait()
        if buf is not None:
            flat.copy_(buf)                     # bfloat16 -> fp32 into the arena slice AdamW reads

    def wait_oldest(self):
        """Make the compute stream wait for the oldest outstanding collective only; returns its buffer (or None)."""
        if not self._pending:
            return None
        work, flat, buf = self._pending.pop(0)
        self._land(work, flat, buf)
        return flat

    def finish(self):
        # the mean needs no extra pass: every loss gradient is pre-scaled by 1/world
        # (swn_hyper.grad_scale), so SUM over ranks is already the average
        for work, flat, buf in self._pending:
            self._land(work, flat, buf)
        self._pending = []

    def allreduce_mean(self, flat):
        self.begin(flat)
        self.finish()


def gather_style_context(model, targets_local):
    """Texture stage under data parallelism: all-gather the generated and the target images of every rank and hand them
    to the library so that the style term's Gram spans the global batch (swn_model_set_style_context).  Call after
    forward, before backward_G.  B x 3 x H x W floats per rank (12.6 MB at bs 16, 256 x 256): negligible next to the
    gradient exchange."""
    world, rank = dist.get_world_size(), dist.get_rank()
    out = model.output()
    tgt = targets_local.to(device=out.device, dtype=torch.float32).contiguous()
    allo = torch.empty((world * out.shape[0],) + tuple(out.shape[1:]), dtype=torch.float32, device=out.device)
    allt = torch.empty_like(allo)
    dist.all_gather_into_tensor(allo, out.contiguous())
    dist.all_gather_into_tensor(allt, tgt)
    model.set_style_context(allo, allt, rank * out.shape[0])


def generator_backward_with_exchange(model, label_real, xchg, net_g=0):
    """backward_G in buckets with the exchange AND the optimizer pipelined behind it:
        bwd part k ; begin all-reduce(bucket k) ; [wait bucket k-1 ; AdamW(bucket k-1)]
    so a bucket's transfer runs under the next bucket's back-propagation and its AdamW under the transfer of the one
    after.  Equivalent to backward_G + one all-reduce + optimizer_step (AdamW is elementwise on the arena)."""
    gG = model.grad_arena(net_g)
    ranges = []
    for part in range(model.backward_G_parts()):
        off, cnt = model.backward_G_part(label_real, part)
        xchg.begin(gG[off:off + cnt])
        if ranges:                                  # the previous bucket has had a whole part's time to arrive
            xchg.wait_oldest()
            o, c = ranges[-1]
            model.optimizer_step_range(net_g, o, c, first=len(ranges) == 1)
        ranges.append((off, cnt))
    xchg.wait_oldest()
    o, c = ranges[-1]
    model.optimizer_step_range(net_g, o, c, first=len(ranges) == 1)
    xchg.finish()


def broadcast_arena(flat, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)


# ---- library-owned exchange (swn_ctx_attach_comm / swn_model_step_dp) -----------------------------------------------------------
class _NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]        # rccl.h: NCCL_UNIQUE_ID_BYTES


def _loaded_rccl():
    """The librccl this process ALREADY holds (torch's): dlopen by the path of the mapped file returns the same handle, so the
    communicator created here and the ncclAllReduce handed to the library belong to the one RCCL instance in the process."""
    paths = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "librccl" in line or "libnccl" in line:
                    paths.append(line.split()[-1])
    except OSError:
        pass
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    for cand in paths + [os.path.join(libdir, "librccl.so"), "/opt/rocm/lib/librccl.so", "librccl.so"]:
        try:
            return C.CDLL(cand)
        except OSError:
            continue
    raise RuntimeError("librccl not found (torch.distributed's nccl backend loads it)")


class NativeComm:
    """A communicator for the library's own gradient exchange (allreduce_attach(rccl_comm)).

    backend "rccl": ncclGetUniqueId on rank 0, broadcast through the default process group, ncclCommInitRank on every rank --
    through ctypes on the RCCL the process already holds -- and the ADDRESS of that library's ncclAllReduce handed to
    swn_ctx_attach_comm together with the communicator: from then on the library enqueues the collectives itself, on a stream of
    its own, ordered against its compute streams by its own events (swn_model_step_dp).  No tensor of the host framework and none
    of its streams take part in the exchange.

    backend "gloo" (CPU tests, host simulator): the same C-ABI hook driven by a Python callback that all-reduces the host buffer
    through torch.distributed -- it exercises the library-side sequencing (bucket ranges, ranged AdamW, step counters), not
    stream ordering."""

    def __init__(self, ctx, backend=None):
        self.ctx = ctx
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.backend = backend or ("rccl" if ctx.lib.is_device else "gloo")
        self._keep = []
        if self.backend == "rccl":
            self._init_rccl()
        else:
            self._init_callback()

    def _agree(self, ok, what):
        """Every rank enters or skips each collective of the bring-up TOGETHER: all-reduce(MIN) of `ok` over the default process group
        before the next stage; if any rank failed `what`, all ranks raise here (the caller, open_native_comm, then agrees on the torch
        form) instead of one rank leaving while its peers sit in the uid broadcast or in ncclCommInitRank."""
        if self.world > 1:
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if ok and not int(t.item()):
                raise RuntimeError("library-owned exchange: %s failed on another rank" % what)

    def _init_rccl(self):
        uid, err = _NcclUniqueId(), None
        try:                                    # stage 1 (local): the library, its entry points, rank 0's unique id
            if self.rank == 0 and os.environ.get("SWAPNET_TEST_RCCL_FAIL") == "rank0-before-broadcast":
                raise RuntimeError("injected: rank 0 fails before the uid broadcast (tests/test_data_parallel.py)")
            dll = self.dll = _loaded_rccl()
            dll.ncclGetUniqueId.argtypes = [C.POINTER(_NcclUniqueId)]
            dll.ncclGetUniqueId.restype = C.c_int
            dll.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
            dll.ncclCommInitRank.restype = C.c_int
            dll.ncclCommDestroy.argtypes = [C.c_void_p]
            dll.ncclCommDestroy.restype = C.c_int
            if self.rank == 0:
                rc = dll.ncclGetUniqueId(C.byref(uid))
                if rc != 0:
                    raise RuntimeError("ncclGetUniqueId failed: %d" % rc)
            if self.rank != 0 and os.environ.get("SWAPNET_TEST_RCCL_FAIL") == "rank1-before-init":
                raise RuntimeError("injected: a non-zero rank fails before ncclCommInitRank")
        except Exception as e:                                    # noqa: BLE001 -- agreed on below, then re-raised
            err = e
        self._agree(err is None, "loading RCCL / ncclGetUniqueId")
        if err is not None:
            raise err
        if self.world > 1:                      # stage 2: every rank is here, so every rank takes part in the broadcast
            box = [C.string_at(C.addressof(uid), 128)] if self.rank == 0 else [None]      # (raw memory: c_char arrays truncate at NUL)
            dist.broadcast_object_list(box, src=0)
            C.memmove(C.addressof(uid), box[0], 128)
        if os.environ.get("SWAPNET_TEST_RCCL_FAIL"):               # (the injection tests stop short of the device call)
            raise RuntimeError("injected: bring-up stopped before ncclCommInitRank")
        torch.cuda.set_device(self.ctx.device)
        comm = C.c_void_p()                     # stage 3: entered by all ranks together; its outcome is agreed on by open_native_comm
        rc = dll.ncclCommInitRank(C.byref(comm), self.world, uid, self.rank)
        if rc != 0:
            raise RuntimeError("ncclCommInitRank failed: %d" % rc)
        self.comm = comm
        fn = C.cast(dll.ncclAllReduce, C.c_void_p).value
        self.ctx.attach_comm(fn, comm, self.world)
        self.ctx.native_comm = self            # owner: Context.close() closes the communicator before the context goes

    def _init_callback(self):
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
        self.calls = []

        def allreduce(send, recv, count, dtype, op, comm, stream):
            try:
                if dtype != 7 or op != 0:
                    return 100
                if send != recv:
                    C.memmove(recv, send, count * 4)
                t = torch.frombuffer((C.c_float * count).from_address(recv), dtype=torch.float32)
                self.calls.append(int(count))
                if dist.is_initialized() and dist.get_world_size() > 1:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                return 0
            except Exception:                                  # never unwind through the C caller
                return 101
        cb = proto(allreduce)
        self._keep.append(cb)
        self.comm = None
        self.ctx.attach_comm(C.cast(cb, C.c_void_p).value, None, self.world)
        self.ctx.native_comm = self

    def close(self):
        """Detach (the library joins AND drains its exchange stream: swn_ctx_attach_comm(NULL)), then destroy the communicator.
        Call before torch.distributed.destroy_process_group()."""
        if getattr(self.ctx, "native_comm", None) is self:
            self.ctx.native_comm = None
        if getattr(self.ctx, "handle", None):
            self.ctx.attach_comm(None, None, 1)
            self.ctx.sync()
        if self.backend == "rccl" and self.comm:
            self.dll.ncclCommDestroy(self.comm)
            self.comm = None


def native_comm_requested(ctx=None, world=None):
    """Which exchange a data-parallel step uses.  SWAPNET_NATIVE_COMM=1 selects the LIBRARY-OWNED one (swn_model_step_dp: RCCL's
    all-reduce on a stream and with events the library owns; the gloo callback on the host simulator), =0 the torch.distributed
    calls of GradExchange.  Unset: the library-owned form only at world size 1 on a device build -- where it HAS run on the MI355X
    (bit-equal D arena, 25.5 vs 25.4 ms/step, profiles/native_ab_r04.txt) -- and the torch form for world > 1: no box with more than
    one GPU was ever available to this build, so the library-owned exchange has never run against a real peer (stream / event ordering
    under a live ring is untested), and an unmeasured path is not a default.  Opt in with SWAPNET_NATIVE_COMM=1; the bench line's
    `exchange` says which one ran."""
    v = os.environ.get("SWAPNET_NATIVE_COMM")
    if v is not None:
        return v == "1"
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    on_device = bool(ctx.lib.is_device) if ctx is not None else torch.cuda.is_available()
    return on_device and world == 1


def open_native_comm(ctx):
    """NativeComm(ctx) agreed on by every rank: if the communicator cannot be brought up on ANY rank, all ranks close theirs and the
    caller uses GradExchange (both are RCCL exchanges of the same buffers; the choice is reported, never silent).  Returns the
    communicator or None."""
    comm, err = getattr(ctx, "native_comm", None), None
    if comm is not None:             # one communicator per context: a second model on the same context (warp + texture in one process)
        return comm                  # shares it instead of re-attaching over it and leaking the first ncclComm
    try:
        comm = NativeComm(ctx)
    except Exception as e:                                    # noqa: BLE001
        err = e
    ok = 1 if comm is not None else 0
    if dist.is_initialized() and dist.get_world_size() > 1:
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = int(t.item())
    if not ok:
        import sys
        print("swapnet_amd.parallel: library-owned exchange NOT available (%r on this rank) -- using torch.distributed all-reduce "
              "per bucket" % (err,), file=sys.stderr, flush=True)
        if comm is not None:
            comm.close()
        return None
    return comm
