"""TEST INFRASTRUCTURE: runs the repo's bench.py -- unchanged, it has no CPU path -- with the host simulator standing in for the GPU, so that
the lines of bench.py no 1-GPU box ever executes (the N > 1 launch / exchange / report path, `--stage joint`) run in CPU CI.

    python tests/bench_on_hostsim.py <bench.py arguments>        (under torch.distributed.run for N > 1, SWAPNET_DIST_BACKEND=gloo)

What is stubbed, in THIS process only: engine.Context -> a context on tests/hostsim's build of the C-ABI; torch.cuda.{is_available,
set_device, synchronize, current_device, get_device_name, Event} -> trivial answers (the simulator is synchronous); torch.tensor(device="cuda")
-> cpu.  The JSON line that comes out is a contract check (fields, rank table, whole-job rate arithmetic), never a measurement."""
import os
import runpy
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from swapnet_amd import _C, engine          # noqa: E402
from tests import backends                  # noqa: E402

_lib = _C.Lib(backends.build_hostsim())
_Context = engine.Context


def _sim_context(device=None, lib=None, workspace_mb=512, **kw):
    return _Context(lib=_lib, workspace_mb=64)


engine.Context = _sim_context
_local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.is_available = lambda: True
torch.cuda.device_count = lambda: int(os.environ.get("SWAPNET_SIM_DEVICES", "8"))
os.environ["SWAPNET_BENCH_ENTRY"] = os.path.abspath(__file__)        # bench.py's self-launch (--gpus N, no launcher) re-enters through this wrapper
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.current_device = lambda: _local
torch.cuda.get_device_name = lambda *a, **k: "host simulator (tests/bench_on_hostsim.py)"


class _Event:                                    # torch.cuda.Event of the per-bucket timing section: nothing to time on the simulator
    def __init__(self, enable_timing=False):
        pass

    def record(self, *a, **k):
        pass

    def elapsed_time(self, other):
        return 0.0


torch.cuda.Event = _Event
_tensor = torch.tensor


def _tensor_on_cpu(*a, **k):
    if str(k.get("device", "")).startswith("cuda"):
        k["device"] = "cpu"
    return _tensor(*a, **k)


torch.tensor = _tensor_on_cpu
sys.argv = [os.path.join(REPO, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
