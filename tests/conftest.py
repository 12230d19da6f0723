import os
import sys
import time

import pytest

_SESSION_T0 = time.monotonic()

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# Kernel routing under test.  The library picks kernels / algorithmic forms per layer from the layer's shape and the SWN_*
# switches.  The tests run under the DEFAULT environment -- the one bench.py, smoke() and any user run
# under -- unless a test opts in to the small-channel Winograd routing with @pytest.mark.small_channel_winograd: the operator-
# level cases and the 64x64 model cases have too few channels / too small maps to reach the Winograd and strided-Winograd forms
# by the product thresholds (>= 256 coarse channels on >= 16x16 maps), and SWN_WINO_MINC=32 puts them on those forms anyway
# (harder numerics, same kernels).  The full-size tests (256x256; BASELINE.json C2 / C3) never carry the marker and
# additionally assert their launch list against a scrubbed-environment run (tests/backends.py default_route).
ROUTING_SWITCHES = ("SWN_WINO_MINC", "SWN_WINOGRAD", "SWN_WINO_S2", "SWN_WINO_M", "SWN_TAIL4", "SWN_PHASE4", "SWN_HEAD_TAPN", "SWN_NARROW",
                    "SWN_DMA", "SWN_DMA_WIDE", "SWN_SPLIT", "SWN_PRECUT", "SWN_PC_PLANES", "SWN_WGRAD_PLANES", "SWN_TAIL_SPLIT", "SWN_AMAX_FUSED",
                    "SWN_PAIR", "SWN_PREFETCH", "SWN_STREAM_ADAMW", "SWN_OVERLAP", "SWN_SIM_PAIR", "SWN_SIM_SLOT_REPORT", "SWN_ROI_WAVE", "SWN_CONV_STATS", "SWN_IN_PAIR_XCD", "SWN_PHASE_ZFAST", "SWN_PREFETCH_AHEAD", "SWN_CE_EARLY", "SWN_VT_EARLY", "SWN_BIAS_MAIN")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "small_channel_winograd: run with SWN_WINO_MINC=32 (Winograd forms from 32 channels up, "
                                       "any map size) instead of the product's routing thresholds")


# Wall-clock budget of the GPU suite.  The driver runs `pytest tests/ -x -q -m gpu` in ONE process under a 1 200 s limit and a kill
# at the limit reports the whole suite as failed with no test named.  Most of the suite's time is the CPU oracle (float64 steps at
# 256 x 256), i.e. it depends on the host cores of the box.  Past the budget the remaining GPU tests are not started -- and the
# session FAILS (round-4 advice: a skip made the driver's `-x -q` run exit 0 with tests never run), naming what did not run.
# SWAPNET_GPU_SUITE_BUDGET_S=0 disables the guard.
GPU_SUITE_BUDGET_S = float(os.environ.get("SWAPNET_GPU_SUITE_BUDGET_S", "1100"))
_BUDGET_SKIPPED = []


def pytest_runtest_setup(item):
    if GPU_SUITE_BUDGET_S > 0 and item.get_closest_marker("gpu"):
        spent = time.monotonic() - _SESSION_T0
        if spent > GPU_SUITE_BUDGET_S:
            _BUDGET_SKIPPED.append(item.nodeid)
            pytest.skip("GPU suite wall budget: %.0f s spent of %.0f s (the driver kills the process at 1 200 s); "
                        "run this test on its own, or with SWAPNET_GPU_SUITE_BUDGET_S=0" % (spent, GPU_SUITE_BUDGET_S))


def pytest_terminal_summary(terminalreporter):
    if _BUDGET_SKIPPED:
        terminalreporter.write_line("GPU suite wall budget reached: %d test(s) NOT run (the session fails): %s"
                                    % (len(_BUDGET_SKIPPED), ", ".join(_BUDGET_SKIPPED)))


def pytest_sessionfinish(session, exitstatus):
    if _BUDGET_SKIPPED and session.exitstatus == 0:
        session.exitstatus = 1


@pytest.fixture(autouse=True)
def kernel_routing(request, monkeypatch):
    if os.environ.get("SWAPNET_TEST_KEEP_SWITCHES") != "1":       # (a failing test under one switch at a time)
        for k in ROUTING_SWITCHES:
            monkeypatch.delenv(k, raising=False)      # nothing leaks in from the invoking shell
    if request.node.get_closest_marker("small_channel_winograd"):
        monkeypatch.setenv("SWN_WINO_MINC", "32")
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")
