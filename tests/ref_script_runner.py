"""TEST INFRASTRUCTURE.  Runs one of the reference's own, UNCHANGED entry scripts (/root/reference/train.py or
inference.py) in this process with
  * the third-party modules this image lacks (torchvision, visdom, dominate, seaborn, adabound) replaced by the
    stand-ins of oracle/ref_stubs.py (ToTensor / Normalize functional, so the reference's dataloader really loads),
  * `models`, `modules`, `optimizers` resolved to swapnet_amd (swapnet_amd.install_as_reference_packages()),
  * the CI host simulator as the back-end library (there is no GPU in the build container; on a GPU box the product
    library is used when SWAPNET_RUNNER_DEVICE=gpu).
usage: python tests/ref_script_runner.py <train.py|inference.py> <workdir> [script args...]"""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    script, workdir, args = sys.argv[1], sys.argv[2], sys.argv[3:]
    from oracle import ref_stubs
    ref_stubs.install(functional_transforms=True)
    if os.environ.get("SWAPNET_RUNNER_DEVICE", "sim") != "gpu":
        from swapnet_amd import _C
        from tests import backends
        _C._default = _C.Lib(backends.build_hostsim())
    import swapnet_amd
    swapnet_amd.install_as_reference_packages()
    os.chdir(workdir)
    sys.argv = [script] + args
    runpy.run_path(os.path.join(ref_stubs.REFERENCE_ROOT, script), run_name="__main__")


if __name__ == "__main__":
    main()
