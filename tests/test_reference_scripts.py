"""SURVEY.md 8(b) "Callers": the reference's UNCHANGED train.py and inference.py, executed end to end against
swapnet_amd (models / modules / optimizers swapped in by install_as_reference_packages()) on a tiny synthetic
on-disk dataset in the reference's own formats: cloth/*.npz (scipy CSC label maps, datasets/data_utils.py:298-343),
body/*.png, normalization_stats.json (datasets/data_utils.py:30-38).  train.py runs 3 optimisation steps
(`--max_dataset_size 6 --batch_size 2`), writes args.json and the latest_* checkpoints; inference.py then rebuilds
the warp model from that checkpoint and writes the warped cloth .npz files (the hand-off format of the texture stage).
Skipped where /root/reference is not mounted (the GPU box)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.skipif(not os.path.isfile("/root/reference/train.py"), reason="reference tree not mounted"),
              pytest.mark.small_channel_winograd]     # 64x64 datasets: tests/conftest.py


def _make_dataset(root, n=6, size=64):
    from PIL import Image
    from scipy import sparse
    rs = np.random.RandomState(0)
    for sub in ("cloth", "body"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    for i in range(n):
        lab = rs.randint(0, 19, size=(size // 8, size // 8)).repeat(8, 0).repeat(8, 1)
        sparse.save_npz(os.path.join(root, "cloth", "s%03d.npz" % i), sparse.csc_matrix(lab))
        Image.fromarray(rs.randint(0, 255, size=(size, size, 3), dtype=np.uint8)).save(os.path.join(root, "body", "s%03d.png" % i))
    with open(os.path.join(root, "normalization_stats.json"), "w") as f:
        for key in ("body", "texture"):
            f.write(json.dumps({"path": key, "means": [0.5, 0.5, 0.5], "stds": [0.25, 0.25, 0.25]}) + "\n")


def _run(script, workdir, args):
    env = dict(os.environ, PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "ref_script_runner.py"), script, workdir] + args,
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (script, r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout + r.stderr


def test_reference_train_and_inference_scripts_run_unchanged(tmp_path):
    import torch
    data, ckpt, res = str(tmp_path / "data"), str(tmp_path / "ckpt"), str(tmp_path / "results")
    _make_dataset(data)
    out = _run("train.py", str(tmp_path), [
        "--name", "warp", "--model", "warp", "--dataroot", data, "--checkpoints_dir", ckpt, "--batch_size", "2",
        "--load_size", "64", "--crop_size", "64", "--max_dataset_size", "6", "--n_epochs", "1", "--num_workers", "0",
        "--input_transforms", "none", "--display_id", "0", "--no_html", "--no_confirm", "--print_freq", "2",
        "--checkpoint_freq", "1"])
    assert "The number of training images = 6" in out
    run_dir = os.path.join(ckpt, "warp")
    args = json.load(open(os.path.join(run_dir, "args.json")))
    assert args["model"] == "warp" and args["lambda_ce"] == 100 and args["optimizer_G"] == "AdamW"
    for f in ("latest_net_generator.pth", "latest_net_discriminator.pth", "latest_optim_G.pth", "latest_optim_D.pth",
              "1_net_generator.pth"):
        assert os.path.isfile(os.path.join(run_dir, f)), f
    sd = torch.load(os.path.join(run_dir, "latest_net_generator.pth"))
    assert sd["upsample_and_pad.2.weight"].shape == (19, 192, 4, 4) and len(sd) == 33
    osd = torch.load(os.path.join(run_dir, "latest_optim_G.pth"))
    assert float(osd["state"][0]["step"]) == 3.0                      # 6 images / batch 2 = 3 optimisation steps
    log = open(os.path.join(run_dir, "loss_log.txt")).read()
    assert "G_ce" in log and "D_fake" in log                           # Visualizer.print_current_losses got our loss dict
    # ---- inference.py, warp stage, from the checkpoint train.py just wrote
    out = _run("inference.py", str(tmp_path), [
        "--warp_checkpoint", os.path.join(run_dir, "latest_net_generator.pth"), "--dataroot", data, "--results_dir", res,
        "--max_dataset_size", "3", "--num_workers", "0", "--no_confirm", "--skip_intermediates", "--checkpoints_dir", ckpt])
    assert "Warp results stored in" in out
    from scipy.sparse import load_npz
    files = sorted(f for f in os.listdir(os.path.join(res, "warp")) if f.endswith(".npz"))
    assert len(files) == 3, files
    lab = load_npz(os.path.join(res, "warp", files[0])).toarray()
    assert lab.shape == (64, 64) and lab.min() >= 0 and lab.max() < 19


def test_reference_train_script_runs_data_parallel_under_a_torchrun_environment(tmp_path):
    """VERDICT r2 #6a: the unchanged train.py started once per rank with torchrun's environment (WORLD_SIZE = 2) trains
    data-parallel without any change to the script: swapnet_amd.parallel.launched_data_parallel() makes every model
    initialise the process group, take rank 0's weights, shuffle its own way through the dataset and average the
    gradient arenas (gloo here, RCCL on a node).  The two replicas see DIFFERENT batches, so bit-equal weights after three
    steps prove the exchange; a single-process run on the same data ends elsewhere."""
    import socket
    import torch
    data, ckpt = str(tmp_path / "data"), str(tmp_path / "ckpt")
    _make_dataset(data)
    args = ["--name", "warp", "--model", "warp", "--dataroot", data, "--checkpoints_dir", ckpt, "--batch_size", "2",
            "--load_size", "64", "--crop_size", "64", "--max_dataset_size", "6", "--n_epochs", "1", "--num_workers", "0",
            "--input_transforms", "none", "--display_id", "0", "--no_html", "--no_confirm", "--print_freq", "2",
            "--checkpoint_freq", "1"]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, PYTHONPATH=REPO, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SWAPNET_DIST_BACKEND="gloo", SWAPNET_FORCE_DEVICE="0", OMP_NUM_THREADS="4",
                   SWAPNET_SAVE_ALL_RANKS="1")          # ranks > 0 write no checkpoint by default (rank 0's is the job's)
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "ref_script_runner.py"), "train.py", str(tmp_path)] + args,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env))
    outs = [p.communicate(timeout=1500)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    r0 = torch.load(os.path.join(ckpt, "warp", "latest_net_generator.pth"))
    r1 = torch.load(os.path.join(ckpt, "warp", "rank1", "latest_net_generator.pth"))
    assert set(r0) == set(r1) and len(r0) == 33
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k                     # replicas stayed in lock-step
    d0 = torch.load(os.path.join(ckpt, "warp", "latest_net_discriminator.pth"))
    d1 = torch.load(os.path.join(ckpt, "warp", "rank1", "latest_net_discriminator.pth"))
    assert all(torch.equal(d0[k], d1[k]) for k in d0)
    assert float(torch.load(os.path.join(ckpt, "warp", "rank1", "latest_optim_G.pth"))["state"][0]["step"]) == 3.0


def _make_texture_dataset(root, n=6, size=64):
    """dataroot/texture/*.png + the matching cloth/*.npz + rois.csv (12 rows of xmin, ymin, xmax, ymax per image id,
    datasets/texture_dataset.py:74-78,116-119) in the reference's formats."""
    from PIL import Image
    _make_dataset(root, n, size)
    rs = np.random.RandomState(1)
    os.makedirs(os.path.join(root, "texture"), exist_ok=True)
    rows = ["id,xmin,ymin,xmax,ymax"]
    for i in range(n):
        Image.fromarray(rs.randint(0, 255, size=(size, size, 3), dtype=np.uint8)).save(os.path.join(root, "texture", "s%03d.png" % i))
        for r in range(12):
            x1, y1 = rs.randint(0, size - 2, size=2)
            w, h = rs.randint(0, size // 2, size=2)
            if r == 5:
                w = h = 0                                   # a degenerate box per image, as in real rois.csv files
            rows.append("s%03d,%d,%d,%d,%d" % (i, x1, y1, min(x1 + w, size - 1), min(y1 + h, size - 1)))
    with open(os.path.join(root, "rois.csv"), "w") as f:
        f.write("\n".join(rows) + "\n")


def test_reference_scripts_run_the_texture_stage_unchanged(tmp_path):
    """`train.py --model texture` on a texture/ + cloth/ + rois.csv dataset (the reference's TextureDataset,
    ROI flips and all) and `inference.py --texture_checkpoint`, both unchanged, against this package."""
    import torch
    data, ckpt, res = str(tmp_path / "data"), str(tmp_path / "ckpt"), str(tmp_path / "results")
    _make_texture_dataset(data)
    out = _run("train.py", str(tmp_path), [
        "--name", "texture", "--model", "texture", "--dataroot", data, "--checkpoints_dir", ckpt, "--batch_size", "2",
        "--load_size", "64", "--crop_size", "64", "--max_dataset_size", "6", "--n_epochs", "1", "--num_workers", "0",
        "--display_id", "0", "--no_html", "--no_confirm", "--print_freq", "2", "--checkpoint_freq", "1"])
    assert "The number of training images = 6" in out
    run_dir = os.path.join(ckpt, "texture")
    args = json.load(open(os.path.join(run_dir, "args.json")))
    assert args["model"] == "texture"
    sd = torch.load(os.path.join(run_dir, "latest_net_generator.pth"))
    assert "encode.model.0.weight" in sd and sd["encode.model.0.weight"].shape == (36, 36, 4, 4)
    osd = torch.load(os.path.join(run_dir, "latest_optim_G.pth"))
    assert float(osd["state"][0]["step"]) == 3.0
    log = open(os.path.join(run_dir, "loss_log.txt")).read()
    assert "G_l1" in log and "D_fake" in log
    # ---- inference.py, texture stage alone, from that checkpoint (cloth_dir = the dataset's own segmentations)
    out = _run("inference.py", str(tmp_path), [
        "--texture_checkpoint", os.path.join(run_dir, "latest_net_generator.pth"), "--dataroot", data, "--results_dir", res,
        "--max_dataset_size", "3", "--num_workers", "0", "--no_confirm", "--checkpoints_dir", ckpt])
    assert "Textured results stored in" in out
    imgs = [f for _, _, fs in os.walk(os.path.join(res, "texture")) for f in fs if f.endswith(".png")]
    assert len(imgs) >= 3, imgs
