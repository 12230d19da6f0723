"""Synthetic batches shaped like the reference's data; used by bench.py,
smoke() and the examples.  Real data would come from the reference's own dataloader
(datasets/), which swapnet_amd deliberately does not replace."""
import torch


def _blocky_labels(B, H, W, n_labels, tile, g):
    lab = torch.randint(0, n_labels, (B, max(H // tile, 1), max(W // tile, 1)), generator=g)
    return lab.repeat_interleave(tile, 1).repeat_interleave(tile, 2)[:, :H, :W]


def onehot_zero_background(lab, n_labels=19):
    """datasets/data_utils.py:330-343: the scipy-sparse label matrix drops label 0, so the
    background pixel is the ALL-ZERO vector."""
    oh = torch.nn.functional.one_hot(lab.long(), n_labels).movedim(-1, -3).float().contiguous()   # NCHW-dense, like a
    oh[..., 0, :, :] = 0.0                                                                       # DataLoader batch
    return oh


def warp_batch(B, H, W, seed=1234, n_labels=19, tile=8):
    """bodys (B,3,H,W) ~ N(0,1); target/input cloth one-hot (B,19,H,W) from blocky label maps
    (input = flipped + rolled target, mimicking warp_dataset.py:99-111 augmentation)."""
    g = torch.Generator().manual_seed(seed)
    bodys = torch.randn((B, 3, H, W), generator=g)
    lab = _blocky_labels(B, H, W, n_labels, tile, g)
    targets = onehot_zero_background(lab, n_labels)
    inputs = onehot_zero_background(torch.roll(lab.flip(2), shifts=(3, -2), dims=(1, 2)), n_labels)
    return dict(bodys=bodys, input_cloths=inputs, target_cloths=targets,
                cloth_paths=[""] * B, body_paths=[""] * B)


def texture_batch(B, H, W, seed=1234, n_labels=19, num_roi=12, tile=8):
    g = torch.Generator().manual_seed(seed)
    tex = torch.randn((B, 3, H, W), generator=g).clamp_(-3, 3)
    tgt = torch.randn((B, 3, H, W), generator=g).clamp_(-3, 3)
    cloths = onehot_zero_background(_blocky_labels(B, H, W, n_labels, tile, g), n_labels)
    x1 = torch.randint(0, W - 1, (B, num_roi), generator=g)
    y1 = torch.randint(0, H - 1, (B, num_roi), generator=g)
    w = torch.randint(0, W // 2 + 1, (B, num_roi), generator=g)
    h = torch.randint(0, H // 2 + 1, (B, num_roi), generator=g)
    rois = torch.stack((x1, y1, (x1 + w).clamp_(max=W - 1), (y1 + h).clamp_(max=H - 1)), dim=-1).float()
    rois[:, 0] = torch.tensor([W - 1, 0, W - 1, 0], dtype=torch.float32)     # one degenerate box per sample
    return dict(input_textures=tex, rois=rois, cloths=cloths, target_textures=tgt,
                cloth_paths=[""] * B, texture_paths=[""] * B)


def fill_inputs(model, kind, B, H, W, seed=1234):
    """Hands a synthetic batch of the stage's shape to a native model through its set_input slots (bench.py's order)."""
    if kind == "texture":
        batch = texture_batch(B, H, W, seed=seed)
        for i, k in enumerate(("input_textures", "rois", "cloths", "target_textures")):
            model.set_input(i, batch[k])
    else:
        batch = warp_batch(B, H, W, seed=seed)
        model.set_input(0, batch["bodys"]); model.set_input(1, batch["input_cloths"]); model.set_input(2, batch["target_cloths"])
    return batch
