"""util.decode_labels of the reference (/root/reference/util/decode_labels.py:24-55): argmax
over the channel axis -> 19-colour palette, as one integer HIP kernel instead of a per-pixel
Python loop.  Bit-exact (tests/test_ops.py::test_label_decode_and_onehot_bit_exact)."""
import torch

from .. import _C, engine

n_classes = 19


def decode_cloth_labels(pt_tensor, num_images=-1, num_classes=n_classes, ctx=None):
    ctx = ctx or engine.default_context()
    n, c, h, w = pt_tensor.shape
    if num_images < 0:
        num_images = n
    assert n >= num_images, "Batch size %d should be greater or equal than number of images to save %d." % (n, num_images)
    x = pt_tensor[:num_images].detach().to(device=ctx.device, dtype=torch.float32).contiguous()
    rgb = torch.empty((num_images, 3, h, w), dtype=torch.uint8, device=ctx.device)
    ctx.lib.call("swn_op_decode_labels", ctx.handle, _C.ptr(x), num_images, c, h, w, _C.ptr(rgb))
    return rgb.cpu()


def argmax_labels(pt_tensor, ctx=None):
    """max_only of compress_and_save_cloth (datasets/data_utils.py:322): (B,C,H,W) -> int32 (B,H,W)."""
    ctx = ctx or engine.default_context()
    n, c, h, w = pt_tensor.shape
    x = pt_tensor.detach().to(device=ctx.device, dtype=torch.float32).contiguous()
    lab = torch.empty((n, h, w), dtype=torch.int32, device=ctx.device)
    ctx.lib.call("swn_op_argmax_labels", ctx.handle, _C.ptr(x), n, c, h, w, _C.ptr(lab))
    return lab


def labels_to_onehot(labels, n_labels=n_classes, ctx=None):
    """to_onehot_tensor (datasets/data_utils.py:330-343) on the device: int labels (B,H,W) ->
    float one-hot (B,n_labels,H,W) with the background (label 0) as the all-zero vector."""
    ctx = ctx or engine.default_context()
    n, h, w = labels.shape
    lab = labels.to(device=ctx.device, dtype=torch.int32).contiguous()
    out = torch.empty((n, n_labels, h, w), dtype=torch.float32, device=ctx.device)
    ctx.lib.call("swn_op_labels_to_onehot", ctx.handle, _C.ptr(lab), n, n_labels, h, w, _C.ptr(out))
    return out
