"""The few host-side helpers of the reference that its models call on display ticks
(/root/reference/datasets/data_utils.py:41-90, util/util.py): not on the hot path."""
import os


def unnormalize(tensor, mean, std, clamp=True, inplace=False):
    """datasets/data_utils.py:41-58 (note the reference compares `tensor.shape == 4`, which is
    never true, so the per-channel loop always runs over dim 0 -- reproduced)."""
    if not inplace:
        tensor = tensor.clone()
    for t, m, s in zip(tensor, mean, std):
        t.mul_(s).add_(m)
        if clamp:
            t.clamp_(0, 1)
    return tensor


def scale_tensor(tensor, scale_each=False, range=None):
    """datasets/data_utils.py:61-90 (torchvision.utils.make_grid's normalisation)."""
    tensor = tensor.clone()

    def norm_ip(img, lo, hi):
        img.clamp_(min=lo, max=hi)
        img.add_(-lo).div_(hi - lo + 1e-5)

    def norm_range(t):
        if range is not None:
            norm_ip(t, range[0], range[1])
        else:
            norm_ip(t, float(t.min()), float(t.max()))

    if scale_each:
        for t in tensor:
            norm_range(t)
    else:
        norm_range(tensor)
    return tensor


class PromptOnce:
    @staticmethod
    def makedirs(path, confirm=False):
        os.makedirs(path, exist_ok=True)
