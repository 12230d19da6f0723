"""Two-stage inference on the device.

The reference's inference.py runs the warp stage, writes `fakes[0].argmax(dim=0)` as a sparse
.npz per image (inference.py:140-149, datasets/data_utils.py:311-327), then a second pass
loads those files as the cloth input of the texture stage (inference.py:169-180,
data_utils.py:298-343).  Here the hand-off stays in HBM: warp forward -> argmax labels
(`swn_op_argmax_labels`) -> device-side one-hot expansion straight into the texture model's
input buffers (`swn_model_set_input_labels`) -> texture forward.  The label map is exactly
what the .npz would contain, so the result equals the reference's two-pass pipeline.
"""
import torch

from . import engine
from .modules.native import NativeBackend
from .util.decode_labels import argmax_labels


class TwoStagePipeline:
    def __init__(self, warp_state_dict, texture_state_dict, img_size=128, num_roi=12, ctx=None, lib=None):
        self.ctx = ctx or engine.default_context(lib=lib)
        self.warp = NativeBackend("warp", is_train=False, ctx=self.ctx)
        self.texture = NativeBackend("texture", is_train=False, num_roi=num_roi, ctx=self.ctx,
                                     default_shape=(1, img_size, img_size))
        self.warp.load_state_dict(engine.NET_G, warp_state_dict)
        self.texture.load_state_dict(engine.NET_G, texture_state_dict)

    @torch.no_grad()
    def __call__(self, bodys, input_cloths, textures, rois, return_labels=False):
        """bodys (B,3,H,W), input_cloths (B,19,H,W) one-hot or (B,H,W) int labels, textures
        (B,3,H,W), rois (B,R,4) -> generated textures (B,3,H,W) [, warped cloth labels (B,H,W)]."""
        B, _, H, W = bodys.shape
        w = self.warp.ensure(B, H, W)
        w.set_input(0, bodys)
        if input_cloths.dim() == 3 and not input_cloths.is_floating_point():
            w.set_input_labels(1, input_cloths)
        else:
            w.set_input(1, input_cloths)
        w.forward(training=False)
        labels = argmax_labels(w.output(), ctx=self.ctx)          # == compress_and_save_cloth's max_only
        t = self.texture.ensure(B, H, W)
        t.set_input(0, textures)
        t.set_input(1, rois)
        t.set_input_labels(2, labels)                             # == to_onehot_tensor(load_npz(...))
        t.forward(training=False)
        out = t.output()
        return (out, labels) if return_labels else out
