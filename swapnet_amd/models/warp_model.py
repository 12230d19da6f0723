"""WarpModel of the reference (/root/reference/models/warp_model.py:13-183): warp stage =
WarpModule generator against a PatchGAN conditioned on cat(body, cloth); G loss = lambda_ce *
CE(fakes, argmax(targets)) + lambda_gan * GAN; `--warp_mode ce` trains G alone."""
from argparse import ArgumentParser

from ..modules.swapnet_modules import WarpModule
from ..util.decode_labels import decode_cloth_labels, labels_to_onehot
from ..util.util import unnormalize
from .base_gan import BaseGAN


class WarpModel(BaseGAN):
    KIND = "warp"

    @staticmethod
    def modify_commandline_options(parser: ArgumentParser, is_train):
        if is_train:
            parser.add_argument("--warp_mode", default="gan", choices=("gan", "ce"))
            parser.add_argument("--lambda_ce", type=float, default=100, help="weight for cross entropy loss in final term")
            parser.set_defaults(display_ncols=4)
        return super(WarpModel, WarpModel).modify_commandline_options(parser, is_train)

    def __init__(self, opt):
        self.body_channels = opt.body_channels if opt.body_representation == "labels" else 3
        self.cloth_channels = opt.cloth_channels if opt.cloth_representation == "labels" else 3
        BaseGAN.__init__(self, opt)
        self.visual_names = ["inputs_decoded", "bodys_unnormalized", "fakes_decoded"]
        if self.is_train:
            self.visual_names.append("targets_decoded")
            self.backend.set_hyper(lambda_ce=opt.lambda_ce, warp_mode_ce=int(opt.warp_mode != "gan"))
            if opt.warp_mode != "gan":
                self.model_names = ["generator"]
                self.loss_names = "G"                  # sic: warp_model.py:71 assigns a string
                self.optimizer_names = ["G"]
            else:
                self.loss_names += ["G_ce"]

    def compute_visuals(self):
        as_onehot = lambda t: labels_to_onehot(t, self.cloth_channels, ctx=self.backend.ctx) if t.dim() == 3 else t
        self.inputs_decoded = decode_cloth_labels(as_onehot(self.inputs), ctx=self.backend.ctx)
        self.bodys_unnormalized = unnormalize(self.bodys.cpu(), *self.opt.body_norm_stats)
        if self.is_train:
            self.targets_decoded = decode_cloth_labels(as_onehot(self.targets), ctx=self.backend.ctx)
        self.fakes_decoded = decode_cloth_labels(self.fakes, ctx=self.backend.ctx)

    def define_G(self):
        return WarpModule(body_channels=self.body_channels, cloth_channels=self.cloth_channels, backend=self.backend)

    def get_D_inchannels(self):
        return self.cloth_channels + self.body_channels

    def set_input(self, input):
        """warp_model.py:99-104.  The three tensors go straight into the library's NHWC buffers
        (the conditioned D input is assembled there, never concatenated)."""
        self.bodys, self.inputs = input["bodys"], input["input_cloths"]
        B, _, H, W = self.bodys.shape
        m = self.backend.ensure(B, H, W)
        m.set_input(0, self.bodys)
        self._set_cloth(m, 1, self.inputs)
        if self.is_train:
            self.targets = input["target_cloths"]
            self._set_cloth(m, 2, self.targets)
        self.image_paths = tuple(zip(input["cloth_paths"], input["body_paths"]))
        self._fakes = None

    def forward(self):
        training = bool(self.net_generator.training)
        self._step += 1 if not self.is_train else 0
        self._native().forward(training, seed=self._step if training else 0)
        self._fakes = None
