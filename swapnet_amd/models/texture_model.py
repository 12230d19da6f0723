"""TextureModel of the reference (/root/reference/models/texture_model.py:16-180): texture stage
= TextureModule generator against a PatchGAN conditioned on cat(cloth, texture); G loss = GAN +
lambda_l1 * L1 + lambda_content * content + lambda_style * style."""
from argparse import ArgumentParser

from ..modules.losses import PerceptualLoss
from ..modules.swapnet_modules import TextureModule
from ..util.decode_labels import decode_cloth_labels, labels_to_onehot
from ..util.util import scale_tensor, unnormalize
from .base_gan import BaseGAN


class TextureModel(BaseGAN):
    KIND = "texture"

    @staticmethod
    def modify_commandline_options(parser: ArgumentParser, is_train):
        parser = super(TextureModel, TextureModel).modify_commandline_options(parser, is_train)
        if is_train:
            parser.add_argument("--netG", default="swapnet", choices=["swapnet", "unet_128"])
            parser.add_argument("--lambda_l1", type=float, default=10, help="weight for L1 loss in final term")
            parser.add_argument("--lambda_content", type=float, default=20, help="weight for content loss in final term")
            parser.add_argument("--lambda_style", type=float, default=1e-8, help="weight for content loss in final term")
            parser.add_argument("--vgg_weights", default=None, help="(swapnet_amd) file with torchvision vgg16 weights "
                                "(vgg16().state_dict() or .features.state_dict()) for the perceptual loss")
            parser.set_defaults(display_ncols=5)
        return parser

    def __init__(self, opt):
        super().__init__(opt)
        self.visual_names = ["textures_unnormalized", "cloths_decoded", "fakes", "fakes_scaled"]
        if self.is_train:
            self.visual_names.append("targets_unnormalized")
            if opt.lambda_style != 0 and getattr(opt, "batch_size", 1) * opt.texture_channels > 1024:
                raise ValueError("texture stage with the style term on (lambda_style=%g): batch_size * texture_channels "
                                 "must be <= 1024 (rows of the image Gram); got %d x %d"
                                 % (opt.lambda_style, opt.batch_size, opt.texture_channels))
            self.criterion_perceptual = PerceptualLoss(use_style=opt.lambda_style != 0, backend=self.backend)
            self._init_vgg()
            self.backend.set_hyper(lambda_l1=opt.lambda_l1, lambda_content=opt.lambda_content,
                                   lambda_style=opt.lambda_style)
            for loss in ["l1", "content", "style"]:
                if getattr(opt, "lambda_" + loss) != 0:
                    self.loss_names.append(f"G_{loss}")

    def _init_vgg(self):
        """The reference's PerceptualLoss builds torchvision's vgg16(pretrained=True) (perceptual.py:26).  Sources
        tried in order: (1) a file named by --vgg_weights / $SWAPNET_VGG16_WEIGHTS holding vgg16().state_dict() or
        vgg16().features.state_dict(); (2) torchvision's pretrained model when torchvision is importable and its
        checkpoint is cached or downloadable.  If neither is available the network starts from seeded random
        weights (private generator, global RNG untouched) and -- because lambda_content would then optimise
        against random features, a DIFFERENT objective than the reference's -- a RuntimeWarning says so loudly;
        $SWAPNET_REQUIRE_PRETRAINED_VGG=1 turns that into an error."""
        import os
        import warnings
        import torch
        crit = self.criterion_perceptual
        path = getattr(self.opt, "vgg_weights", None) or os.environ.get("SWAPNET_VGG16_WEIGHTS")
        feats = None
        if path:
            sd = torch.load(path, map_location="cpu")
            sd = sd.get("state_dict", sd)
            feats = {k[len("features."):] if k.startswith("features.") else k: v for k, v in sd.items()
                     if k.startswith("features.") or k.split(".")[0].isdigit()}
            source = path
        else:
            try:
                from torchvision.models import vgg16
                feats = vgg16(pretrained=True).features.state_dict()
                source = "torchvision vgg16(pretrained=True)"
            except Exception:            # torchvision absent, or no cached checkpoint and no network
                feats = None
        if feats is not None:
            crit.load_vgg16_features(feats)
            self.vgg_source = source
            print("PerceptualLoss: VGG16 features loaded from %s" % source)
            return
        g = torch.Generator().manual_seed(4242)
        sd = {}
        for name, shape in crit.native_param_shapes().items():
            if name.endswith(".bias"):
                sd[name] = torch.randn(shape, generator=g) * 0.05
            else:
                sd[name] = torch.randn(shape, generator=g) * (2.0 / (shape[0] * 9)) ** 0.5
        crit.load_state_dict(sd)
        self.vgg_source = "seeded-random"
        if self.opt.lambda_content != 0:
            msg = ("PerceptualLoss: pretrained VGG16 weights are NOT available (no --vgg_weights / "
                   "$SWAPNET_VGG16_WEIGHTS file, torchvision checkpoint not obtainable): the content loss "
                   "(lambda_content=%g) will be computed on SEEDED RANDOM VGG16 features, which is not the "
                   "reference's objective.  Load real weights with "
                   "model.criterion_perceptual.load_vgg16_features(vgg16(pretrained=True).features.state_dict())."
                   % self.opt.lambda_content)
            if os.environ.get("SWAPNET_REQUIRE_PRETRAINED_VGG") == "1":
                raise RuntimeError(msg)
            warnings.warn(msg, RuntimeWarning, stacklevel=2)

    def compute_visuals(self):
        self.textures_unnormalized = unnormalize(self.textures.cpu(), *self.opt.texture_norm_stats)
        # util/draw_rois (ROI rectangle overlay, display only) is out of scope
        cl = labels_to_onehot(self.cloths, self.opt.cloth_channels, ctx=self.backend.ctx) if self.cloths.dim() == 3 else self.cloths
        self.cloths_decoded = decode_cloth_labels(cl, ctx=self.backend.ctx)
        self.fakes_scaled = scale_tensor(self.fakes.cpu(), scale_each=True)
        if self.is_train:
            self.targets_unnormalized = unnormalize(self.targets.cpu(), *self.opt.texture_norm_stats)

    def get_D_inchannels(self):
        return self.opt.texture_channels + self.opt.cloth_channels

    def define_G(self):
        netG = getattr(self.opt, "netG", "swapnet")
        if netG == "swapnet":
            return TextureModule(texture_channels=self.opt.texture_channels, cloth_channels=self.opt.cloth_channels,
                                 num_roi=self.opt.body_channels, img_size=self.opt.crop_size,
                                 norm_type=getattr(self.opt, "norm", "instance"), backend=self.backend)
        if netG == "unet_128":
            raise NotImplementedError("netG unet_128 (plain pix2pix baseline) is not implemented natively")
        raise ValueError("Cannot find implementation for " + netG)

    def set_input(self, input):
        """texture_model.py:113-119."""
        self.textures, self.rois, self.cloths = input["input_textures"], input["rois"], input["cloths"]
        B, _, H, W = self.textures.shape
        m = self.backend.ensure(B, H, W)
        m.set_input(0, self.textures)
        m.set_input(1, self.rois)
        self._set_cloth(m, 2, self.cloths)
        if self.is_train:
            self.targets = input["target_textures"]
            m.set_input(3, self.targets)
        self.image_paths = tuple(zip(input["cloth_paths"], input["texture_paths"]))
        self._fakes = None

    def forward(self):
        training = bool(self.net_generator.training)
        self._native().forward(training, seed=self._step if training else 0)
        self._fakes = None
