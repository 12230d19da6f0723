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
            parser.set_defaults(display_ncols=5)
        return parser

    def __init__(self, opt):
        super().__init__(opt)
        self.visual_names = ["textures_unnormalized", "cloths_decoded", "fakes", "fakes_scaled"]
        if self.is_train:
            self.visual_names.append("targets_unnormalized")
            self.criterion_perceptual = PerceptualLoss(use_style=opt.lambda_style != 0, backend=self.backend)
            self._init_vgg()
            self.backend.set_hyper(lambda_l1=opt.lambda_l1, lambda_content=opt.lambda_content,
                                   lambda_style=opt.lambda_style)
            for loss in ["l1", "content", "style"]:
                if getattr(opt, "lambda_" + loss) != 0:
                    self.loss_names.append(f"G_{loss}")

    def _init_vgg(self):
        """The reference downloads torchvision's pretrained VGG16 (perceptual.py:26).  There is no
        network here: start from seeded random weights (private generator, global RNG untouched)
        and let the user load real ones with
        `model.criterion_perceptual.load_vgg16_features(vgg16(pretrained=True).features.state_dict())`."""
        import torch
        g = torch.Generator().manual_seed(4242)
        sd = {}
        for name, shape in self.criterion_perceptual.native_param_shapes().items():
            if name.endswith(".bias"):
                sd[name] = torch.randn(shape, generator=g) * 0.05
            else:
                sd[name] = torch.randn(shape, generator=g) * (2.0 / (shape[0] * 9)) ** 0.5
        self.criterion_perceptual.load_state_dict(sd)
        print("PerceptualLoss: VGG16 initialised with seeded random weights; load pretrained "
              "features with criterion_perceptual.load_vgg16_features(...)")

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
