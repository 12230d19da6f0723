"""BaseGAN of the reference (/root/reference/models/base_gan.py:14-231): GAN flags, the G / D /
GANLoss / optimizer wiring and `optimize_parameters` -- here one fused native step."""
from abc import ABC, abstractmethod
from argparse import ArgumentParser

import os

import torch

from .. import engine, modules, optimizers, parallel
from ..modules import discriminators
from ..modules.loss import GANLoss
from ..modules.native import NativeBackend
from .base_model import BaseModel


class BaseGAN(BaseModel, ABC):
    KIND = None

    @staticmethod
    def modify_commandline_options(parser: ArgumentParser, is_train):
        """Same flags, defaults and aliases as base_gan.py:16-128."""
        if is_train:
            parser.add_argument("--gan_mode", help="gan regularization to use", default="vanilla",
                                choices=("vanilla", "wgan", "wgan-gp", "lsgan", "dragan-gp", "dragan-lp",
                                         "mescheder-r1-gp", "mescheder-r2-gp"))
            parser.add_argument("--lambda_gan", type=float, default=1.0, help="weight for adversarial loss")
            parser.add_argument("--lambda_discriminator", type=float, default=1.0, help="weight for discriminator loss")
            parser.add_argument("--lambda_gp", help="weight parameter for gradient penalty", type=float, default=10)
            parser.add_argument("--discriminator", default="basic", choices=("basic", "pixel", "n_layers"),
                                help="what discriminator type to use")
            parser.add_argument("--n_layers_D", type=int, default=3, help="only used if discriminator==n_layers")
            parser.add_argument("--norm", type=str, default="instance",
                                help="instance normalization or batch normalization [instance | batch | none]")
            parser.add_argument("--optimizer_G", "--opt_G", "--optim_G", help="optimizer for generator",
                                default="AdamW", choices=("AdamW", "AdaBound"))
            parser.add_argument("--lr", "--g_lr", "--learning_rate", type=float, default=0.0001,
                                help="initial learning rate for generator")
            parser.add_argument("--beta1", type=float, default=0.5, help="momentum term of adam")
            parser.add_argument("--optimizer_D", "--opt_D", "--optim_D", help="optimizer for discriminator",
                                default="AdamW", choices=("AdamW", "AdaBound"))
            parser.add_argument("--d_lr", type=float, default=0.0004, help="initial learning rate for Discriminator")
            parser.add_argument("--d_wt_decay", "--d_weight_decay", dest="d_weight_decay", default=0.01, type=float,
                                help="optimizer L2 weight decay")
            parser.add_argument("--gan_label_mode", default="smooth", choices=("hard", "smooth"),
                                help="whether to use hard (real 1.0 and fake 0.0) or smooth "
                                     "(real [0.7, 1.1] and fake [0., 0.3]) values for labels")
        return parser

    def __init__(self, opt):
        super().__init__(opt)
        self.backend = NativeBackend(self.KIND, is_train=self.is_train, dropout=0.5,
                                     num_roi=getattr(opt, "body_channels", 12), device=self.gpu_id,
                                     lib=getattr(opt, "_swapnet_lib", None),
                                     default_shape=(getattr(opt, "batch_size", 1), getattr(opt, "crop_size", 128),
                                                    getattr(opt, "crop_size", 128)),
                                     # warp: the representation flags resolved by WarpModel.__init__ (warp_model.py:49-55);
                                     # texture: cloths are always label one-hots of --cloth_channels (texture_model.py:105)
                                     body_channels=getattr(self, "body_channels", 3),
                                     cloth_channels=getattr(self, "cloth_channels", getattr(opt, "cloth_channels", 19)))
        if self.is_train and getattr(opt, "discriminator", "basic") == "n_layers":
            # define_D(..., opt.n_layers_D) (base_gan.py:147): the native networks of a stage are built together, on first use
            self.backend.n_layers_D = int(opt.n_layers_D)
        elif self.is_train and getattr(opt, "discriminator", "basic") == "pixel":
            self.backend.n_layers_D = 0             # --discriminator pixel (base_gan.py:61-65): the 1x1 PixelDiscriminator
        self.net_generator = self.define_G()
        modules.init_weights(self.net_generator, opt.init_type, opt.init_gain)      # base_gan.py:141
        self.model_names = ["generator"]
        self._step = 0
        self.world = 1
        self._xchg = None
        if self.is_train:
            self.net_discriminator = discriminators.define_D(
                self.get_D_inchannels(), 64, opt.discriminator, opt.n_layers_D, opt.norm, backend=self.backend)
            modules.init_weights(self.net_discriminator, opt.init_type, opt.init_gain)
            self.model_names.append("discriminator")
            use_smooth = opt.gan_label_mode == "smooth"
            self.criterion_GAN = GANLoss(opt.gan_mode, smooth_labels=use_smooth)
            self.criterion_GAN.ctx = self.backend.ctx
            if opt.lambda_discriminator:
                self.loss_names = ["D", "D_real", "D_fake"]
                if self.criterion_GAN.gp_mode:                 # base_gan.py:163-164
                    self.loss_names += ["D_gp"]
            self.loss_names += ["G"]
            if opt.lambda_gan:
                self.loss_names += ["G_gan"]
            self.optimizer_G = optimizers.define_optimizer(self.net_generator, opt, "G")
            self.optimizer_D = optimizers.define_optimizer(self.net_discriminator, opt, "D")
            self.optimizer_names = ("G", "D")
            if self.criterion_GAN.gp_mode and self.KIND != "warp":
                raise NotImplementedError("gan mode %s: the gradient-penalty objectives are implemented for the warp "
                                          "stage (the reference's texture-stage call passes unconditioned tensors to "
                                          "the conditional discriminator and fails)" % opt.gan_mode)
            if self.criterion_GAN.gp_mode and opt.discriminator == "pixel":
                raise NotImplementedError("gan mode %s with --discriminator pixel: the native gradient penalty walks the "
                                          "NLayerDiscriminator (basic / n_layers)" % opt.gan_mode)
            self.backend.set_hyper(gan_mode=self.criterion_GAN.native_mode, lambda_gan=opt.lambda_gan,
                                   gp_mode=self.criterion_GAN.gp_mode, lambda_gp=getattr(opt, "lambda_gp", 10.0))
            for n in ("D", "D_real", "D_fake", "G", "G_gan", "G_ce", "G_l1", "G_content", "G_style", "D_gp"):
                setattr(self, "loss_" + n, 0.0)
            if parallel.launched_data_parallel():      # the unchanged train.py under torchrun (parallel.py)
                rank, _ = parallel.init_from_env()
                self.enable_data_parallel()
                # every rank walks the dataset in its own order (the DataLoader's sampler seeds itself from this RNG)
                torch.manual_seed(torch.initial_seed() + 7919 * rank)

    # ---- data parallel (new design; the reference is single-device: SURVEY.md 2a) ---------
    def enable_data_parallel(self):
        """One process per GPU: average the two flat gradient arenas over RCCL between each
        backward and its optimizer step.  Call after torch.distributed is initialised."""
        import torch.distributed as dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self._xchg = parallel.GradExchange(self.world)
        self.backend.set_hyper(grad_scale=1.0 / self.world)
        for net in (engine.NET_G, engine.NET_D):
            parallel.broadcast_arena(self.backend.cur.weight_arena(net) if self.backend.cur else
                                     self.backend.any_model().weight_arena(net))
        return self

    @abstractmethod
    def get_D_inchannels(self):
        pass

    @abstractmethod
    def define_G(self):
        pass

    def _native(self):
        return self.backend.cur

    @staticmethod
    def _set_cloth(m, slot, t):
        """Cloth segmentations may arrive one-hot (B,19,H,W) float -- the reference dataloader's format --
        or as the integer label map (B,H,W) of the on-disk .npz: the latter is
        expanded on the device (1/19 of the host-to-device bytes)."""
        if t.dim() == 3 and not t.is_floating_point():
            m.set_input_labels(slot, t)
        else:
            m.set_input(slot, t)

    def _ce_only(self):
        return getattr(self.opt, "warp_mode", "gan") == "ce" and self.KIND == "warp"

    def _draw_labels(self):
        """The three smooth-label scalars in the reference's draw order: backward_D fake, real
        (warp_model.py:116,120), backward_G real (:158).  --warp_mode ce never calls GANLoss
        (warp_model.py:169-183): no draws, so a seeded run consumes the RNG like the reference."""
        if self._ce_only():
            return [0.0, 0.0, 0.0]
        c = self.criterion_GAN
        labels = [c.sample_label(False), c.sample_label(True)]
        if c.gp_mode and getattr(self.opt, "gp_host_random", False):
            # gradient_penalty's draws sit between backward_D's labels and backward_G's (loss.py:141-147): drawn here
            # from the global torch RNG in the reference's order so that a seeded run reproduces the reference's CPU path;
            # by default the library draws them on the device (no host round trip of a (B,22,H,W) tensor)
            m = self._native()
            beta = torch.rand(m.B, self.get_D_inchannels(), m.H, m.W) if c.gp_mode >= 2 else None
            alpha = torch.rand([m.B, 1, 1, 1])
            m.set_gp_random(alpha, beta)
        labels.append(c.sample_label(True))
        return labels

    def optimize_parameters(self):
        """base_gan.py:194-203: forward, D step, G step."""
        m = self._native()
        self._step += 1
        training = bool(self.net_generator.training)
        seed = (self._step * 1000003 + torch.initial_seed()) % (2 ** 62)
        labels = self._draw_labels()
        self.optimizer_G.sync(); self.optimizer_D.sync()          # live param_groups (lr schedulers)
        if self.world == 1:
            # SWAPNET_CAPTURED_STEP=1: the step as a recorded hipGraph (swn_model_step_captured; bit-identical results)
            m.step(labels, training=training, seed=seed, captured=os.environ.get("SWAPNET_CAPTURED_STEP") == "1")
        elif self._native_exchange():
            # the library drives the exchange itself (swn_model_step_dp: RCCL's all-reduce on a stream and with events the library
            # owns, AdamW of each bucket behind its all-reduce); SWAPNET_NATIVE_COMM=0: the torch.distributed calls below
            rank = torch.distributed.get_rank()
            labels = parallel.broadcast_floats(labels)
            if self.KIND == "texture" and getattr(self.opt, "lambda_style", 0) != 0:
                m.forward(training, seed + rank)
                parallel.gather_style_context(m, self.targets)      # the style Gram spans the global batch
                m.step_dp(labels, training=training, seed=seed + rank, after_forward=True)
            else:
                m.step_dp(labels, training=training, seed=seed + rank)
        else:
            rank = torch.distributed.get_rank()
            labels = parallel.broadcast_floats(labels)          # rank 0's smooth-label draws on every rank
            m.forward(training, seed + rank)
            if not self._ce_only():                             # --warp_mode ce: generator only (warp_model.py:178-183)
                m.backward_D(labels[0], labels[1])
                self._xchg.allreduce_mean(m.grad_arena(engine.NET_D))
                m.optimizer_step(engine.NET_D)
            if self.KIND == "texture" and getattr(self.opt, "lambda_style", 0) != 0:
                parallel.gather_style_context(m, self.targets)      # the style Gram spans the global batch
            # buckets: exchange of bucket k under the back-propagation of bucket k+1, its AdamW under the next exchange
            parallel.generator_backward_with_exchange(m, labels[2], self._xchg)
        self._losses_stale = True
        self._fakes = None

    def _native_exchange(self):
        """Library-owned exchange up?  Decided once, by all ranks together (parallel.open_native_comm); the communicator belongs to
        the context and is closed with it."""
        if not hasattr(self, "_native_comm"):
            self._native_comm = parallel.open_native_comm(self.backend.ctx) if parallel.native_comm_requested(self.backend.ctx) else None
        return self._native_comm is not None

    # individual phases keep working too (the reference exposes them as methods)
    def backward_D(self):
        c = self.criterion_GAN
        self._native().backward_D(c.sample_label(False), c.sample_label(True))
        self._losses_stale = True

    def backward_G(self):
        self._native().backward_G(self.criterion_GAN.sample_label(True))
        self._losses_stale = True

    def _fetch_losses(self):
        if getattr(self, "_losses_stale", False):
            for k, v in self._native().losses().items():
                setattr(self, "loss_" + k, v)
            self._losses_stale = False

    @property
    def fakes(self):
        """self.fakes of the reference: (B,C,H,W) tensor, materialised from the NHWC buffer on
        demand (display ticks, inference) instead of every step."""
        if getattr(self, "_fakes", None) is None:
            self._fakes = self._native().output()
        return self._fakes

    @fakes.setter
    def fakes(self, v):
        self._fakes = v
