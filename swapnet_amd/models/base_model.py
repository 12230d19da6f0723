"""BaseModel API of the reference (/root/reference/models/base_model.py:10-245): what train.py
and inference.py call on a model.  Behaviour kept: attribute names, checkpoint file names
(`{epoch}_net_{name}.pth`, `{epoch}_optim_{name}.pth`) and their state-dict key layout.

This file restates the reference's boundary class ("KEEP"): pure control plane -- getters, checkpoint
file naming, `setup` / `eval` / `test` / `print_networks` -- that the reference's unchanged train.py and inference.py call by
name: the method names and signatures are the reference's, the bodies go through two small helpers (`_attrs`, `_ckpt_file`).
Nothing here is on the hot path; it is not to grow (the step, the networks and the losses live behind the C ABI)."""
import os
from abc import ABC, abstractmethod
from collections import OrderedDict

import torch

from ..util.util import PromptOnce


class BaseModel(ABC):
    def __init__(self, opt):
        self.opt = opt
        from .. import parallel
        self._dp_rank = 0
        if parallel.launched_data_parallel():          # started under torchrun: one process per GPU (parallel.py)
            opt.gpu_id = parallel.launch_device(opt.gpu_id)
            self._dp_rank = int(os.environ.get("RANK", "0"))
        self.gpu_id = opt.gpu_id
        self.is_train = opt.is_train
        # the reference maps gpu_id None -> CPU (base_model.py:36-40); this back end is HIP-only
        if self.gpu_id is None or (isinstance(self.gpu_id, int) and self.gpu_id < 0):
            raise RuntimeError("swapnet_amd runs on an MI355X only (gpu_id=%r); the reference's CPU path "
                               "is not part of this library" % (self.gpu_id,))
        self.device = torch.device(f"cuda:{self.gpu_id}")
        # one checkpoint directory for the job: every rank LOADS from it (--continue_train / inference under torchrun find the
        # files a single-GPU run or a smaller world wrote), only rank 0 WRITES to it (save_checkpoint below)
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        if self.is_train and not self._dp_rank:
            PromptOnce.makedirs(self.save_dir, not getattr(opt, "no_confirm", True))
        self.loss_names = []
        self.model_names = []
        self.visual_names = []
        self.optimizer_names = []
        self.image_paths = []
        self.metric = 0

    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    @abstractmethod
    def set_input(self, input):
        pass

    @abstractmethod
    def forward(self):
        pass

    @abstractmethod
    def optimize_parameters(self):
        pass

    # ---- small helpers: the reference keeps its networks / optimizers / losses / visuals as attributes named by prefix + name ----
    def _attrs(self, prefix, names):
        """(name, attribute) for every string in `names` (the reference's lists may hold non-string placeholders)."""
        return [(n, getattr(self, prefix + n)) for n in names if isinstance(n, str)]

    def _ckpt_file(self, directory, epoch, kind, name):
        """`{epoch}_net_{name}.pth` / `{epoch}_optim_{name}.pth`: the reference's file names (base_model.py:156-173) -- API,
        pretrained checkpoints are distributed under them."""
        return os.path.join(directory, "%s_%s_%s.pth" % (epoch, kind, name))

    def setup(self, opt):
        """base_model.py:76-87: restore when testing / continuing, then report the networks."""
        if opt.continue_train or not self.is_train:
            self.load_checkpoint_dir(opt.load_epoch)
        self.print_networks(opt.verbose)
        return self

    def eval(self):
        for _, net in self._attrs("net_", self.model_names):
            net.eval()
        return self

    def test(self):
        with torch.no_grad():
            self.forward()
            self.compute_visuals()

    def compute_visuals(self):
        pass

    def get_image_paths(self):
        return self.image_paths

    def get_current_visuals(self):
        return OrderedDict(self._attrs("", self.visual_names))

    def get_current_losses(self):
        """One small D2H copy + sync per call (base_model.py:139-147 does float(loss) per name)."""
        self._fetch_losses()
        return OrderedDict((n, float(v)) for n, v in self._attrs("loss_", self.loss_names))

    def _fetch_losses(self):
        pass

    def save_checkpoint(self, epoch):
        target = self.save_dir
        if self._dp_rank:
            # data-parallel replicas hold identical weights and optimizer state: rank 0's files are the checkpoint.
            # SWAPNET_SAVE_ALL_RANKS=1 (diagnostics / the lock-step test) makes the others write theirs to <save_dir>/rank<r>/
            if os.environ.get("SWAPNET_SAVE_ALL_RANKS") != "1":
                return
            target = os.path.join(self.save_dir, "rank%d" % self._dp_rank)
            os.makedirs(target, exist_ok=True)
        for kind, prefix, names in (("net", "net_", self.model_names), ("optim", "optimizer_", self.optimizer_names)):
            for name, obj in self._attrs(prefix, names):
                torch.save(obj.state_dict(), self._ckpt_file(target, epoch, kind, name))          # CPU tensors, reference keys

    def load_model_weights(self, model_name, weights_file):
        print("restoring net_%s <- %s" % (model_name, weights_file))
        weights = torch.load(weights_file, map_location="cpu")
        if hasattr(weights, "_metadata"):           # (torch.save of a Module.state_dict() carries it; the native nets do not want it)
            del weights._metadata
        getattr(self, "net_" + model_name).load_state_dict(weights)
        return self

    def load_checkpoint_dir(self, epoch):
        for name, _ in self._attrs("net_", self.model_names):
            self.load_model_weights(name, self._ckpt_file(self.save_dir, epoch, "net", name))
        if self.is_train:
            for name, optimizer in self._attrs("optimizer_", self.optimizer_names):
                src = self._ckpt_file(self.save_dir, epoch, "optim", name)
                print("restoring optimizer_%s <- %s" % (name, src))
                optimizer.load_state_dict(torch.load(src, map_location="cpu"))
        return self

    def print_networks(self, verbose):
        print("---------- Networks initialized -------------")
        for name, net in self._attrs("net_", self.model_names):
            count = 0
            for shape in net.native_param_shapes().values():
                n = 1
                for d in shape:
                    n *= int(d)
                count += n
            if verbose:
                print(net)
            print("[Network %s] Total number of parameters : %.3f M" % (name, count / 1e6))
        print("-----------------------------------------------")

    def set_requires_grad(self, nets, requires_grad=False):
        """Kept for API compatibility; gradient flow is fixed by the native step (D weight
        gradients are simply not computed during backward_G: SURVEY.md quirk 5)."""
        pass
