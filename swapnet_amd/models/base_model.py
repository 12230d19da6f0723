"""BaseModel API of the reference (/root/reference/models/base_model.py:10-245): what train.py
and inference.py call on a model.  Behaviour kept: attribute names, checkpoint file names
(`{epoch}_net_{name}.pth`, `{epoch}_optim_{name}.pth`) and their state-dict key layout.

This file is a RESTATEMENT of the reference's boundary class ("KEEP"): pure control plane --
getters, checkpoint file naming, `setup` / `eval` / `test` / `print_networks` -- that the reference's unchanged train.py and
inference.py call by name, so most of its methods necessarily read like the reference's.  Nothing here is on the hot path;
it is not to grow (the step, the networks and the losses live behind the C ABI)."""
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

    def setup(self, opt):
        if not self.is_train or opt.continue_train:
            self.load_checkpoint_dir(opt.load_epoch)
        self.print_networks(opt.verbose)
        return self

    def eval(self):
        for name in self.model_names:
            if isinstance(name, str):
                getattr(self, "net_" + name).eval()
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
        visual_ret = OrderedDict()
        for name in self.visual_names:
            if isinstance(name, str):
                visual_ret[name] = getattr(self, name)
        return visual_ret

    def get_current_losses(self):
        """One small D2H copy + sync per call (base_model.py:139-147 does float(loss) per name)."""
        self._fetch_losses()
        errors_ret = OrderedDict()
        for name in self.loss_names:
            if isinstance(name, str):
                errors_ret[name] = float(getattr(self, "loss_" + name))
        return errors_ret

    def _fetch_losses(self):
        pass

    def save_checkpoint(self, epoch):
        save_dir = self.save_dir
        if self._dp_rank:
            # data-parallel replicas hold identical weights and optimizer state: rank 0's files are the checkpoint.
            # SWAPNET_SAVE_ALL_RANKS=1 (diagnostics / the lock-step test) makes the others write theirs to <save_dir>/rank<r>/
            if os.environ.get("SWAPNET_SAVE_ALL_RANKS") != "1":
                return
            save_dir = os.path.join(self.save_dir, f"rank{self._dp_rank}")
            os.makedirs(save_dir, exist_ok=True)
        for name in self.model_names:
            if isinstance(name, str):
                save_path = os.path.join(save_dir, f"{epoch}_net_{name}.pth")
                net = getattr(self, f"net_{name}")
                torch.save(net.state_dict(), save_path)          # CPU tensors, reference keys
        for name in self.optimizer_names:
            if isinstance(name, str):
                save_path = os.path.join(save_dir, f"{epoch}_optim_{name}.pth")
                torch.save(getattr(self, f"optimizer_{name}").state_dict(), save_path)

    def load_model_weights(self, model_name, weights_file):
        net = getattr(self, f"net_{model_name}")
        print(f"loading the model {model_name} from {weights_file}")
        state_dict = torch.load(weights_file, map_location="cpu")
        if hasattr(state_dict, "_metadata"):
            del state_dict._metadata
        net.load_state_dict(state_dict)
        return self

    def load_checkpoint_dir(self, epoch):
        for name in self.model_names:
            if isinstance(name, str):
                self.load_model_weights(name, os.path.join(self.save_dir, f"{epoch}_net_{name}.pth"))
        if self.is_train:
            for name in self.optimizer_names:
                if isinstance(name, str):
                    load_path = os.path.join(self.save_dir, f"{epoch}_optim_{name}.pth")
                    print(f"loading the optimizer {name} from {load_path}")
                    getattr(self, f"optimizer_{name}").load_state_dict(torch.load(load_path, map_location="cpu"))
        return self

    def print_networks(self, verbose):
        print("---------- Networks initialized -------------")
        for name in self.model_names:
            if isinstance(name, str):
                net = getattr(self, "net_" + name)
                num_params = sum(int(torch.tensor(s).prod()) for s in net.native_param_shapes().values())
                if verbose:
                    print(net)
                print("[Network %s] Total number of parameters : %.3f M" % (name, num_params / 1e6))
        print("-----------------------------------------------")

    def set_requires_grad(self, nets, requires_grad=False):
        """Kept for API compatibility; gradient flow is fixed by the native step (D weight
        gradients are simply not computed during backward_G: SURVEY.md quirk 5)."""
        pass
