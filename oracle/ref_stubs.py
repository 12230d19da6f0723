"""TEST INFRASTRUCTURE ONLY -- never imported by the swapnet_amd product path.

Makes the *real* reference (andrewjong/SwapNet, mounted read-only at
/root/reference) importable inside the build container so that
oracle/make_golden.py can run its own WarpModel / TextureModel on CPU and
record golden vectors.  /root/reference does not exist
on the GPU box: nothing under tests/ -m gpu, bench.py or smoke() imports this.

Five third-party modules the reference imports are absent from this image
(torchvision, adabound, seaborn, dominate, visdom).  They are replaced by
stand-ins in sys.modules *before* the reference is imported:

  torchvision.ops.RoIAlign     -> oracle.swapnet_oracle.roi_align  (restatement of
                                  torchvision 0.4.0 ROIAlign_cpu.cpp, "parity unpinned")
  torchvision.models.vgg16/19  -> same layer list (cfg D / E), seeded random weights
                                  (pretrained weights are not obtainable offline)
  torchvision.transforms, adabound, seaborn, dominate, visdom -> inert placeholders
"""
import sys
import types

import torch
from torch import nn

REFERENCE_ROOT = "/root/reference"

VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M",
             512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


def make_vgg_features(cfg=None):
    """torchvision.models.vgg16().features layer list (conv3x3 pad1 + ReLU(inplace),
    maxpool 2x2) filled with the oracle's *seeded random* weights
    (oracle.swapnet_oracle.vgg16_feature_params; private generator, so the global
    RNG stream used for kaiming init and the smooth labels is untouched)."""
    from oracle.swapnet_oracle import VGG16_CFG, vgg16_feature_params
    cfg = cfg or VGG16_CFG
    params = vgg16_feature_params() if cfg == VGG16_CFG else None
    layers, cin, ci = [], 3, 0
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            conv = nn.Conv2d(cin, v, kernel_size=3, padding=1)
            if params is not None:
                with torch.no_grad():
                    conv.weight.copy_(params[ci][0])
                    conv.bias.copy_(params[ci][1])
            layers += [conv, nn.ReLU(inplace=True)]
            cin = v
            ci += 1
    return nn.Sequential(*layers)


class _VGG(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.features = make_vgg_features(cfg)


def _vgg16(pretrained=False, **kw):
    return _VGG(None)


def _vgg19(pretrained=False, **kw):
    return _VGG(VGG19_CFG)


class _RoIAlignStub(nn.Module):
    """Signature of torchvision.ops.RoIAlign @0.4.0 (no `aligned` argument)."""

    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    def forward(self, input, rois):
        from oracle.swapnet_oracle import roi_align
        return roi_align(input, rois, self.output_size, self.spatial_scale,
                         self.sampling_ratio)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Inert:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Inert()

    def __enter__(self):            # `with dominate.document(...)`, `with tr():` (util/html.py)
        return self

    def __exit__(self, *exc):
        return False

    def render(self, *a, **k):
        return "<html></html>"


class _ToTensor:
    """transforms.ToTensor: PIL / HxWxC uint8 -> float CxHxW in [0,1]"""

    def __call__(self, pic):
        import numpy as np
        a = np.asarray(pic)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(a.copy()).permute(2, 0, 1)
        return t.float().div(255) if t.dtype == torch.uint8 else t.float()


class _Normalize:
    """transforms.Normalize(mean, std)"""

    def __init__(self, mean, std, inplace=False):
        self.mean, self.std = torch.tensor(list(mean), dtype=torch.float32), torch.tensor(list(std), dtype=torch.float32)

    def __call__(self, t):
        return (t - self.mean[:, None, None]) / self.std[:, None, None]


def _tf_resize(img, size, interpolation=2):
    """transforms.functional.resize on a PIL image: int = the smaller edge (torchvision semantics), bilinear"""
    w, h = img.size
    if isinstance(size, int):
        if (w <= h and w == size) or (h <= w and h == size):
            return img
        size = (size, int(size * h / w)) if w < h else (int(size * w / h), size)
    else:
        size = (size[1], size[0])
    return img.resize(size, 2)


def _tf_hflip(img):
    from PIL import Image
    return img.transpose(Image.FLIP_LEFT_RIGHT)


def _tf_vflip(img):
    from PIL import Image
    return img.transpose(Image.FLIP_TOP_BOTTOM)


def install(functional_transforms=False):
    """Install the stand-ins and put the reference first on sys.path.  functional_transforms=True makes ToTensor /
    Normalize real (needed when the reference's own dataloader is run, tests/test_reference_scripts.py)."""
    if functional_transforms and "torchvision.transforms" in sys.modules:
        sys.modules["torchvision.transforms"].ToTensor = _ToTensor
        sys.modules["torchvision.transforms"].Normalize = _Normalize
    if "torchvision" not in sys.modules:
        tv = _mod("torchvision")
        tv.ops = _mod("torchvision.ops", RoIAlign=_RoIAlignStub)
        tv.models = _mod("torchvision.models", vgg16=_vgg16, vgg19=_vgg19)
        tf = _mod("torchvision.transforms")
        for n in ("Compose", "ToTensor", "Normalize", "RandomAffine", "RandomPerspective",
                  "RandomHorizontalFlip", "Resize", "CenterCrop", "RandomCrop", "Lambda",
                  "ToPILImage", "ColorJitter", "Pad"):
            setattr(tf, n, _Inert)
        if functional_transforms:
            tf.ToTensor, tf.Normalize = _ToTensor, _Normalize
        tf.functional = _mod("torchvision.transforms.functional")
        if functional_transforms:       # what datasets/texture_dataset.py and data_utils.random_image_roi_flip call
            tf.functional.to_tensor = _ToTensor()
            tf.functional.resize = _tf_resize
            tf.functional.hflip, tf.functional.vflip = _tf_hflip, _tf_vflip
        tf.transforms = tf                      # `from torchvision.transforms import transforms`
        sys.modules["torchvision.transforms.transforms"] = tf
        tv.transforms = tf
        tv.utils = _mod("torchvision.utils", make_grid=_Inert(), save_image=_Inert())
    if "adabound" not in sys.modules:
        _mod("adabound", AdaBound=_Inert)
    if "seaborn" not in sys.modules:
        _mod("seaborn", color_palette=lambda *a, **k: [(0, 0, 0)] * 12)
    if "dominate" not in sys.modules:
        d = _mod("dominate", document=_Inert)
        d.tags = _mod("dominate.tags")
        for n in ("meta", "h3", "table", "tr", "td", "p", "a", "img", "br"):
            setattr(d.tags, n, _Inert)
    if "visdom" not in sys.modules:
        _mod("visdom", Visdom=_Inert)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # the pip package `datasets` (HuggingFace) would shadow the reference's datasets/
    for name in list(sys.modules):
        if name == "datasets" or name.startswith("datasets."):
            if not getattr(sys.modules[name], "__file__", "").startswith(REFERENCE_ROOT):
                del sys.modules[name]
