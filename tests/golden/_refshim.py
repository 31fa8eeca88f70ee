"""Import the *unmodified* reference from /root/reference inside this (GPU-less, torchvision-less)
container, so golden vectors can be produced by executing the reference's own code.

TEST INFRASTRUCTURE ONLY.  Nothing here is used by the product path or on the GPU box
(/root/reference does not exist there).  No reference source is copied: the reference modules are
imported from where they lie; the third-party packages it needs but this image lacks
(cv2, imageio, wandb, skimage, gdown, torchvision, timm) are replaced by minimal stand-ins whose
arithmetic is either irrelevant to the hot path (cv2/imageio/wandb/...) or restated from the public
definition (torchvision ResNet topology, tensor Resize == F.interpolate(bicubic, antialias);
timm DropPath / trunc_normal_).  See SURVEY.md section 8(c).
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))
_PRODUCT = os.path.normpath(os.path.join(_HERE, "..", "..", "dynamo-depth_amd"))


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "Trainer.py"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load_product_resnet():
    """Product ResNet blocks, loaded by file path under a private name (no `networks` clash)."""
    spec = importlib.util.spec_from_file_location("_dd_product_resnet", os.path.join(_PRODUCT, "networks", "resnet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _install_torchvision():
    pr = _load_product_resnet()

    class TVResNet(nn.Module):
        """torchvision-signature ResNet(block, layers) built from the product's blocks."""

        def __init__(self, block, layers, num_classes=1000):
            super().__init__()
            self.inplanes = 64
            self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
            self.layer1 = self._make_layer(block, 64, layers[0])
            self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
            self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
            self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
            self.fc = nn.Linear(512 * block.expansion, num_classes)
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                elif isinstance(m, nn.BatchNorm2d):
                    nn.init.constant_(m.weight, 1)
                    nn.init.constant_(m.bias, 0)

        def _make_layer(self, block, planes, blocks, stride=1):
            down = None
            if stride != 1 or self.inplanes != planes * block.expansion:
                down = nn.Sequential(
                    nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                    nn.BatchNorm2d(planes * block.expansion))
            seq = [block(self.inplanes, planes, stride, down)]
            self.inplanes = planes * block.expansion
            seq += [block(self.inplanes, planes) for _ in range(1, blocks)]
            return nn.Sequential(*seq)

    def _factory(block, layers):
        def make(pretrained=False, weights=None, **kw):
            assert not pretrained and weights is None, "no network: pretrained weights unavailable"
            return TVResNet(block, layers)
        return make

    class _W:  # stands for torchvision.models.ResNetXX_Weights
        IMAGENET1K_V1 = None

    resnet_mod = _mod("torchvision.models.resnet", BasicBlock=pr.BasicBlock, Bottleneck=pr.Bottleneck, ResNet=TVResNet)
    models = _mod(
        "torchvision.models", ResNet=TVResNet, resnet=resnet_mod,
        resnet18=_factory(pr.BasicBlock, [2, 2, 2, 2]), resnet34=_factory(pr.BasicBlock, [3, 4, 6, 3]),
        resnet50=_factory(pr.Bottleneck, [3, 4, 6, 3]), resnet101=_factory(pr.Bottleneck, [3, 4, 23, 3]),
        resnet152=_factory(pr.Bottleneck, [3, 8, 36, 3]),
        ResNet18_Weights=_W, ResNet34_Weights=_W, ResNet50_Weights=_W, ResNet101_Weights=_W, ResNet152_Weights=_W)

    class InterpolationMode:
        BICUBIC = "bicubic"
        BILINEAR = "bilinear"
        NEAREST = "nearest"

    class Resize:
        def __init__(self, size, interpolation="bilinear", antialias=None):
            self.size, self.mode, self.antialias = tuple(size), interpolation, bool(antialias)

        def __call__(self, img):
            if not torch.is_tensor(img):
                from PIL import Image
                return img.resize(self.size[::-1], Image.BICUBIC)
            kw = {} if self.mode == "nearest" else {"align_corners": False, "antialias": self.antialias}
            return F.interpolate(img, self.size, mode=self.mode, **kw)

    class ToTensor:
        def __call__(self, pic):
            import numpy as np
            arr = np.asarray(pic, dtype=np.uint8)
            return torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div(255)

    class ColorJitter:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            raise RuntimeError("ColorJitter stand-in: goldens are generated with augmentation off")

    transforms = _mod("torchvision.transforms", Resize=Resize, InterpolationMode=InterpolationMode,
                      ToTensor=ToTensor, ColorJitter=ColorJitter)
    tv_datasets = _mod("torchvision.datasets")
    _mod("torchvision", models=models, transforms=transforms, datasets=tv_datasets)


def _install_timm():
    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            return x * mask / keep

    layers = _mod("timm.models.layers", DropPath=DropPath, trunc_normal_=nn.init.trunc_normal_)
    models = _mod("timm.models", layers=layers)
    _mod("timm", models=models)


def _install_misc():
    _mod("cv2", INTER_NEAREST=0)
    _mod("imageio")
    _mod("gdown")
    tr = _mod("skimage.transform")
    _mod("skimage", transform=tr)

    class _Img:
        def __init__(self, *a, **k):
            pass

    _mod("wandb", init=lambda *a, **k: None, log=lambda *a, **k: None, Image=_Img)
    try:
        import tqdm  # noqa: F401
    except Exception:
        _mod("tqdm", tqdm=lambda x, **k: x)
    from PIL import Image
    if not hasattr(Image, "ANTIALIAS"):
        Image.ANTIALIAS = Image.LANCZOS


_REF_TOPLEVEL = ("options", "tools", "utils", "Trainer", "networks", "datasets", "train")


def import_reference():
    """Returns a namespace with the reference's modules (options, tools, utils, Trainer, networks, datasets)."""
    if not reference_available():
        raise RuntimeError("reference not mounted at " + REFERENCE_ROOT)
    _install_torchvision()
    _install_timm()
    _install_misc()
    # Trainer.py:32 asserts cuda_id < device_count() *before* its CPU fallback at :33
    torch.cuda.device_count = lambda: 1
    saved = {k: sys.modules.pop(k) for k in list(sys.modules)
             if k.split(".")[0] in _REF_TOPLEVEL}
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        ns = types.SimpleNamespace()
        for name in ("options", "tools", "utils", "networks", "datasets", "Trainer"):
            setattr(ns, name, importlib.import_module(name))
        for name, mod in list(sys.modules.items()):
            if name.split(".")[0] in _REF_TOPLEVEL:
                assert (getattr(mod, "__file__", "") or "").startswith(REFERENCE_ROOT), name
    finally:
        sys.path.remove(REFERENCE_ROOT)
        # leave the reference modules importable only through `ns`; restore whatever was there
        for k in list(sys.modules):
            if k.split(".")[0] in _REF_TOPLEVEL:
                del sys.modules[k]
        sys.modules.update(saved)
    return ns


def make_opt(ref, **over):
    """Reference options (options.py:270-303) for a CPU run with scratch weights."""
    argv = over.pop("argv", [])
    opt = ref.options.DynamoOptions().parse(args=argv)
    opt.weights_init = "scratch"
    opt.num_workers = 0
    opt.ddp = False
    opt.local_world_size = 1
    opt.print_opt = False
    opt.log_dir = over.pop("log_dir", "/tmp/dd_ref_logs")
    for k, v in over.items():
        setattr(opt, k, v)
    return opt
