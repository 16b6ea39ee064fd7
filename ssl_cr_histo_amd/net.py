"""Model objects with the reference's names, parameter order and state_dict keys (models/net.py:8-115), backed by the
HIP engine instead of torch ops.

They are ordinary ``nn.Module`` parameter containers -- ``named_parameters()`` order (freeze-by-index,
eval_BreastPathQ_SSL_CR.py:408-441), ``state_dict()/load_state_dict()`` keys and NCHW fp32 shapes (reference
checkpoints load unchanged, optional ``module.`` prefix handled by :func:`strip_module_prefix`), ``copy.deepcopy``
(teacher refresh, :515-516) all behave like the reference's modules.  ``forward`` runs the engine (forward only, no
autograd graph); training goes through :mod:`ssl_cr_histo_amd.steps`.
"""
from collections import OrderedDict

import torch
from torch import nn


class _BasicBlock(nn.Module):
    """parameter layout of torchvision's BasicBlock (conv1,bn1,conv2,bn2[,downsample.0,downsample.1])."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))


class ResNet18Params(nn.Module):
    """torchvision.models.resnet18(pretrained=False) with ``fc = Sequential()`` (models/net.py:32-34): same attribute
    tree, kaiming-normal(fan_out) conv init and BN weight 1 / bias 0 like torchvision 0.8.1.  Holds parameters only."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cfg = [(64, 64, 1), (64, 128, 2), (128, 256, 2), (256, 512, 2)]
        for i, (cin, cout, s) in enumerate(cfg, 1):
            setattr(self, f"layer{i}", nn.Sequential(_BasicBlock(cin, cout, s), _BasicBlock(cout, cout, 1)))
        self.fc = nn.Sequential()
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def bn_modules(self):
        """the 20 BatchNorm2d modules in graph order (matches the engine's bn index)."""
        out = [self.bn1]
        for i in range(1, 5):
            for blk in getattr(self, f"layer{i}"):
                out += [blk.bn1, blk.bn2]
                if blk.downsample is not None:
                    out.append(blk.downsample[1])
        return out

    def forward(self, x):
        raise RuntimeError("ResNet18Params is a parameter container; call the owning TripletNet(_Finetune)")


class _EngineModule(nn.Module):
    """forward-only engine call shared by TripletNet / TripletNet_Finetune."""
    triplet = False

    def _run(self, xs):
        from .engine import get_engine
        x0 = xs[0]
        if not x0.is_cuda:
            raise RuntimeError("ssl_cr_histo_amd modules run on the MI355X engine only: move inputs/parameters to cuda")
        eng = get_engine(x0.device)
        bound = eng.bind(self, None)
        return bound.forward(xs, train=self.training)[0]


class TripletNet(_EngineModule):
    """models/net.py:25-66 -- siamese ResNet18 on three tiles + pairwise fc (RSP pretraining)."""
    triplet = True

    def __init__(self, model="resnet18"):
        super().__init__()
        if model != "resnet18":
            raise NotImplementedError("not supported model type: {}".format(model))     # resnet50 is outside the hot path
        self.model = ResNet18Params()
        self.fc = nn.Sequential(nn.Linear(512 * 2, 512), nn.ReLU(True), nn.Linear(512, 256))

    def forward(self, i1, i2, i3):
        return self._run((i1, i2, i3))


class TripletNet_Finetune(_EngineModule):
    """models/net.py:70-103 -- the same weights, one input fed three times (computed once by the engine)."""

    def __init__(self, model="resnet18"):
        super().__init__()
        if model != "resnet18":
            raise NotImplementedError("not supported model type: {}".format(model))
        self.model = ResNet18Params()
        self.fc = nn.Sequential(nn.Linear(512 * 2, 512), nn.ReLU(True), nn.Linear(512, 256))

    def forward(self, i):
        return self._run((i,))


class Classifier(nn.Module):
    """models/net.py:8-20."""

    def __init__(self, in_features, num_classes):
        super().__init__()
        self.classifier = nn.Sequential(nn.Linear(in_features, 128), nn.ReLU(True), nn.Linear(128, num_classes))

    def forward(self, x):
        from . import kernels as K
        h = K.linear_fwd(x.contiguous(), self.classifier[0].weight, self.classifier[0].bias, relu=True)
        return K.linear_fwd(h, self.classifier[2].weight, self.classifier[2].bias)


class FinetuneResNet(nn.Module):
    """models/net.py:107-115."""

    def __init__(self, num_classes):
        super().__init__()
        self.classifier = nn.Sequential(nn.Linear(256 * 3, num_classes))

    def forward(self, x):
        from . import kernels as K
        return K.linear_fwd(x.contiguous(), self.classifier[0].weight, self.classifier[0].bias)


def strip_module_prefix(state_dict):
    """the reference saves DataParallel-wrapped modules and strips ``module.`` with ``k[7:]``
    (eval_Camelyon_SSL_CR.py:405-412); do the same only where the prefix is present."""
    return OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in state_dict.items())


def unwrap(m):
    """accept nn.DataParallel-wrapped modules like the reference's train() does (the engine is one process per GPU)."""
    return m.module if isinstance(m, nn.DataParallel) else m
