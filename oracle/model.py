"""Functional fp32 restatement of the reference model graphs (test oracle).

Parameters travel as an ``OrderedDict[str, torch.Tensor]`` whose keys and shapes are
exactly the reference ``state_dict()`` keys (``models/net.py``): ``model.*`` is the
torchvision-0.8.1 ResNet18 with ``fc = Sequential()`` (``models/net.py:32-34,77-79``),
``fc.0/fc.2`` the pairwise head (``models/net.py:35-36,80-81``), ``classifier.*`` the task
head (``models/net.py:12-15`` Classifier, ``:111`` FinetuneResNet).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # torch.nn.BatchNorm2d default, used by torchvision resnet
BN_MOMENTUM = 0.1

# (name, cin, cout, stride, has_downsample) -- torchvision resnet18 BasicBlock layout
BLOCKS = [
    ("layer1.0", 64, 64, 1, False), ("layer1.1", 64, 64, 1, False),
    ("layer2.0", 64, 128, 2, True), ("layer2.1", 128, 128, 1, False),
    ("layer3.0", 128, 256, 2, True), ("layer3.1", 256, 256, 1, False),
    ("layer4.0", 256, 512, 2, True), ("layer4.1", 512, 512, 1, False),
]


def backbone_param_specs():
    """[(key, shape, kind)] in ``named_parameters()`` order (indices 0..59)."""
    specs = [("conv1.weight", (64, 3, 7, 7), "conv"),
             ("bn1.weight", (64,), "bn_w"), ("bn1.bias", (64,), "bn_b")]
    for name, cin, cout, stride, ds in BLOCKS:
        specs += [(f"{name}.conv1.weight", (cout, cin, 3, 3), "conv"),
                  (f"{name}.bn1.weight", (cout,), "bn_w"), (f"{name}.bn1.bias", (cout,), "bn_b"),
                  (f"{name}.conv2.weight", (cout, cout, 3, 3), "conv"),
                  (f"{name}.bn2.weight", (cout,), "bn_w"), (f"{name}.bn2.bias", (cout,), "bn_b")]
        if ds:
            specs += [(f"{name}.downsample.0.weight", (cout, cin, 1, 1), "conv"),
                      (f"{name}.downsample.1.weight", (cout,), "bn_w"),
                      (f"{name}.downsample.1.bias", (cout,), "bn_b")]
    return specs


def bn_names():
    """The 20 BatchNorm layers (key prefix under ``model.``) and channel counts, in graph order."""
    out = [("bn1", 64)]
    for name, cin, cout, stride, ds in BLOCKS:
        out += [(f"{name}.bn1", cout), (f"{name}.bn2", cout)]
        if ds:
            out.append((f"{name}.downsample.1", cout))
    return out


def net_param_specs():
    """TripletNet / TripletNet_Finetune parameters, ``named_parameters()`` order (0..63)."""
    specs = [("model." + k, s, kind) for k, s, kind in backbone_param_specs()]
    specs += [("fc.0.weight", (512, 1024), "lin_w"), ("fc.0.bias", (512,), "lin_b"),
              ("fc.2.weight", (256, 512), "lin_w"), ("fc.2.bias", (256,), "lin_b")]
    return specs


def classifier_param_specs(kind, num_classes):
    """``kind='finetune'``: FinetuneResNet (models/net.py:107-115); ``'mlp'``: Classifier (:8-20)."""
    if kind == "finetune":
        return [("classifier.0.weight", (num_classes, 768), "lin_w"),
                ("classifier.0.bias", (num_classes,), "lin_b")]
    return [("classifier.0.weight", (128, 768), "lin_w"), ("classifier.0.bias", (128,), "lin_b"),
            ("classifier.2.weight", (num_classes, 128), "lin_w"),
            ("classifier.2.bias", (num_classes,), "lin_b")]


def _draw(rs, shape, kind):
    if kind == "conv":       # kaiming-normal fan_out like torchvision, via a version-stable RNG
        fan_out = shape[0] * shape[2] * shape[3]
        return (rs.standard_normal(shape) * np.sqrt(2.0 / fan_out)).astype(np.float32)
    if kind == "bn_w":       # not the default 1/0: a loaded checkpoint has arbitrary affine
        return rs.uniform(0.5, 1.5, shape).astype(np.float32)
    if kind == "bn_b":
        return (rs.standard_normal(shape) * 0.1).astype(np.float32)
    if kind == "lin_w":
        bound = 1.0 / np.sqrt(shape[1])
        return rs.uniform(-bound, bound, shape).astype(np.float32)
    if kind == "lin_b":
        return rs.uniform(-0.05, 0.05, shape).astype(np.float32)
    raise ValueError(kind)


def init_state(seed, specs, with_bn_buffers=True, random_running_stats=False):
    """Seeded state_dict (numpy legacy ``RandomState`` => identical on every box/version).

    Returns ``OrderedDict`` of torch fp32 tensors (``num_batches_tracked`` int64) with the
    reference's key order (params of a BN followed by its three buffers).
    """
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for key, shape, kind in specs:
        sd[key] = torch.from_numpy(_draw(rs, shape, kind))
        if kind == "bn_b" and with_bn_buffers:
            base = key[:-len("bias")]
            c = shape[0]
            if random_running_stats:
                sd[base + "running_mean"] = torch.from_numpy((rs.standard_normal(c) * 0.5).astype(np.float32))
                sd[base + "running_var"] = torch.from_numpy(rs.uniform(0.5, 2.0, c).astype(np.float32))
            else:
                sd[base + "running_mean"] = torch.zeros(c)
                sd[base + "running_var"] = torch.ones(c)
            sd[base + "num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    return sd


def split_state(sd):
    """-> (params requiring grad by default, buffers)."""
    params, bufs = OrderedDict(), OrderedDict()
    for k, v in sd.items():
        (bufs if ("running_" in k or "num_batches" in k) else params)[k] = v
    return params, bufs


# --------------------------------------------------------------------------------------------
# forward graphs
# --------------------------------------------------------------------------------------------
def _bn(x, p, b, prefix, train):
    """BatchNorm2d: train => batch stats + running update (momentum .1, unbiased running var)."""
    rm, rv = b[prefix + ".running_mean"], b[prefix + ".running_var"]
    y = F.batch_norm(x, rm, rv, p[prefix + ".weight"], p[prefix + ".bias"], train, BN_MOMENTUM, BN_EPS)
    if train:
        b[prefix + ".num_batches_tracked"] += 1
    return y


def backbone_forward(p, b, x, train, pre="model.", taps=None):
    """torchvision resnet18 minus fc: conv1-bn1-relu-maxpool-layer1..4-avgpool-flatten -> [N,512]."""
    x = F.conv2d(x, p[pre + "conv1.weight"], None, 2, 3)
    x = F.relu(_bn(x, p, b, pre + "bn1", train))
    x = F.max_pool2d(x, 3, 2, 1)
    if taps is not None:
        taps["stem"] = x
    for name, cin, cout, stride, ds in BLOCKS:
        q = pre + name
        out = F.conv2d(x, p[q + ".conv1.weight"], None, stride, 1)
        out = F.relu(_bn(out, p, b, q + ".bn1", train))
        out = F.conv2d(out, p[q + ".conv2.weight"], None, 1, 1)
        out = _bn(out, p, b, q + ".bn2", train)
        if ds:
            idn = F.conv2d(x, p[q + ".downsample.0.weight"], None, stride, 0)
            idn = _bn(idn, p, b, q + ".downsample.1", train)
        else:
            idn = x
        x = F.relu(out + idn)
        if taps is not None:
            taps[name] = x
    return torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)


def fc_head(p, e):
    """models/net.py:35-36 : Linear(1024,512)-ReLU-Linear(512,256)."""
    h = F.relu(F.linear(e, p["fc.0.weight"], p["fc.0.bias"]))
    return F.linear(h, p["fc.2.weight"], p["fc.2.bias"])


def triplet_forward(p, b, i1, i2, i3, train):
    """TripletNet.forward, models/net.py:50-66 -- three backbone passes, pairwise cat, shared fc."""
    e1 = backbone_forward(p, b, i1, train)
    e2 = backbone_forward(p, b, i2, train)
    e3 = backbone_forward(p, b, i3, train)
    f12 = fc_head(p, torch.cat((e1, e2), 1))
    f23 = fc_head(p, torch.cat((e2, e3), 1))
    f13 = fc_head(p, torch.cat((e1, e3), 1))
    return torch.cat((f12, f23, f13), 1)


def finetune_forward(p, b, i, train, faithful=True):
    """TripletNet_Finetune.forward, models/net.py:86-103.

    ``faithful=True`` runs the backbone three times on the same input exactly like the
    reference (BN running stats updated 3x).  ``faithful=False`` is the de-triplicated
    form the engine uses: one pass, the running-stat update replayed 3x.
    """
    if faithful:
        return triplet_forward(p, b, i, i, i, train)
    if train:
        snap = {k: v.clone() for k, v in b.items()}
    e = backbone_forward(p, b, i, train)
    if train:
        for k in b:
            if k.endswith("running_mean") or k.endswith("running_var"):
                # r1 = (1-m) r0 + m s  =>  s = (r1 - (1-m) r0)/m ; apply twice more
                r0, r1 = snap[k], b[k]
                s = (r1 - (1 - BN_MOMENTUM) * r0) / BN_MOMENTUM
                r = r1
                for _ in range(2):
                    r = (1 - BN_MOMENTUM) * r + BN_MOMENTUM * s
                b[k] = r
            elif k.endswith("num_batches_tracked"):
                b[k] = b[k] + 2
    f = fc_head(p, torch.cat((e, e), 1))
    return torch.cat((f, f, f), 1)


def classifier_forward(p, feats):
    """FinetuneResNet (1 Linear) or Classifier (Linear-ReLU-Linear), by the keys present."""
    if "classifier.2.weight" in p:
        h = F.relu(F.linear(feats, p["classifier.0.weight"], p["classifier.0.bias"]))
        return F.linear(h, p["classifier.2.weight"], p["classifier.2.bias"])
    return F.linear(feats, p["classifier.0.weight"], p["classifier.0.bias"])
