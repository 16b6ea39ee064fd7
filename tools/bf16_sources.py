"""Which bf16 roundings cost the gradient accuracy?  CPU emulation (oracle/bf16_emul.py's model of the engine's bf16 storage) with
the rounding switched on per tensor class and per stage: weights, forward activations, activation gradients.
    python tools/bf16_sources.py [N] [HW]
Prints the relative L2 error of selected parameter gradients against the un-rounded fp32 run of the same SSL_CR (MSE) iteration."""
import os
import sys
from collections import OrderedDict

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cases as C  # noqa: E402
from oracle import model as M  # noqa: E402


class _RF(torch.autograd.Function):            # round forward only
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g


class _RB(torch.autograd.Function):            # round backward only
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


def mk(fwd, bwd):
    def q(t):
        if fwd:
            t = _RF.apply(t)
        if bwd:
            t = _RB.apply(t)
        return t
    return q


STAGES = ["stem", "layer1", "layer2", "layer3", "layer4"]


def backbone(p, x, cfg, pre="model."):
    """cfg: {'w': set(stages), 'f': set(stages), 'b': set(stages)} -- where weights / forward activations / gradients are rounded."""
    def bn(t, k):
        return F.batch_norm(t, None, None, p[k + ".weight"], p[k + ".bias"], True, 0.1, 1e-5)

    def qa(stage):
        return mk(stage in cfg["f"], stage in cfg["b"])

    def w(k, stage):
        return _RF.apply(p[pre + k]) if stage in cfg["w"] else p[pre + k]
    q = qa("stem")
    x = q(F.conv2d(x, w("conv1.weight", "stem"), None, 2, 3))
    x = q(F.max_pool2d(F.relu(bn(x, pre + "bn1")), 3, 2, 1))
    for name, cin, cout, stride, ds in M.BLOCKS:
        st = name.split(".")[0]
        q = qa(st)
        n = pre + name
        o = q(F.conv2d(x, w(name + ".conv1.weight", st), None, stride, 1))
        o = q(F.relu(bn(o, n + ".bn1")))
        o = q(F.conv2d(o, w(name + ".conv2.weight", st), None, 1, 1))
        o = bn(o, n + ".bn2")
        if ds:
            i = q(F.conv2d(x, w(name + ".downsample.0.weight", st), None, stride, 0))
            i = bn(i, n + ".downsample.1")
        else:
            i = x
        x = q(F.relu(o + i))
    return torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)


def grads(p, x, y, u_s, lt, cfg):
    for v in p.values():
        v.grad = None
    e = backbone(p, torch.cat((x, u_s)), cfg)
    f = M.fc_head(p, torch.cat((e, e), 1))
    logits = M.classifier_forward(p, torch.cat((f, f, f), 1))
    nx = x.shape[0]
    loss = F.mse_loss(logits[:nx], y.view(-1, 1)) + F.mse_loss(lt, logits[nx:])
    loss.backward()
    return {k: v.grad.clone() for k, v in p.items()}


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    hw = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    torch.set_num_threads(8)
    sd = M.init_state(C.PARAM_SEED, M.net_param_specs(), random_running_stats=True)
    csd = M.init_state(C.PARAM_SEED + 1, M.classifier_param_specs("finetune", 1))
    pn, _ = M.split_state(sd)
    pc, _ = M.split_state(csd)
    p = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in list(pn.items()) + list(pc.items()))
    nx = N * 3 // 10
    nu = N - nx
    x, u_s = C.u8(1, (nx, 3, hw, hw)).float(), C.u8(2, (nu, 3, hw, hw)).float()
    y = C.f32(3, (nx,))
    lt = C.f32(4, (nu, 1)) * 30.0
    none, allst = set(), set(STAGES)
    ref = grads(p, x, y, u_s, lt, dict(w=none, f=none, b=none))
    keys = ["model.conv1.weight", "model.layer1.0.conv1.weight", "model.layer1.1.conv2.weight", "model.layer2.0.conv1.weight",
            "model.layer3.0.conv1.weight", "model.layer4.0.conv1.weight", "model.layer4.1.conv2.weight", "fc.0.weight"]
    early = {"stem", "layer1"}
    cfgs = OrderedDict([
        ("all rounded (engine bf16 mode)", dict(w=allst, f=allst, b=allst)),
        ("weights only", dict(w=allst, f=none, b=none)),
        ("forward activations only", dict(w=none, f=allst, b=none)),
        ("gradients only", dict(w=none, f=none, b=allst)),
        ("all but gradients in stem+layer1", dict(w=allst, f=allst, b=allst - early)),
        ("all but gradients in stem..layer2", dict(w=allst, f=allst, b=allst - early - {"layer2"})),
        ("all but any gradient", dict(w=allst, f=allst, b=none)),
        ("all but fwd activations in stem+layer1", dict(w=allst, f=allst - early, b=allst)),
        ("all but fwd+grad in stem+layer1", dict(w=allst, f=allst - early, b=allst - early)),
        ("all but forward activations", dict(w=allst, f=none, b=allst)),
        ("all but weights", dict(w=none, f=allst, b=allst)),
    ])
    print(f"N={N} ({nx}+{nu}) {hw}x{hw}; relative L2 error of the gradient vs the un-rounded fp32 run")
    print(" " * 42 + " ".join(f"{k.replace('model.', '').replace('.weight', ''):>14s}" for k in keys))
    for name, cfg in cfgs.items():
        g = grads(p, x, y, u_s, lt, cfg)
        print(f"{name:42s}" + " ".join(f"{float((g[k] - ref[k]).norm() / ref[k].norm()):14.3e}" for k in keys))


if __name__ == "__main__":
    main()
