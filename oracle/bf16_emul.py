"""bf16-storage emulation of the student backbone on CPU (test oracle).

The engine's bf16 mode stores activations, conv weights and activation gradients in bf16 (fp32 accumulate, fp32
BatchNorm statistics, fp32 heads/losses/optimizer).  On small, randomly initialised problems that rounding alone moves
gradients by tens of percent relative to fp32 (BatchNorm backward subtracts large projections; ReLU / max-pool masks
flip).  This module reproduces the SAME KIND of rounding with plain torch CPU ops -- round-to-bf16 in forward and in
backward at every tensor the engine materialises -- so the tests can tell "bf16 noise of the expected size" from a
kernel bug: the engine's error against the fp32 oracle must stay within a small factor of this emulation's error.
It is a yardstick for tolerances only; parity is carried by the fp32 engine mode.
"""
import torch
import torch.nn.functional as F

from . import model as M


class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


rnd = _Round.apply


def _bn(x, p, pre):
    return F.batch_norm(x, None, None, p[pre + ".weight"], p[pre + ".bias"], True, 0.1, 1e-5)


def backbone_train(p, x, q=rnd, pre="model."):
    """train-mode resnet18 backbone with rounding `q` where the engine stores bf16 (q = identity gives fp32)."""
    w = (lambda k: q(p[pre + k]))
    x = q(F.conv2d(x, w("conv1.weight"), None, 2, 3))
    x = q(F.max_pool2d(F.relu(_bn(x, p, pre + "bn1")), 3, 2, 1))
    for name, cin, cout, stride, ds in M.BLOCKS:
        n = pre + name
        o = q(F.conv2d(x, w(name + ".conv1.weight"), None, stride, 1))
        o = q(F.relu(_bn(o, p, n + ".bn1")))                 # rounded on the consumer's load path
        o = q(F.conv2d(o, w(name + ".conv2.weight"), None, 1, 1))
        o = _bn(o, p, n + ".bn2")
        if ds:
            i = q(F.conv2d(x, w(name + ".downsample.0.weight"), None, stride, 0))
            i = _bn(i, p, n + ".downsample.1")
        else:
            i = x
        x = q(F.relu(o + i))
    return torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)


def ssl_cr_grads(kind, p, x, y, u_s, logits_t, lambda_u, emulate):
    """gradients of the SSL_CR loss w.r.t. every entry of `p` (leaf tensors), bf16-emulated or plain fp32."""
    for v in p.values():
        v.grad = None
    q = rnd if emulate else (lambda t: t)
    e = backbone_train(p, torch.cat((x, u_s)), q)
    f = M.fc_head(p, torch.cat((e, e), 1))
    logits = M.classifier_forward(p, torch.cat((f, f, f), 1))
    nx = x.shape[0]
    if kind == "mse":
        loss = F.mse_loss(logits[:nx], y.view(-1, 1)) + lambda_u * F.mse_loss(logits_t, logits[nx:])
    else:
        loss = F.cross_entropy(logits[:nx], y) + lambda_u * F.cross_entropy(logits[nx:], torch.softmax(logits_t, -1).max(-1)[1])
    loss.backward()
    return {k: v.grad.clone() for k, v in p.items()}, logits.detach(), float(loss.detach())


def rsp_grads(p, i1, i2, i3, target, emulate):
    """gradients of the RSP pre-training loss (pretrain_BreastPathQ.py:42-61: TripletNet -> Classifier -> CrossEntropyLoss)
    w.r.t. every entry of `p`; three backbone passes with their OWN batch statistics (models/net.py:50-66)."""
    for v in p.values():
        v.grad = None
    q = rnd if emulate else (lambda t: t)
    e1, e2, e3 = (backbone_train(p, i, q) for i in (i1, i2, i3))
    f = torch.cat((M.fc_head(p, torch.cat((e1, e2), 1)), M.fc_head(p, torch.cat((e2, e3), 1)), M.fc_head(p, torch.cat((e1, e3), 1))), 1)
    logits = M.classifier_forward(p, f)
    loss = F.cross_entropy(logits, target)
    loss.backward()
    return {k: v.grad.clone() for k, v in p.items()}, logits.detach(), float(loss.detach())


def sup_grads(kind, p, x, y, emulate):
    """gradients of the supervised fine-tuning loss (eval_Kather_SSL.py:51-79 / eval_Camelyon_SSL.py:52-98 'ce',
    eval_BreastPathQ_SSL.py:52-84 'mse') w.r.t. every entry of `p` (TripletNet_Finetune de-triplicated)."""
    for v in p.values():
        v.grad = None
    q = rnd if emulate else (lambda t: t)
    e = backbone_train(p, x, q)
    f = M.fc_head(p, torch.cat((e, e), 1))
    logits = M.classifier_forward(p, torch.cat((f, f, f), 1))
    loss = F.mse_loss(logits, y.view(-1, 1)) if kind == "mse" else F.cross_entropy(logits, y)
    loss.backward()
    return {k: v.grad.clone() for k, v in p.items()}, logits.detach(), float(loss.detach())


def backbone_eval(p, b, x, q=rnd, pre="model.", fold=False):
    """eval-mode resnet18 backbone the way the engine's bf16 mode runs the TEACHER (and validate()): every conv output stored in bf16
    after its scale / bias / residual / ReLU epilogue (q = identity gives fp32).  Round 6: the filters are the PLAIN weights rounded
    to bf16 and BatchNorm's scale gamma / sqrt(var + eps) is applied in fp32 in the epilogue (sslcr_conv_desc.out_scale).  fold=True is the engine of rounds 1-5: the scale folded into the
    filters BEFORE the rounding (tools/bf16_teacher_fold_experiment.py compares the two)."""
    def conv(x, cname, bname, stride, pad):
        s = p[pre + bname + ".weight"] / torch.sqrt(b[pre + bname + ".running_var"] + 1e-5)
        sh = p[pre + bname + ".bias"] - b[pre + bname + ".running_mean"] * s
        if fold:
            return F.conv2d(x, q(p[pre + cname + ".weight"] * s.view(-1, 1, 1, 1)), sh, stride, pad)
        return F.conv2d(x, q(p[pre + cname + ".weight"]), None, stride, pad) * s.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    x = q(x)
    x = F.max_pool2d(q(F.relu(conv(x, "conv1", "bn1", 2, 3))), 3, 2, 1)
    for name, cin, cout, stride, ds in M.BLOCKS:
        o = q(F.relu(conv(x, name + ".conv1", name + ".bn1", stride, 1)))
        i = q(conv(x, name + ".downsample.0", name + ".downsample.1", stride, 0)) if ds else x
        x = q(F.relu(conv(o, name + ".conv2", name + ".bn2", 1, 1) + i))
    return torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)


def teacher_logits(p, b, u_w, emulate, chunk=64):
    """the teacher's logits on the weakly augmented unlabeled batch (eval_BreastPathQ_SSL_CR.py:115-121: eval mode,
    no_grad, TripletNet_Finetune de-triplicated), bf16-emulated or plain."""
    q = rnd if emulate else (lambda t: t)
    with torch.no_grad():
        out = []
        for i in range(0, u_w.shape[0], chunk):
            e = backbone_eval(p, b, u_w[i:i + chunk], q)
            f = M.fc_head(p, torch.cat((e, e), 1))
            out.append(M.classifier_forward(p, torch.cat((f, f, f), 1)))
        return torch.cat(out)


# ---------------------------------------------------------------- whole epochs under bf16 storage (round 6)
def backbone_forward_emulated(p, b, x, train, pre="model.", taps=None):
    """drop-in for oracle.model.backbone_forward with the engine's bf16 STORAGE points rounded: train mode like backbone_train
    (and BatchNorm's running statistics updated from the rounded tensors, as oracle.model._bn does), eval mode like backbone_eval
    (plain bf16 filters, BatchNorm's scale and shift in the fp32 epilogue)."""
    if not train:
        return backbone_eval(p, b, x, rnd, pre)
    q = rnd
    w = (lambda k: q(p[pre + k]))
    x = q(F.conv2d(x, w("conv1.weight"), None, 2, 3))
    x = q(F.max_pool2d(F.relu(M._bn(x, p, b, pre + "bn1", True)), 3, 2, 1))
    for name, cin, cout, stride, ds in M.BLOCKS:
        n = pre + name
        o = q(F.conv2d(x, w(name + ".conv1.weight"), None, stride, 1))
        o = q(F.relu(M._bn(o, p, b, n + ".bn1", True)))
        o = q(F.conv2d(o, w(name + ".conv2.weight"), None, 1, 1))
        o = M._bn(o, p, b, n + ".bn2", True)
        if ds:
            i = q(F.conv2d(x, w(name + ".downsample.0.weight"), None, stride, 0))
            i = M._bn(i, p, b, n + ".downsample.1", True)
        else:
            i = x
        x = q(F.relu(o + i))
    return torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)


class emulating:
    """`with emulating():` -- every oracle.model forward inside (steps / epochs: train(), validate()) stores what the engine's bf16
    mode stores in bf16.  A yardstick for tolerances (tests/golden/make_bf16_yard_small.py), never a parity claim."""

    def __enter__(self):
        self._saved = M.backbone_forward
        M.backbone_forward = backbone_forward_emulated
        return self

    def __exit__(self, *exc):
        M.backbone_forward = self._saved
        return False
