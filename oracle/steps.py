"""fp32 restatement of the reference step bodies + optimizers (test oracle, torch CPU).

Every function works on explicit dicts:  ``p`` = parameters (leaf tensors, ``requires_grad``
set by the caller = the reference's freeze-by-index, ``eval_BreastPathQ_SSL_CR.py:408-441``),
``b`` = BN buffers, updated in place like ``nn.BatchNorm2d`` does.
"""
import math

import torch
import torch.nn.functional as F

from . import model as M


def apply_freeze(p_net, modules):
    """requires_grad = (index >= modules) over named_parameters() order
    (eval_BreastPathQ_SSL_CR.py:433-441)."""
    for idx, (k, v) in enumerate(p_net.items()):
        v.requires_grad_(idx >= modules)


# ------------------------------------------------------------------ optimizers (explicit math)
class Adam:
    """torch.optim.Adam semantics as the reference uses it (eval_BreastPathQ_SSL_CR.py:481):
    L2 weight decay added to the gradient, bias-corrected, eps outside the sqrt."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [q for q in params if q.requires_grad]
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, betas[0], betas[1], eps, weight_decay
        self.m = [torch.zeros_like(q) for q in self.params]
        self.v = [torch.zeros_like(q) for q in self.params]
        self.t = 0

    def step(self):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        with torch.no_grad():
            for q, m, v in zip(self.params, self.m, self.v):
                if q.grad is None:
                    continue
                g = q.grad + self.wd * q
                m.mul_(self.b1).add_(g, alpha=1 - self.b1)
                v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
                q.addcdiv_(m, denom, value=-self.lr / bc1)

    def zero_grad(self):
        for q in self.params:
            q.grad = None


class SGDNesterov:
    """torch.optim.SGD(momentum, nesterov=True, weight_decay) (eval_Camelyon_SSL_CR.py:514,
    pretrain_BreastPathQ.py:245): g += wd*p; buf = mu*buf + g (buf=g on first step);
    p -= lr*(g + mu*buf)."""

    def __init__(self, params, lr, momentum=0.9, weight_decay=0.0):
        self.params = [q for q in params if q.requires_grad]
        self.lr, self.mu, self.wd = lr, momentum, weight_decay
        self.buf = [None] * len(self.params)

    def step(self):
        with torch.no_grad():
            for i, q in enumerate(self.params):
                if q.grad is None:
                    continue
                g = q.grad + self.wd * q
                if self.buf[i] is None:
                    self.buf[i] = g.clone()
                else:
                    self.buf[i].mul_(self.mu).add_(g)
                q.add_(g + self.mu * self.buf[i], alpha=-self.lr)

    def zero_grad(self):
        for q in self.params:
            q.grad = None


class Lookahead:
    """models/optimiser/RAdam/lookahead.py:81-106 (pullback_momentum='none'): every call steps
    the inner optimizer (with whatever .grad is present -- the reference calls it once per epoch
    with the last batch's stale gradients, pretrain_BreastPathQ.py:293); every ``la_steps``-th
    call: p = alpha*p + (1-alpha)*cached; cached = p."""

    def __init__(self, opt, la_steps=5, la_alpha=0.8):
        self.opt, self.k, self.alpha, self.n = opt, la_steps, la_alpha, 0
        self.cached = [q.detach().clone() for q in opt.params]

    def step(self):
        self.opt.step()
        self.n += 1
        if self.n >= self.k:
            self.n = 0
            with torch.no_grad():
                for q, c in zip(self.opt.params, self.cached):
                    q.mul_(self.alpha).add_(c, alpha=1.0 - self.alpha)
                    c.copy_(q)


class AverageMeter:
    """util.py:26-46."""

    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


# ------------------------------------------------------------------ step bodies
def ssl_cr_step(kind, ps, bs, pt, bt, opt, x, y, u_w, u_s, lambda_u, faithful=True):
    """One consistency-training iteration.

    kind='mse': eval_BreastPathQ_SSL_CR.py:65-100 ; kind='ce': eval_Camelyon_SSL_CR.py:94-121
    (= eval_Kather_SSL_CR.py).  ``x`` [Nx,3,H,W], ``u_w/u_s`` [Nu,3,H,W] float tensors in
    0..255 (the reference feeds raw uint8 -> .float(), no normalisation); ``y`` [Nx] float
    (mse) or int64 (ce).  ps/bs = student params/buffers (net + classifier merged), pt/bt =
    teacher.  Returns dict(loss, loss_x, loss_u, feats, logits_x, logits_u_s, logits_u_w[, acc]).
    """
    with torch.no_grad():                                   # teacher: eval + no_grad (:43-44,77-79)
        feat_u_w = M.finetune_forward(pt, bt, u_w, False, faithful)
        logits_u_w = M.classifier_forward(pt, feat_u_w)
    inputs = torch.cat((x, u_s))
    feats = M.finetune_forward(ps, bs, inputs, True, faithful)
    logits = M.classifier_forward(ps, feats)
    nx = x.shape[0]
    logits_x, logits_u_s = logits[:nx], logits[nx:]
    out = {}
    if kind == "mse":
        loss_x = F.mse_loss(logits_x, y.view(-1, 1), reduction="mean")
        loss_u = F.mse_loss(logits_u_w, logits_u_s, reduction="mean")
    else:
        loss_x = F.cross_entropy(logits_x, y, reduction="mean")
        targets_u = torch.softmax(logits_u_w, dim=-1).max(dim=-1)[1]   # no confidence threshold
        loss_u = F.cross_entropy(logits_u_s, targets_u, reduction="mean")
        out["acc"] = (logits_x.argmax(1) == y).sum().item() / nx
    loss = loss_x + lambda_u * loss_u
    if opt is not None:
        opt.zero_grad()
        loss.backward()
        opt.step()
    out.update(loss=loss.item(), loss_x=loss_x.item(), loss_u=loss_u.item(), feats=feats.detach(),
               logits_x=logits_x.detach(), logits_u_s=logits_u_s.detach(), logits_u_w=logits_u_w)
    return out


def rsp_step(p, b, opt, i1, i2, i3, target, train=True):
    """pretrain_BreastPathQ.py:42-72 (train) / :110-128 (validate): TripletNet -> Classifier ->
    CrossEntropyLoss -> SGD-Nesterov."""
    ctx = torch.enable_grad() if train else torch.no_grad()
    with ctx:
        feats = M.triplet_forward(p, b, i1, i2, i3, train)
        output = M.classifier_forward(p, feats)
        loss = F.cross_entropy(output, target)
        if train and opt is not None:
            opt.zero_grad()
            loss.backward()
            opt.step()
    acc = (output.argmax(1) == target).sum().item() / target.shape[0]
    return dict(loss=loss.item(), acc=acc, feats=feats.detach(), output=output.detach())


def supervised_step(kind, p, b, opt, x, y, faithful=True, train=True):
    """eval_Camelyon_SSL.py:52-98 / eval_BreastPathQ_SSL.py:52-84 / eval_Kather_SSL.py:51-79
    (student only), and every ``validate()`` body of the eval_* scripts when ``train=False``."""
    ctx = torch.enable_grad() if train else torch.no_grad()
    with ctx:
        feats = M.finetune_forward(p, b, x, train, faithful)
        logits = M.classifier_forward(p, feats)
        if kind == "mse":
            loss = F.mse_loss(logits, y.view(-1, 1), reduction="mean")
        else:
            loss = F.cross_entropy(logits, y, reduction="mean")
        if train and opt is not None:
            opt.zero_grad()
            loss.backward()
            opt.step()
    out = dict(loss=loss.item(), feats=feats.detach(), logits=logits.detach())
    if kind == "ce":
        out["acc"] = (logits.argmax(1) == y).sum().item() / x.shape[0]
    return out


def teacher_refresh(ps, bs, pt, bt, ema_decay=0.0):
    """eval_BreastPathQ_SSL_CR.py:515-516: teacher = deepcopy(student) each epoch, i.e. EMA with
    decay 0.  decay>0 is the north-star's EMA extension (buffers are always copied)."""
    with torch.no_grad():
        for k in pt:
            pt[k].mul_(ema_decay).add_(ps[k].detach(), alpha=1.0 - ema_decay)
        for k in bt:
            bt[k].copy_(bs[k])
