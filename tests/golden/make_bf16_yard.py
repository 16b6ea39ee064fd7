"""bf16-storage yardsticks of the full-size SSL_CR iterations, from the CPU oracle alone (no reference import needed):
    python tests/golden/make_bf16_yard.py          -> tests/golden/bf16_yard.npz

For bpq_cr_full / cam_cr_full / rsp_full (oracle/cases.py) and the forward-only config (fwd_full) the iteration's forward is run twice on the CPU, in float64 and in fp32 with
every tensor the engine's bf16 mode stores rounded to bf16 (oracle/bf16_emul.py) -- the student in train mode AND the
teacher in eval mode with BatchNorm folded into bf16 filters, which the `loss_bf16emul` field of the goldens leaves exact.
The relative distance between the two runs, per returned quantity, is what bf16 STORAGE alone does to that quantity; the
GPU test (tests/test_engine_gpu.py) holds the engine's bf16 mode to a small multiple of it instead of a flat constant.
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bf16_emul as B  # noqa: E402
from oracle import cases as C  # noqa: E402
from oracle import model as OM  # noqa: E402


def params(classes, dtype, kind_cls="finetune", rand_stats=True):
    sd = OM.init_state(C.PARAM_SEED, OM.net_param_specs(), random_running_stats=rand_stats)
    csd = OM.init_state(C.PARAM_SEED + 1, OM.classifier_param_specs(kind_cls, classes))
    p_net, b_net = OM.split_state(sd)
    p_cls, _ = OM.split_state(csd)
    p = OrderedDict((k, v.to(dtype)) for k, v in list(p_net.items()) + list(p_cls.items()))
    b = OrderedDict((k, v.to(dtype) if v.is_floating_point() else v) for k, v in b_net.items())
    return p, b


def batch(name):
    c = C.CASES[name]
    hw = c["hw"]
    if c["script"] == "bpq_cr":
        (x, y), = C.labeled_batches(name)
        (u_w, u_s), = C.unlabeled_batches(name)
        return x.reshape(-1, 3, hw, hw), y.reshape(-1), u_w, u_s
    (tx, ty), = C.labeled_batches_cls(name, 1000, 1)
    (nx, ny), = C.labeled_batches_cls(name, 1100, 0)
    (tuw, tus), = C.unlabeled_batches(name, 2000)
    (nuw, nus), = C.unlabeled_batches(name, 2100)
    tx, nx = tx.reshape(-1, 3, hw, hw), nx.reshape(-1, 3, hw, hw)
    torch.manual_seed(777)          # eval_Camelyon_SSL_CR.py:94-110 shuffles the concatenated halves; same seed as make_golden.py
    p_x, p_uw, p_us = torch.randperm(2 * len(tx)), torch.randperm(2 * len(tuw)), torch.randperm(2 * len(tus))
    return (torch.cat([tx, nx])[p_x], torch.cat([ty.reshape(-1), ny.reshape(-1)])[p_x], torch.cat([tuw, nuw])[p_uw],
            torch.cat([tus, nus])[p_us])


def forward(name, dtype, emulate):
    c = C.CASES[name]
    kind = "mse" if c["script"] == "bpq_cr" else "ce"
    p, b = params(c["classes"], dtype)
    x, y, u_w, u_s = batch(name)
    q = B.rnd if emulate else (lambda t: t)
    with torch.no_grad():
        lt = B.teacher_logits(p, b, u_w.to(dtype), emulate)
        e = B.backbone_train(p, torch.cat([x, u_s]).to(dtype), q)
        f = OM.fc_head(p, torch.cat((e, e), 1))
        feats = torch.cat((f, f, f), 1)
        logits = OM.classifier_forward(p, feats)
        nx = x.shape[0]
        if kind == "mse":
            lx, lu = F.mse_loss(logits[:nx], y.to(dtype).view(-1, 1)), F.mse_loss(lt, logits[nx:])
        else:
            lx = F.cross_entropy(logits[:nx], y)
            lu = F.cross_entropy(logits[nx:], torch.softmax(lt, -1).max(-1)[1])
    return dict(ret=torch.stack([lx + c["lambda_u"] * lu, lx, lu]).double(), lt=lt.double(), logits=logits.double(),
                feats=feats.double())


def forward_rsp(dtype, emulate):
    """pretrain_BreastPathQ.py:42-61 at its default size (rsp_full): three train-mode backbone passes, pairwise fc, 6-way CE"""
    p, _ = params(6, dtype, "mlp", False)
    (i1, i2, i3, t), = C.rsp_batches("rsp_full")
    q = B.rnd if emulate else (lambda t: t)
    with torch.no_grad():
        e1, e2, e3 = (B.backbone_train(p, i.to(dtype), q) for i in (i1, i2, i3))
        feats = torch.cat((OM.fc_head(p, torch.cat((e1, e2), 1)), OM.fc_head(p, torch.cat((e2, e3), 1)), OM.fc_head(p, torch.cat((e1, e3), 1))), 1)
        logits = OM.classifier_forward(p, feats)
        loss = F.cross_entropy(logits, t.long().reshape(-1))
    return dict(ret=torch.stack([loss]).double(), logits=logits.double(), feats=feats.double())


def forward_only(mode, dtype, emulate):
    """BASELINE config 2 (fwd_full): TripletNet_Finetune.forward on 256 images, eval (BatchNorm folded: the teacher's path) / train"""
    p, b = params(1, dtype)
    x = C.u8(5100, (256, 3, 256, 256)).to(dtype)
    q = B.rnd if emulate else (lambda t: t)
    with torch.no_grad():
        if mode == "eval":
            e = torch.cat([B.backbone_eval(p, b, x[i:i + 64], q) for i in range(0, 256, 64)])
        else:
            e = B.backbone_train(p, x, q)
        f = OM.fc_head(p, torch.cat((e, e), 1))
    return dict(feats=torch.cat((f, f, f), 1).double())


def feat_errs(out, key, em, ex):
    out[f"{key}/feats_err"] = np.array([rel(em["feats"], ex["feats"])])
    out[f"{key}/feats_rowl2_err"] = np.array([rel(em["feats"].norm(dim=1), ex["feats"].norm(dim=1))])
    out[f"{key}/feats_colsum_err"] = np.array([rel(em["feats"].sum(0), ex["feats"].sum(0))])
    out[f"{key}/feats_rowl2_f64"] = ex["feats"].norm(dim=1).numpy()


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-300))


def main():
    out = {}
    for name in ("bpq_cr_full", "cam_cr_full"):
        ex, em = forward(name, torch.float64, False), forward(name, torch.float32, True)
        out[f"{name}/ret_f64"] = ex["ret"].numpy()
        out[f"{name}/ret_bf16emul"] = em["ret"].numpy()
        out[f"{name}/ret_err"] = ((em["ret"] - ex["ret"]).abs() / ex["ret"].abs()).numpy()
        out[f"{name}/teacher_logits_err"] = np.array([rel(em["lt"], ex["lt"])])
        out[f"{name}/student_logits_err"] = np.array([rel(em["logits"], ex["logits"])])
        out[f"{name}/feats_err"] = np.array([rel(em["feats"], ex["feats"])])
        out[f"{name}/feats_rowl2_err"] = np.array([rel(em["feats"].norm(dim=1), ex["feats"].norm(dim=1))])
        out[f"{name}/feats_colsum_err"] = np.array([rel(em["feats"].sum(0), ex["feats"].sum(0))])
        out[f"{name}/argmax_flips"] = np.array([int((em["lt"].argmax(-1) != ex["lt"].argmax(-1)).sum())])
        for k in sorted(out):
            if k.startswith(name):
                print(k, out[k])
    ex, em = forward_rsp(torch.float64, False), forward_rsp(torch.float32, True)
    out["rsp_full/ret_f64"] = ex["ret"].numpy()
    out["rsp_full/ret_bf16emul"] = em["ret"].numpy()
    out["rsp_full/ret_err"] = ((em["ret"] - ex["ret"]).abs() / ex["ret"].abs()).numpy()
    out["rsp_full/student_logits_err"] = np.array([rel(em["logits"], ex["logits"])])
    feat_errs(out, "rsp_full", em, ex)
    for mode in ("eval", "train"):
        feat_errs(out, f"fwd_full/{mode}", forward_only(mode, torch.float32, True), forward_only(mode, torch.float64, False))
    for k in sorted(out):
        if k.startswith(("rsp_full", "fwd_full")) and out[k].size < 8:
            print(k, out[k])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bf16_yard.npz"), **out)


if __name__ == "__main__":
    main()
