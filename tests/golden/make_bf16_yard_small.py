"""bf16-storage yardsticks of the SMALL-epoch cases (the reference's train() / validate() loops at small sizes: bpq_cr_f60/f0,
cam_cr_f60/f0, kather_cr_f0, rsp, cam_sup, bpq_sup, kather_sup), from the CPU oracle alone:
    python tests/golden/make_bf16_yard_small.py          -> tests/golden/bf16_yard_small.npz

Every case's epoch (oracle/epochs.py, the loops tests/test_oracle_golden.py pins to the reference's own numbers) is run twice on the
CPU from the same seeded state: as it is (fp32) and with every tensor the engine's bf16 mode stores rounded to bf16
(oracle/bf16_emul.py:emulating -- student in train mode, teacher / validate() with BatchNorm folded into bf16 filters).  The relative
distance between the two runs, per returned quantity, is what bf16 STORAGE alone does to that quantity over the epoch's iterations
and optimizer steps; tests/test_engine_gpu*.py hold the engine's bf16 mode to a small multiple of it (yard_small) instead of the
flat 6e-2 / 1e-1 constants the small-epoch keys sat under until round 6.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import bf16_emul as B  # noqa: E402
from oracle import cases as C  # noqa: E402
from oracle import epochs as E  # noqa: E402
from oracle import steps as S  # noqa: E402
from _util import merged, oracle_state  # noqa: E402


def relx(a, b):
    return abs(float(a) - float(b)) / (abs(float(b)) + 1e-30)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-300))


def student_teacher(classes, modules):
    pn_s, bn_s, pc_s = oracle_state("finetune", classes, True)
    pn_t, bn_t, pc_t = oracle_state("finetune", classes, True)
    S.apply_freeze(pn_s, modules)
    for v in pc_s.values():
        v.requires_grad_(True)
    return merged(pn_s, pc_s), bn_s, merged(pn_t, pc_t), bn_t


def single(kind_cls, classes, rand_stats=False):
    pn, bn, pc = oracle_state(kind_cls, classes, rand_stats)
    p = merged(pn, pc)
    for v in p.values():
        v.requires_grad_(True)
    return p, bn


def run_bpq_cr(name):
    c = C.CASES[name]
    ps, bs, pt, bt = student_teacher(1, c["modules"])
    opt = S.Adam(ps.values(), c["lr"], (0.9, 0.999), 1e-8, c["wd"])
    ret = E.bpq_cr_train(ps, bs, pt, bt, opt, C.labeled_batches(name), C.unlabeled_batches(name), c["lambda_u"], False)
    val = E.bpq_cr_validate(ps, bs, C.val_batches_reg(name), False)
    return {"ret0": ret[0], "ret1": ret[1], "ret2": ret[2], "feats": ret[3], "val": val}


def run_cam_cr(name):
    c = C.CASES[name]
    ps, bs, pt, bt = student_teacher(2, c["modules"])
    opt = S.SGDNesterov(ps.values(), c["lr"], 0.9, c["wd"])
    torch.manual_seed(777)
    ret = E.cam_cr_train(ps, bs, pt, bt, opt, C.labeled_batches_cls(name, 1000, 1), C.labeled_batches_cls(name, 1100, 0),
                         C.unlabeled_batches(name, 2000), C.unlabeled_batches(name, 2100), c["lambda_u"], c["hw"], False)
    torch.manual_seed(778)
    val = E.cam_cr_validate(ps, bs, C.val_batches_cls(name, 4000, 1), C.val_batches_cls(name, 4100, 0), False)
    return {"ret0": ret[0], "ret1": ret[1], "ret2": ret[2], "feats": ret[4], "val": val[0]}


def run_kather_cr(name):
    c = C.CASES[name]
    ps, bs, pt, bt = student_teacher(9, c["modules"])
    opt = S.Adam(ps.values(), c["lr"], (0.9, 0.999), 1e-8, c["wd"])
    ret = E.kather_cr_train(ps, bs, pt, bt, opt, C.labeled_batches_kather(name), C.unlabeled_batches(name), c["lambda_u"], False)
    val = E.kather_cr_validate(ps, bs, C.val_batches_kather(name), False)
    return {"ret0": ret[0], "ret1": ret[1], "ret2": ret[2], "val": val[0]}


def run_rsp(name):
    c = C.CASES[name]
    p, bn = single("mlp", 6)
    opt = S.SGDNesterov(p.values(), c["lr"], 0.9, c["wd"])
    ret = E.rsp_epoch(p, bn, opt, C.rsp_batches(name), c["hw"], True)
    val = E.rsp_epoch(p, bn, None, C.rsp_batches(name, 3500), c["hw"], False)
    return {"ret0": ret[0], "feats": ret[2], "val": val[0]}


def run_cam_sup(name):
    c = C.CASES[name]
    p, bn = single("finetune", 2)
    opt = S.SGDNesterov(p.values(), c["lr"], 0.9, c["wd"])
    torch.manual_seed(779)
    ret = E.cam_sup_train(p, bn, opt, C.labeled_batches_cls(name, 1000, 1), C.labeled_batches_cls(name, 1100, 0), c["hw"], False)
    return {"ret0": ret[0], "feats": ret[2]}


def run_bpq_sup(name):
    c = C.CASES[name]
    p, bn = single("finetune", 1)
    opt = S.Adam(p.values(), c["lr"], (0.9, 0.999), 1e-8, c["wd"])
    ret = E.bpq_sup_train(p, bn, opt, C.labeled_batches(name), c["hw"], False)
    return {"ret0": ret[0], "feats": ret[1]}


def run_kather_sup(name):
    c = C.CASES[name]
    p, bn = single("finetune", c["classes"])
    opt = S.Adam(p.values(), c["lr"], (0.9, 0.999), 1e-8, c["wd"])
    ret = E.kather_sup_train(p, bn, opt, C.sup_batches_kather(name), c["hw"], False)
    val = E.kather_sup_validate(p, bn, C.val_batches_kather(name), False)
    return {"ret0": ret[0], "val": val[0]}


RUNS = [("bpq_cr_f60", run_bpq_cr), ("bpq_cr_f0", run_bpq_cr), ("cam_cr_f60", run_cam_cr), ("cam_cr_f0", run_cam_cr),
        ("kather_cr_f0", run_kather_cr), ("rsp", run_rsp), ("cam_sup", run_cam_sup), ("bpq_sup", run_bpq_sup), ("kather_sup", run_kather_sup)]


def main():
    out = {}
    for name, fn in RUNS:
        exact = fn(name)
        with B.emulating():
            emu = fn(name)
        for k, v in exact.items():
            e = rel(emu[k], v) if torch.is_tensor(v) else relx(emu[k], v)
            out[f"{name}/{k}_err"] = np.array([e])
            if not torch.is_tensor(v):
                out[f"{name}/{k}_fp32"] = np.array([float(v)])
            print(f"{name}/{k}: bf16-storage emulation vs fp32 oracle {e:.3e}", flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bf16_yard_small.npz"), **out)


if __name__ == "__main__":
    main()
