#!/usr/bin/env python
"""full-size SSL_CR iteration: per-parameter gradient of the bf16 engine against the fp32 engine (same weights, same inputs):
relative L2 error, cosine, norm ratio.  python tools/grad_bf16_vs_fp32.py [case]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cases as C
from oracle import model as OM
from ssl_cr_histo_amd import engine as E, net

name = sys.argv[1] if len(sys.argv) > 1 else "bpq_cr_full"
c = C.CASES[name]
DEV = "cuda:0"

def build():
    m, cl = net.TripletNet_Finetune("resnet18"), net.FinetuneResNet(1)
    m.load_state_dict(OM.init_state(C.PARAM_SEED, OM.net_param_specs(), random_running_stats=True))
    cl.load_state_dict(OM.init_state(C.PARAM_SEED + 1, OM.classifier_param_specs("finetune", 1)))
    return m.to(DEV), cl.to(DEV)

(xl, yl), = C.labeled_batches(name)[:1]
(uw, us), = C.unlabeled_batches(name)[:1]
hw = c["hw"]
res = {}
for dt in ("fp32", "bf16"):
    E._engines.clear()
    E.set_engine(E.Engine(DEV, dt))
    eng = E.get_engine(DEV)
    mt, ct = build(); ms, cs = build()
    for p in mt.parameters(): p.requires_grad = False
    te, st = eng.bind(mt, ct), eng.bind(ms, cs)
    mt.eval(); ms.train()
    r = eng.step_ssl_cr(te, st, "mse", xl.reshape(-1, 3, hw, hw), yl.reshape(-1), uw, us, c["lambda_u"])
    names = [k for k, _ in list(ms.named_parameters()) + list(cs.named_parameters())]
    res[dt] = ([st.grad(i).cpu().double().reshape(-1) for i in range(len(names))], r["losses"].cpu(), r["logits"].cpu().double(), r["logits_t"].cpu().double())
print("losses fp32", res["fp32"][1].tolist(), "bf16", res["bf16"][1].tolist())
for j, w in ((2, "logits"), (3, "logits_t")):
    a, b = res["fp32"][j], res["bf16"][j]
    print(w, "rel err", float((a - b).norm() / a.norm()), "rms", float(a.pow(2).mean().sqrt()))
for i, k in enumerate(names):
    a, b = res["fp32"][0][i], res["bf16"][0][i]
    print(f"{i:2d} {k:40s} rel {float((a-b).norm()/a.norm()):.3f} cos {float((a*b).sum()/(a.norm()*b.norm())):.4f} ratio {float(b.norm()/a.norm()):.4f}")
