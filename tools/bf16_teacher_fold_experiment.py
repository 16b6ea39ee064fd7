"""Round-5 review item 6: is the bf16 mode's loss error on the full-size BreastPathQ iteration (1.9e-3 measured, 1.75e-3 emulated; north_star asks
1e-3) caused by FOLDING the teacher's BatchNorm into bf16 filters?  CPU emulation (oracle/bf16_emul.py conventions) of the teacher forward
(eval_BreastPathQ_SSL_CR.py:77-79) of bpq_cr_full in fp32 arithmetic with bf16 rounding at chosen places, against the float64 run:
  folded      w * gamma / sqrt(var + eps) rounded to bf16, shift fp32 in the epilogue, activations stored bf16   (what the engine does)
  unfolded    w rounded to bf16, scale AND shift fp32 in the epilogue, activations stored bf16                   (the review's proposal)
  acts_only   fp32 filters (no weight rounding at all), activations stored bf16
  weights_only  folded bf16 filters, activations kept fp32
and the loss of the iteration with each teacher next to the bf16-emulated and the exact student.
    python tools/bf16_teacher_fold_experiment.py            (about ten minutes of CPU)"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import bf16_emul as B  # noqa: E402
from oracle import cases as C  # noqa: E402
from oracle import model as M  # noqa: E402
import make_bf16_yard as Y  # noqa: E402


def backbone_eval_variant(p, b, x, variant, pre="model."):
    qa = (lambda t: t) if variant == "weights_only" else B.rnd          # activation storage
    def conv(x, cname, bname, stride, pad):
        s = p[pre + bname + ".weight"] / torch.sqrt(b[pre + bname + ".running_var"] + 1e-5)
        sh = p[pre + bname + ".bias"] - b[pre + bname + ".running_mean"] * s
        w = p[pre + cname + ".weight"]
        if variant == "unfolded":
            return F.conv2d(x, B.rnd(w), None, stride, pad) * s.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        wf = w * s.view(-1, 1, 1, 1)
        return F.conv2d(x, wf if variant == "acts_only" else B.rnd(wf), sh, stride, pad)
    x = F.max_pool2d(qa(F.relu(conv(x, "conv1", "bn1", 2, 3))), 3, 2, 1)
    for name, cin, cout, stride, ds in M.BLOCKS:
        o = qa(F.relu(conv(x, name + ".conv1", name + ".bn1", stride, 1)))
        i = qa(conv(x, name + ".downsample.0", name + ".downsample.1", stride, 0)) if ds else x
        x = qa(F.relu(conv(o, name + ".conv2", name + ".bn2", 1, 1) + i))
    return torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)


def teacher(p, b, u_w, variant):
    out = []
    with torch.no_grad():
        for i in range(0, u_w.shape[0], 64):
            e = backbone_eval_variant(p, b, u_w[i:i + 64], variant)
            f = M.fc_head(p, torch.cat((e, e), 1))
            out.append(M.classifier_forward(p, torch.cat((f, f, f), 1)))
    return torch.cat(out)


def main():
    name = "bpq_cr_full"
    c = C.CASES[name]
    torch.set_num_threads(os.cpu_count() or 8)
    ex = Y.forward(name, torch.float64, False)                # float64: teacher logits, student logits, losses
    em = Y.forward(name, torch.float32, True)                 # the committed emulation (folded teacher, bf16 student)
    p, b = Y.params(c["classes"], torch.float32)
    x, y, u_w, u_s = Y.batch(name)
    nx = x.shape[0]
    print(f"float64 losses (total, lx, lu): {ex['ret'].tolist()}")
    print(f"committed emulation          : {em['ret'].tolist()}  rel err {((em['ret'] - ex['ret']).abs() / ex['ret'].abs()).tolist()}")
    lx64 = F.mse_loss(ex["logits"][:nx], y.double().view(-1, 1))
    for variant in ("folded", "unfolded", "acts_only", "weights_only"):
        lt = teacher(p, b, u_w.float(), variant).double()
        e_lt = Y.rel(lt, ex["lt"])
        row = [f"teacher {variant:12s}: logits rel err {e_lt:.3e}"]
        for sname, lg in (("bf16 student", em["logits"]), ("exact student", ex["logits"])):
            lx = F.mse_loss(lg[:nx], y.double().view(-1, 1))
            lu = F.mse_loss(lt, lg[nx:])
            tot = lx + c["lambda_u"] * lu
            row.append(f"{sname}: loss err {abs(float(tot) - float(ex['ret'][0])) / abs(float(ex['ret'][0])):.3e} (lu err {abs(float(lu) - float(ex['ret'][2])) / abs(float(ex['ret'][2])):.3e})")
        print(" | ".join(row), flush=True)
    print(f"student logits rel err (bf16 emulation vs float64): {Y.rel(em['logits'], ex['logits']):.3e}")


if __name__ == "__main__":
    main()
