"""Run-to-run determinism of the engine: the same step on the same inputs gives the same BITS -- every parameter gradient, the
losses, the features and the BatchNorm running statistics.  Until round 4 the weight gradients of the fp32 (exact-parity) mode,
the BatchNorm-backward sums and the head biases ended in unordered floating-point atomics; a 24-iteration Adam trajectory then
landed on either side of its bound from box to box.  Now every cross-workgroup sum is a slab / row buffer folded in
a fixed order (csrc/wgrad_halo.hip:wgrad_fold_kernel, csrc/bn_eltwise.hip:bn_bwd_sums_kernel, csrc/heads.hip), in every dtype.
Reference behaviour this pins: eval_BreastPathQ_SSL_CR.py:65-100 and pretrain_BreastPathQ.py:42-61 are deterministic on CPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cases as C  # noqa: E402

from test_engine_gpu import _engine, build, freeze  # noqa: E402


def _ssl_cr_once(eng, name):
    c = C.CASES[name]
    mt, ct = build("finetune", "finetune", 1, True)
    ms, cs = build("finetune", "finetune", 1, True)
    freeze(mt, 64)
    freeze(ms, c["modules"])
    (xl, yl), = C.labeled_batches(name)
    (uw, us), = C.unlabeled_batches(name)
    te, st = eng.bind(mt, ct), eng.bind(ms, cs)
    mt.eval()
    ms.train()
    hw = c["hw"]
    r = eng.step_ssl_cr(te, st, "mse", xl.reshape(-1, 3, hw, hw), yl.reshape(-1), uw, us, c["lambda_u"])
    n = len(list(ms.parameters())) + len(list(cs.parameters()))
    out = {"losses": r["losses"].clone(), "feats": r["feats"].clone(), "logits": r["logits"].clone()}
    for i in range(n):
        out[f"grad{i}"] = st.grad(i)
    for k, v in ms.state_dict().items():
        if "running" in k:
            out[k] = v.detach().clone()
    torch.cuda.synchronize()
    return out


def _rsp_once(eng, name):
    c = C.CASES[name]
    model, cls = build("triplet", "mlp", 6, False)
    net_ = eng.bind(model, cls)
    model.train()
    cls.train()
    (i1, i2, i3, tgt), = C.rsp_batches(name)
    hw = c["hw"]
    r = eng.step_supervised(net_, "ce", [v.reshape(-1, 3, hw, hw) for v in (i1, i2, i3)], tgt.long().reshape(-1), train=True)
    n = len(list(model.parameters())) + len(list(cls.parameters()))
    out = {"losses": r["losses"].clone(), "feats": r["feats"].clone()}
    for i in range(n):
        out[f"grad{i}"] = net_.grad(i)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["bpq_cr_full", "rsp_full"])
def test_same_step_twice_gives_the_same_bits(name, dtype):
    """the full-size SSL_CR iteration (student 192 + 448, teacher 448 images; every conv / wgrad / BatchNorm-backward kernel of the
    headline step) and the full-size RSP iteration (3 x 128 images; in bf16 the segment forms), run three times from fresh modules:
    all outputs bit-identical."""
    eng = _engine(dtype)
    fn = _ssl_cr_once if name == "bpq_cr_full" else _rsp_once
    a = fn(eng, name)
    for rep in range(2):
        b = fn(eng, name)
        assert a.keys() == b.keys()
        diff = [k for k in a if not torch.equal(a[k], b[k])]
        assert not diff, f"{name}/{dtype} run {rep + 2}: {len(diff)} of {len(a)} outputs differ in their bits, first: {diff[:6]}"
