"""fp8 (OCP e4m3) forward conv path -- BASELINE config 5 (eval_Camelyon_SSL_CR.py:33-157 "fp8 MFMA conv path").

Kernel level: the HIP kernel (v_mfma_scale_f32_16x16x128_f8f6f4) against a CPU emulation of the SAME quantisation
(oracle/kernels_ref.py: activations and weights rounded to e4m3 exactly like the kernel does, fp32 convolution) -- that pins
layout, quantisation, dequantisation, epilogue and statistics; the tolerance is the bf16 rounding of the stored output.
Engine level: the drop-in Camelyon SSL_CR train() in the fp8 engine mode against the REFERENCE goldens with the measured,
stated fp8 error (north_star: "tolerance measured, not 1e-3")."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cases as C  # noqa: E402
from oracle import kernels_ref as R  # noqa: E402

from _util import held, load_golden, rel_err  # noqa: E402
from test_kernels_gpu import DEV, close, rnd  # noqa: E402

FP8_CASES = [(2, 16, 16, 128, 128), (3, 32, 32, 128, 128), (2, 16, 16, 256, 256), (8, 8, 8, 512, 512), (4, 8, 8, 256, 128),
             (70, 32, 32, 128, 128),       # 280 tiles on 256 persistent workgroups: the tile walk + cross-tile halo prefetch
             (3, 16, 32, 256, 384)]


def _bf(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("case", FP8_CASES)
def test_pack_fp8_matches_the_oracle_quantiser(case):
    from ssl_cr_histo_amd import kernels as K
    _, _, _, Cc, Ko = case
    w = rnd(11, (Ko, Cc, 3, 3), 0.05)
    w[1] *= 37.0                                      # channels of very different magnitude: per-kout scales
    w[2] *= 1e-3
    w8, dq, _ = K.pack_conv_fp8(w.to(DEV))
    wq, dq_ref, _ = R.fp8_weight_pack(w)
    got = w8.view(torch.float8_e4m3fn).float().cpu().permute(0, 3, 1, 2)           # [K,3,3,C] -> [K,C,3,3]
    assert torch.equal(dq.cpu(), dq_ref)
    assert torch.equal(got, wq)
    assert float((got.abs().flatten(1).max(1).values).min()) > 224.0 - 1e-3
    bn = tuple(t.to(DEV) for t in (rnd(12, (Ko,)).abs() + 0.5, rnd(13, (Ko,)), rnd(14, (Ko,)), rnd(15, (Ko,)).abs() + 0.5))
    w8f, dqf, bias = K.pack_conv_fp8(w.to(DEV), bn=bn)
    wqf, dqf_ref, bias_ref = R.fp8_weight_pack(w, bn=tuple(t.cpu() for t in bn))
    assert torch.equal(dqf.cpu(), dqf_ref)
    gotf = w8f.view(torch.float8_e4m3fn).float().cpu().permute(0, 3, 1, 2)
    assert float((gotf != wqf).float().mean()) < 1e-4          # w * gamma / sqrt(var + eps): one fp32 ulp can flip an e4m3 rounding
    close(bias, bias_ref, 1e-6, "folded bias")


@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("case", FP8_CASES)
def test_conv_fp8_vs_quantised_oracle(case, mode):
    from ssl_cr_histo_amd import kernels as K
    N, H, W, Cc, Ko = case
    x = _bf(rnd(21, (N, H, W, Cc), 1.5) + 0.3)
    w = rnd(22, (Ko, Cc, 3, 3), 0.04)
    w8, dq, _ = K.pack_conv_fp8(w.to(DEV))
    wq, dq_ref, _ = R.fp8_weight_pack(w)
    xd = x.to(DEV).to(torch.bfloat16)
    if mode == "train":                               # producer BatchNorm + ReLU on the load path, raw output + statistics
        sc, sh = rnd(23, (Cc,)).abs() + 0.5, rnd(24, (Cc,), 0.5)
        y, stats = K.conv2d_fp8(xd, w8, dq, in_scale=sc.to(DEV), in_shift=sh.to(DEV), in_relu=True, want_stats=True)
        want, raw = R.conv3x3_fp8(x, wq, dq_ref, in_scale=sc, in_shift=sh, in_relu=True)
        close(y, want, 1.2e-2, "fp8 conv raw")
        s, ss = R.channel_stats(raw)
        st = stats.double().sum(0).cpu()
        close(st[0], s, 2e-3, "sum")
        close(st[1], ss, 2e-3, "sumsq")
    else:                                             # folded BatchNorm bias + residual + ReLU
        bias = rnd(25, (Ko,))
        res = _bf(rnd(26, (N, H, W, Ko)))
        y = K.conv2d_fp8(xd, w8, dq, bias=bias.to(DEV), residual=res.to(DEV).to(torch.bfloat16), relu=True)
        want, _ = R.conv3x3_fp8(x, wq, dq_ref, bias=bias, residual=res, relu=True)
        close(y, want, 1.2e-2, "fp8 conv eval")
    # ... and the quantisation error itself, against the un-quantised convolution of the same operands (reported, loosely bounded)
    full = R.conv_fwd(x if mode == "eval" else torch.relu(x * sc + sh), R.krsc(w), 1, 1)
    got_raw = (y.float().cpu() if mode == "train" else None)
    if got_raw is not None:
        e = float((got_raw - full).norm() / full.norm())
        print(f"fp8 vs fp32 conv, relative L2 error of the raw output: {e:.3e}")
        assert e < 6e-2


def test_conv_fp8_x_scale_and_saturation():
    """x_scale moves the activations inside the e4m3 window and is divided out again; values beyond +-448 / x_scale saturate
    (v_cvt_pk_fp8_f32 itself would produce NaN there)."""
    from ssl_cr_histo_amd import kernels as K
    N, H, W, Cc, Ko = 2, 16, 16, 128, 128
    x = _bf(rnd(31, (N, H, W, Cc), 0.01))
    x[0, 3, 3, :8] = 1000.0
    w = rnd(32, (Ko, Cc, 3, 3), 0.04)
    w8, dq, _ = K.pack_conv_fp8(w.to(DEV))
    wq, dq_ref, _ = R.fp8_weight_pack(w)
    for xs in (1.0, 16.0, 0.25):
        amax = torch.zeros(1, device=DEV)
        y = K.conv2d_fp8(x.to(DEV).to(torch.bfloat16), w8, dq, x_scale=xs, amax_out=amax)
        want, _ = R.conv3x3_fp8(x, wq, dq_ref, x_scale=xs)
        assert torch.isfinite(y.float()).all()
        close(y, want, 1.2e-2, f"x_scale {xs}")
        assert abs(float(amax) - 1000.0) <= 4.0, float(amax)        # the amax the delayed scaling feeds on: before scale and clamp (bf16: 1000)


def test_conv_fp8_rejects_unserved_shapes():
    from ssl_cr_histo_amd import kernels as K, _lib as L
    x = torch.zeros((2, 16, 16, 64), dtype=torch.bfloat16, device=DEV)
    w8 = torch.zeros((64, 3, 3, 64), dtype=torch.uint8, device=DEV)
    with pytest.raises(L.SslcrError):
        K.conv2d_fp8(x, w8, torch.ones(64, device=DEV))


# ------------------------------------------------------------------------------------------------ engine level
def _fp8_engine():
    from test_engine_gpu import _engine
    return _engine("fp8")


def test_fp8_camelyon_full_size_step_vs_reference():
    """One Camelyon SSL_CR iteration (eval_Camelyon_SSL_CR.train: CE + hard pseudo-label CE, SGD-Nesterov, the reference's three
    randperm shuffles) at student 640 / teacher 448 images of 256x256 in the fp8 engine mode -- e4m3 forward convs in layers 2-4
    of teacher and student, bf16 everything else and the whole backward -- against the REFERENCE's golden of that iteration.
    north_star: "fp8 ... tolerance measured, not 1e-3".  Measured on MI355X: losses within 2e-2 of the reference's, feature row
    norms within 0.2 (bf16 mode: 6e-2 / 0.2); asserted at those bounds."""
    from ssl_cr_histo_amd import steps
    from test_engine_gpu import build, freeze, ns
    _fp8_engine()
    name = "cam_cr_full"
    c = C.CASES[name]
    g = load_golden(name)
    mt, ct = build("finetune", "finetune", 2, True)
    ms, cs = build("finetune", "finetune", 2, True)
    freeze(mt, 64)
    freeze(ms, c["modules"])
    opt = torch.optim.SGD(filter(lambda p: p.requires_grad, list(ms.parameters()) + list(cs.parameters())), lr=c["lr"], momentum=0.9,
                          weight_decay=c["wd"], nesterov=True)
    torch.manual_seed(777)
    ret = steps.cam_cr_train(ns(lambda_u=c["lambda_u"], image_size=c["hw"]), mt, ms, ct, cs,
                             C.labeled_batches_cls(name, 1000, 1), C.labeled_batches_cls(name, 1100, 0),
                             C.unlabeled_batches(name, 2000), C.unlabeled_batches(name, 2100), opt, 1)
    dev = [abs(ret[i] - g[f"{name}/ret"][i]) / abs(g[f"{name}/ret"][i]) for i in range(3)]
    f = ret[4].cpu().double()
    e_row = rel_err(f.norm(dim=1), g[f"{name}/feats_rowl2"])
    print(f"[fp8] Camelyon full-size step vs the reference: loss deviations {dev}, accuracy {ret[3]} vs {g[f'{name}/ret'][3]}, "
          f"feature row-norm error {e_row:.3e}")
    # 2 x the deviations measured on the MI355X (tests/measured_errors.json), ceilings 6e-2 / 0.25
    for i, d_ in enumerate(dev):
        held(f"{name}/ret{i}/fp8", d_, 6e-2, floor=2e-3)
    held(f"{name}/feats_rowl2/fp8", e_row, 0.25, floor=2e-3)
    assert torch.equal(ret[5].cpu(), torch.from_numpy(g[f"{name}/targets"]))


def test_fp8_config5_per_gpu_shape_vs_parity_engine():
    """BASELINE config 5's per-GPU shape (--batch_size 1024 over 8 GPUs: two class loaders x 128 x 3 labeled = 768, 2 x 896 = 1792
    unlabeled -> student 2560 / teacher 1792 images of 256x256, 4352 distinct patches): ONE step in the fp8 engine mode against the
    same step in the fp32 exact-parity mode (which holds 1e-3 against the reference goldens at 640 / 448 images).  Reports and
    bounds the fp8 error on losses, logits and the classifier gradient at the size the configuration names."""
    from test_engine_gpu import build, freeze
    from ssl_cr_histo_amd import engine as E
    hw, b, mu = 256, 128, 7
    nx, nu = 2 * b * 3, 2 * b * mu
    x = C.u8(9100, (nx, 3, hw, hw)).to(DEV)
    u_w, u_s = C.u8(9101, (nu, 3, hw, hw)).to(DEV), C.u8(9102, (nu, 3, hw, hw)).to(DEV)
    y = C.ints(9103, (nx,), 2).to(DEV)
    out = {}
    for dtype in ("fp32", "fp8"):
        eng = E.Engine(DEV, dtype)
        mt, ct = build("finetune", "finetune", 2, True)
        ms, cs = build("finetune", "finetune", 2, True)
        OMs = {k: v.clone() for k, v in ms.state_dict().items()}
        freeze(mt, 64)
        mt.eval(); ms.train()
        te, st = eng.bind(mt, ct), eng.bind(ms, cs)
        if dtype == "fp8":         # delayed scaling: the first forward of a net runs at scale 1 and records amax; scales follow from the second on
            eng.step_ssl_cr(te, st, "ce", x, y, u_w, u_s, 1.0, backward=False)
            ms.load_state_dict(OMs)                    # undo the warm-up's running-statistics update: same starting state as the fp32 run
        r = eng.step_ssl_cr(te, st, "ce", x, y, u_w, u_s, 1.0)
        torch.cuda.synchronize()
        out[dtype] = dict(losses=r["losses"].cpu().double(), logits=r["logits"].cpu().double(), logits_t=r["logits_t"].cpu().double(),
                          gcls=st.grad(64).cpu().double(), gfc=st.grad(60).cpu().double())
        del eng, te, st, mt, ms
        torch.cuda.empty_cache()
    a, bq = out["fp32"], out["fp8"]
    dl = ((bq["losses"][:3] - a["losses"][:3]).abs() / a["losses"][:3].abs()).tolist()
    e_log = float((bq["logits"] - a["logits"]).norm() / a["logits"].norm())
    e_logt = float((bq["logits_t"] - a["logits_t"]).norm() / a["logits_t"].norm())
    e_g = float((bq["gcls"] - a["gcls"]).norm() / a["gcls"].norm())
    e_fc = float((bq["gfc"] - a["gfc"]).norm() / a["gfc"].norm())
    flips = int(((bq["logits_t"].argmax(1)) != (a["logits_t"].argmax(1))).sum())
    print(f"[fp8 vs fp32 engine, student {nx + nu} / teacher {nu}] loss deviations {dl}; logits rel L2 {e_log:.3e} (teacher {e_logt:.3e}); "
          f"pseudo-label flips {flips}/{nu}; classifier gradient rel L2 {e_g:.3e}, fc.0 gradient {e_fc:.3e}")
    # 2 x the deviations measured on the MI355X (tests/measured_errors.json); the constants are ceilings
    for i, d_ in enumerate(dl):
        held(f"config5/loss{i}/fp8_vs_fp32", d_, 6e-2, floor=1e-3)
    held("config5/logits/fp8_vs_fp32", e_log, 0.3, floor=1e-2)
    held("config5/logits_t/fp8_vs_fp32", e_logt, 0.3, floor=1e-2)
    held("config5/grad_classifier/fp8_vs_fp32", e_g, 0.5, floor=1e-2)
    held("config5/grad_fc0/fp8_vs_fp32", e_fc, 0.5, floor=1e-2)
    assert flips <= max(2, nu // 200), flips
