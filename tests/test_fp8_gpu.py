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


# ------------------------------------------------------------------------------------------------ config 5 at its own shape vs the oracle
_C5 = {}


def _config5_oracle():
    """ONE eval_Camelyon_SSL_CR.train iteration at BASELINE config 5's per-GPU shape (student 768 + 1792 = 2560, teacher 1792 images
    of 256x256) on the CPU oracle (oracle/steps.py:ssl_cr_step, one backbone pass per image).  The reference's own train() cannot
    produce this golden in the build container: TripletNet_Finetune runs the backbone three times on the 2560-image batch and
    autograd keeps ~30 MB per image-pass (~230 GB; the container has 64 GB) -- so the fixture is the oracle, which tests/
    test_oracle_golden.py pins to the reference's functions at every size that does fit, run here on the GPU box's host
    (3 TB, 256 cores; ~1-2 minutes at 32 threads).  Skipped when the host has less than 256 GB available."""
    if _C5:
        return _C5
    avail = 0
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            avail = int(line.split()[1]) // (1 << 20)
    if avail < 256:
        pytest.skip(f"config-5 oracle needs ~150 GB of host memory, {avail} GB available")
    import os
    from collections import OrderedDict
    from oracle import model as OM, steps as S
    hw, b, mu = 256, 128, 7
    nx, nu = 2 * b * 3, 2 * b * mu
    x, u_w, u_s = C.u8(9100, (nx, 3, hw, hw)), C.u8(9101, (nu, 3, hw, hw)), C.u8(9102, (nu, 3, hw, hw))
    y = C.ints(9103, (nx,), 2)

    def state():
        p, bufs = OM.split_state(OM.init_state(C.PARAM_SEED, OM.net_param_specs(), random_running_stats=True))
        pc, _ = OM.split_state(OM.init_state(C.PARAM_SEED + 1, OM.classifier_param_specs("finetune", 2)))
        p = OrderedDict(p)
        p.update(pc)
        return p, bufs
    ps, bs = state()
    pt, bt = state()
    for v in ps.values():
        v.requires_grad_(True)

    class KeepGrads:                       # an "optimizer" that only lets the step run its backward
        def zero_grad(self):
            pass

        def step(self):
            pass
    threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    try:
        o = S.ssl_cr_step("ce", ps, bs, pt, bt, KeepGrads(), x.float(), y, u_w.float(), u_s.float(), 1.0, faithful=False)
    finally:
        torch.set_num_threads(threads)
    names = list(ps.keys())
    _C5.update(x=x, y=y, u_w=u_w, u_s=u_s, loss=(o["loss"], o["loss_x"], o["loss_u"]), acc=o["acc"],
               logits=torch.cat((o["logits_x"], o["logits_u_s"])).double(), logits_t=o["logits_u_w"].double(),
               grads={k: ps[k].grad.double() for k in ("fc.0.weight", "classifier.0.weight", "model.layer4.1.conv2.weight",
                                                       "model.layer2.0.conv1.weight")},
               index={k: names.index(k) for k in names})
    return _C5


@pytest.mark.parametrize("dtype", ["bf16", "fp8"])
def test_config5_per_gpu_shape_vs_oracle(dtype):
    """BASELINE config 5 (eval_Camelyon_SSL_CR.py:94-121 at --batch_size 1024 over 8 GPUs) at its own per-GPU shape, in the bf16
    engine mode and in the fp8 mode (e4m3 forward convs of layers 2-4), against the CPU oracle's iteration on the same inputs:
    losses, student / teacher logits, pseudo-label flips, and the gradients of four parameters from the head down to layer2.
    Bounds = 2 x the deviations measured on the MI355X (tests/measured_errors.json); the constants are ceilings."""
    from test_engine_gpu import build, freeze
    from ssl_cr_histo_amd import engine as E
    g = _config5_oracle()
    x, y, u_w, u_s = (g[k].to(DEV) for k in ("x", "y", "u_w", "u_s"))
    eng = E.Engine(DEV, dtype)
    mt, ct = build("finetune", "finetune", 2, True)
    ms, cs = build("finetune", "finetune", 2, True)
    state0 = {k: v.clone() for k, v in ms.state_dict().items()}
    freeze(mt, 64)
    mt.eval(); ms.train()
    te, st = eng.bind(mt, ct), eng.bind(ms, cs)
    if dtype == "fp8":          # delayed scaling: the first forward calibrates; undo its running-statistics update
        eng.step_ssl_cr(te, st, "ce", x, y, u_w, u_s, 1.0, backward=False)
        ms.load_state_dict(state0)
    r = eng.step_ssl_cr(te, st, "ce", x, y, u_w, u_s, 1.0)
    torch.cuda.synchronize()
    got = r["losses"].cpu().double()
    tag = f"config5_oracle/{dtype}"
    for i, nm in enumerate(("loss", "loss_x", "loss_u")):
        held(f"{tag}/{nm}", abs(float(got[i]) - g["loss"][i]) / abs(g["loss"][i]), 6e-2, floor=1e-3)
    lg, lt = r["logits"].cpu().double(), r["logits_t"].cpu().double()
    held(f"{tag}/logits", float((lg - g["logits"]).norm() / g["logits"].norm()), 0.3, floor=1e-2)
    held(f"{tag}/logits_t", float((lt - g["logits_t"]).norm() / g["logits_t"].norm()), 0.3, floor=1e-2)
    flips = int((lt.argmax(1) != g["logits_t"].argmax(1)).sum())
    assert flips <= max(2, lt.shape[0] // 200), flips
    assert abs(float(got[3]) / x.shape[0] - g["acc"]) <= 3.0 / x.shape[0]
    for k, want in g["grads"].items():
        mine = st.grad(g["index"][k]).cpu().double()
        held(f"{tag}/grad/{k}", float((mine - want).norm() / want.norm()), 0.9, floor=2e-2)
    del eng, te, st
    torch.cuda.empty_cache()
