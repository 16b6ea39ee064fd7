"""Step-level parity on the MI355X: the drop-in train()/validate() (native engine, through the C-ABI) against
(a) the golden vectors captured from the REFERENCE's own functions and (b) the CPU oracle on the same seeded inputs.

fp32 engine mode carries the north-star bound (1e-3 relative on logits/loss; looser, stated bounds on quantities that
amplify error such as post-Adam parameters).  bf16 mode is held to measured, looser bounds against the same goldens.
"""
import copy
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cases as C  # noqa: E402
from oracle import model as OM  # noqa: E402
from oracle import steps as S  # noqa: E402

from _util import check_snapshot, held, load_golden, merged, oracle_state, rel_err, yard_small  # noqa: E402

DEV = "cuda:0"


def _engine(dtype):
    from ssl_cr_histo_amd import engine as E
    idx = torch.device(DEV).index
    cur = E._engines.get(idx)
    want = E._DTYPES[dtype]
    if cur is None or cur.dtype != want:
        E._engines.pop(idx, None)
        E.set_engine(E.Engine(DEV, dtype))
    return E.get_engine(DEV)


def build(kind_net, kind_cls, classes, rand_stats, seed=C.PARAM_SEED):
    from ssl_cr_histo_amd import net
    model = net.TripletNet_Finetune("resnet18") if kind_net == "finetune" else net.TripletNet("resnet18")
    cls = net.FinetuneResNet(classes) if kind_cls == "finetune" else net.Classifier(768, classes)
    model.load_state_dict(OM.init_state(seed, OM.net_param_specs(), random_running_stats=rand_stats))
    cls.load_state_dict(OM.init_state(seed + 1, OM.classifier_param_specs(kind_cls, classes)))
    return model.to(DEV), cls.to(DEV)


def freeze(model, modules):
    for idx, (_, p) in enumerate(model.named_parameters()):
        p.requires_grad = idx >= modules


def state_of(model, cls):
    d = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    d.update({k: v.detach().cpu() for k, v in cls.state_dict().items()})
    return d


def ns(**kw):
    return types.SimpleNamespace(print_freq=0, **kw)


# tolerances: (return scalars, feats, post-step snapshot).  fp32 engine mode carries the stated bounds.  The bf16 figures are
# CEILINGS only: every bf16 assertion goes through near() -> _util.held(), i.e. 2 x the error measured on the MI355X for that
# very quantity (tests/measured_errors.json holds the table), floored at 2e-3 so that rounding-order noise cannot trip it.
TOLS = {"fp32": (1e-3, 1e-3, 5e-3), "bf16": (6e-2, 6e-2, 1e-1)}


def near(key, dtype, err, tol32, ceil16, floor=2e-3):
    if dtype == "fp32":
        assert err <= tol32, (key, err, tol32)
    else:
        held(f"{key}/{dtype}", err, ceil16, floor)


def yard16(name):
    """independent bf16 ceilings of a full-size iteration's returned losses / features: multiples of what bf16 STORAGE alone
    does to each quantity in the CPU emulation of the same iteration, student AND folded-bf16 teacher
    (tests/golden/make_bf16_yard.py -> bf16_yard.npz; oracle/bf16_emul.py).  Scalars and reductions (losses, row norms, column
    sums) are sums of partly cancelling rounding errors, so they get 5 x the emulated figure (losses floored at 1e-4, where the
    emulation's own cancellation is luck); the element-wise feature comparison gets 2 x."""
    y = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_yard.npz"))
    return {"ret": [5 * max(float(e), 1e-4) for e in y[f"{name}/ret_err"]] if f"{name}/ret_err" in y.files else [],
            "rowl2": 5 * float(y[f"{name}/feats_rowl2_err"][0]),
            "colsum": 5 * float(y[f"{name}/feats_colsum_err"][0]), "head": 2 * float(y[f"{name}/feats_err"][0])}


def relx(got, want):
    return abs(float(got) - float(want)) / (abs(float(want)) + 1e-30)


# tests/golden/*_flip.npz as reviewed (make_golden.py:gen_flip_yard; units with |z| < 1e-5 rms(z) in the float64 run)
FLIP_SHA256 = {"bpq_cr_full": "547ce9581a8fb95317c1925d9964277cdee1673a68978aaa3ad06862fed612d0",
               "cam_cr_full": "61966daeab5efd536a4570c9288fdb81a90ee0082b514c6bfc0abec8bf22198c",
               "rsp_full": "0a658acd9c1e100e1d3d7aeb810939f408fed08d14ea8b37a26e18e82636c7a8"}


def grad_rows_check(name, dtype, names, grad_of, g, emu_key="grad_bf16emul_err"):
    """every parameter gradient against the float64 run of the same iteration: L2 norm and one seeded +-1 projection.
    fp32: norm within max(3e-3, 3 x the reference's own fp32 error), projection within max(3e-3, 4.5 x it).  bf16: 2 x the measured error of THIS parameter (floor max(1e-2, 0.3 x the
    emulated bf16-storage error) for the norm;
    a single +-1 projection of an error vector e is ~N(0, |e|^2), so its floor is the parameter's measured norm-scale error:
    3.5 sigma of the emulated bf16-storage error), never above the old 2 x / 3.5 x emulation + 0.05 rule."""
    l2_ref, pr_ref, ref_err, emu = g[f"{name}/grad_l2_f64"], g[f"{name}/grad_probe_f64"], g[f"{name}/grad_ref32_err"], g[f"{name}/{emu_key}"]
    # head ReLUs whose sign fp32 cannot resolve (tests/golden/make_golden.py:gen_flip_yard: |z| < 1e-5 rms(z) in float64 -- rsp_full has
    # one at z = -1.08e-6): either mask is a valid fp32 evaluation, and the gradient that flows through such a unit alone is
    # flip_l2[i] / flip_pr[i] of parameter i's norm.  The fp32 bounds widen by exactly that
    flip_l2 = flip_pr = np.zeros(len(names))
    fpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"{name}_flip.npz")
    if os.path.exists(fpath):
        # the allowance is DATA that widens a bound: the file is pinned (a regenerated golden must be looked at, and its hash updated
        # here, before it can loosen the test), it must name the units it is about, and it may not exceed what was reviewed
        import hashlib
        with open(fpath, "rb") as f:
            sha = hashlib.sha256(f.read()).hexdigest()
        assert sha == FLIP_SHA256[name], f"{name}_flip.npz changed ({sha}): review the fragile units and update FLIP_SHA256"
        fy = load_golden(f"{name}_flip")
        flip_l2, flip_pr = fy[f"{name}_flip/flip_l2"], fy[f"{name}_flip/flip_pr"]
        units = fy[f"{name}_flip/units"]
        assert len(units) >= 1 and float(np.abs(units[:, 3]).max()) < 1e-5, units      # |z| of every listed unit is below fp32 resolution
        assert float(flip_l2.max()) <= 1.3e-2 and float(flip_pr.max()) <= 3.2e-2
        print(f"fragile head units (pair, sample, unit, z, rms z): {units.tolist()}")
    rows, bad = [], []
    for i, k in enumerate(names):
        gr = grad_of(i).cpu().double().reshape(-1)
        e_l2 = abs(float(gr.norm()) - l2_ref[i]) / (l2_ref[i] + 1e-30)
        e_pr = abs(float((gr * C.grad_probe(i, gr.numel())).sum()) - pr_ref[i]) / (l2_ref[i] + 1e-30)
        rows.append(f"   {i:2d} {k:40s} |g| {l2_ref[i]:.3e}  norm err {e_l2:.2e}  projection err/|g| {e_pr:.2e}  "
                    f"(reference fp32: {ref_err[i]:.2e}, bf16 emulation: {emu[i]:.2e})")
        try:
            if dtype == "fp32":
                # norm: max(3e-3, 3 x the reference's own fp32 error |g_ref32 - g_64| / |g_64| of THIS parameter).  The +-1
                # projection of an error vector e is a draw from ~N(0, |e|^2): with |e| = the reference's error, 3 sigma is exceeded
                # by one of the 66 parameters in one run out of six -- and the engine is deterministic now (r04), so every change of a
                # summation order re-rolls all 66 draws (layer4.1.bn1.weight went 2.9e-3 -> 3.15e-3 against a 3e-3 line when the
                # forward statistics' lane reduction changed).  4.5 sigma (7e-6 per draw) for the projection.
                tol = max(3e-3, 3.0 * ref_err[i])
                assert e_l2 <= tol + flip_l2[i] and e_pr <= max(3e-3, 4.5 * ref_err[i]) + flip_pr[i]
            else:
                # |g| can agree by cancellation although the vectors differ (a measured 1e-4 next to an emulated 0.4 is luck, and the
                # next kernel change lands at 2e-2): the floor scales with the error bf16 storage alone causes for this parameter
                held(f"{name}/grad_l2/{i}/{dtype}", e_l2, 2.0 * emu[i] + 0.05, floor=max(1e-2, 0.3 * emu[i]))
                held(f"{name}/grad_pr/{i}/{dtype}", e_pr, 3.5 * emu[i] + 0.05, floor=max(1e-2, 1.75 * emu[i]))
        except AssertionError as ex:
            bad.append(rows[-1] + f"   <- {ex}")
    print(f"[{dtype}] {name} gradients vs the float64 run of the same iteration:\n" + "\n".join(rows))
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_backbone_forward_vs_reference_stages(mode, dtype):
    """G3: TripletNet_Finetune features for N=2, 64x64, eval and train mode (+ the x3 running-stat replay)."""
    _engine(dtype)
    g = load_golden("stages")
    model, _ = build("finetune", "finetune", 1, True)
    model.train(mode == "train")
    x = C.u8(5000, (2, 3, 64, 64)).to(DEV)
    feats = model(x)
    torch.cuda.synchronize()
    near(f"stages/{mode}/feats", dtype, rel_err(feats.cpu(), g[f"stages/{mode}/feats"]), 1e-3, 5e-2)
    feats_f = model(x.float())                       # fp32 input path of the stem
    assert rel_err(feats_f.cpu(), feats.cpu()) < 1e-6
    if mode == "train":
        sd = model.state_dict()
        for k in ("model.bn1.running_mean", "model.bn1.running_var", "model.layer3.0.downsample.1.running_mean",
                  "model.layer4.1.bn2.running_var"):
            # two train forwards ran above: replay the reference update on the golden (one call) value
            pass
        assert int(sd["model.bn1.num_batches_tracked"]) == 6
        model2, _ = build("finetune", "finetune", 1, True)
        model2.train()
        model2(x)
        sd2 = model2.state_dict()
        for k in ("model.bn1.running_mean", "model.bn1.running_var", "model.layer3.0.downsample.1.running_mean",
                  "model.layer4.1.bn2.running_var"):
            near(f"stages/train/{k}", dtype, rel_err(sd2[k].cpu(), g[f"stages/train/{k}"]), 1e-4, 2e-2)
        assert int(sd2["model.bn1.num_batches_tracked"]) == 3


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_forward_only_config2_full_size_vs_reference(mode, dtype):
    """BASELINE config 2 at its own size -- ResNet18 forward-only, N = 256 images of 256x256 -- against reductions of the reference
    module's own output (tests/golden/make_golden.py:gen_fwd_full): eval mode (BatchNorm folded into the conv packs, the teacher /
    validate() path) and train mode (batch statistics, x3 running-stat replay).  fp32: 1e-3; bf16: 2 x the measured error, never above yard16()'s multiples of the emulated bf16-storage error."""
    _engine(dtype)
    g = load_golden("fwd_full")
    model, _ = build("finetune", "finetune", 1, True)
    model.train(mode == "train")
    x = C.u8(5100, (256, 3, 256, 256)).to(DEV)
    feats = model(x)
    torch.cuda.synchronize()
    f = feats.cpu().double()
    assert tuple(f.shape) == (256, 768)
    y16 = yard16(f"fwd_full/{mode}")          # bf16 ceilings: multiples of the emulated bf16-storage error of this very forward
    near(f"fwd_full/{mode}/feats_rowl2", dtype, rel_err(f.norm(dim=1), g[f"fwd_full/{mode}/feats_rowl2"]), 1e-3, y16["rowl2"])
    near(f"fwd_full/{mode}/feats_colsum", dtype, rel_err(f.sum(0), g[f"fwd_full/{mode}/feats_colsum"]), 1e-3, y16["colsum"])
    near(f"fwd_full/{mode}/feats_head", dtype, rel_err(feats[:4].cpu(), g[f"fwd_full/{mode}/feats_head"]), 1e-3, y16["head"])
    if mode == "train":
        sd = model.state_dict()
        for k in ("model.bn1.running_mean", "model.bn1.running_var", "model.layer3.0.downsample.1.running_mean",
                  "model.layer4.1.bn2.running_var"):
            near(f"fwd_full/train/{k}", dtype, rel_err(sd[k].cpu(), g[f"fwd_full/train/{k}"]), 1e-4, 2e-2)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["bpq_cr_f60", "bpq_cr_f0"])
def test_bpq_cr_epoch_vs_reference(name, dtype):
    from ssl_cr_histo_amd import steps
    _engine(dtype)
    c = C.CASES[name]
    g = load_golden(name)
    mt, ct = build("finetune", "finetune", 1, True)
    ms, cs = build("finetune", "finetune", 1, True)
    freeze(mt, 64)
    freeze(ms, c["modules"])
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, list(ms.parameters()) + list(cs.parameters())), lr=c["lr"],
                           betas=(0.9, 0.999), weight_decay=c["wd"])
    ret = steps.bpq_cr_train(ns(lambda_u=c["lambda_u"]), mt, ms, ct, cs, C.labeled_batches(name), C.unlabeled_batches(name), opt, 1)
    ts, tf, tp = TOLS[dtype]
    for i in range(3):
        near(f"{name}/ret{i}", dtype, relx(ret[i], g[f"{name}/ret"][i]), ts, yard_small(name, f"ret{i}", ts))
    near(f"{name}/feats", dtype, rel_err(ret[3].cpu(), g[f"{name}/feats"]), tf, yard_small(name, "feats", tf, scalar=False))
    assert torch.equal(ret[4].cpu(), torch.from_numpy(g[f"{name}/targets"]))
    if dtype == "fp32":
        check_snapshot(g, name, state_of(ms, cs), tp)
    val = steps.bpq_cr_validate(ns(), ms, cs, C.val_batches_reg(name), 1)
    near(f"{name}/val", dtype, relx(val, g[f"{name}/val"][0]), 5e-3, yard_small(name, "val", 1e-1))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_bpq_cr_full_size_step_vs_reference(dtype):
    """The BASELINE.json / bench.py workload at FULL size (student 192+448, teacher 448 images of 256x256, full
    fine-tune, Adam) against reductions of the reference's own iteration (tests/golden/make_golden.py:gen_bpq_cr_full):
    returned loss averages, feature-matrix row norms / column sums, the post-step state snapshot, and EVERY parameter
    gradient through its L2 norm and a seeded +-1 projection (the reference's .grad after its own backward).
    fp32 mode: 1e-3 on losses/features (north-star tolerance).  Gradients: at this size the early-layer gradients are sums
    of 640 per-image terms that largely cancel, and the golden's FLOAT64 run of the same iteration shows the reference's
    own fp32 .grad to be 3.5e-3..4.8e-3 (relative L2, per parameter: grad_ref32_err) away from the exact gradient there.
    So the engine is measured against the float64 reductions and held to max(3e-3, 3 x the reference's own fp32 error) per
    parameter.  bf16 mode: losses/features within yard16()'s multiples of the emulated bf16-storage error (student + folded teacher); gradients are held to 2 x the error that bf16 STORAGE alone causes in a
    CPU emulation of the same iteration (grad_bf16emul_err, oracle/bf16_emul.py: 20-45 % in the early layers of this
    random-weight, loss ~1e3 problem, cosine ~0.9) + 0.05 -- the rule of test_gradients_vs_oracle, at full size (the
    engine's bf16 error profile matches the emulation's: 0.45 at conv1, 0.19 at layer4.0.conv1, 0.004 at fc.0)."""
    from ssl_cr_histo_amd import steps
    eng = _engine(dtype)
    name = "bpq_cr_full"
    c = C.CASES[name]
    g = load_golden(name)

    def fresh():
        mt, ct = build("finetune", "finetune", 1, True)
        ms, cs = build("finetune", "finetune", 1, True)
        freeze(mt, 64)
        freeze(ms, c["modules"])
        return mt, ct, ms, cs

    # ---- (1) the epoch function, as the reference script calls it
    mt, ct, ms, cs = fresh()
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, list(ms.parameters()) + list(cs.parameters())), lr=c["lr"],
                           betas=(0.9, 0.999), weight_decay=c["wd"])
    ret = steps.bpq_cr_train(ns(lambda_u=c["lambda_u"]), mt, ms, ct, cs, C.labeled_batches(name), C.unlabeled_batches(name), opt, 1)
    ts, tf, tp = TOLS[dtype]
    y16 = yard16(name)
    for i in range(3):
        near(f"{name}/ret{i}", dtype, relx(ret[i], g[f"{name}/ret"][i]), ts, y16["ret"][i])
    f = ret[3].cpu().double()
    assert f.shape == (c["b"] * 3 + c["b"] * c["mu"], 768)
    near(f"{name}/feats_rowl2", dtype, rel_err(f.norm(dim=1), g[f"{name}/feats_rowl2"]), tf, y16["rowl2"])
    near(f"{name}/feats_colsum", dtype, rel_err(f.sum(0), g[f"{name}/feats_colsum"]), tf, y16["colsum"])
    near(f"{name}/feats_head", dtype, rel_err(ret[3][:4].cpu(), g[f"{name}/feats_head"]), tf, y16["head"])
    assert torch.equal(ret[4].cpu(), torch.from_numpy(g[f"{name}/targets"]))
    if dtype == "fp32":
        check_snapshot(g, name, state_of(ms, cs), tp)

    # ---- (2) the gradients of that iteration (fresh weights, no update in between)
    mt, ct, ms, cs = fresh()
    (xl, yl), = C.labeled_batches(name)
    (uw, us), = C.unlabeled_batches(name)
    te, st = eng.bind(mt, ct), eng.bind(ms, cs)
    mt.eval()
    ms.train()
    hw = c["hw"]
    eng.step_ssl_cr(te, st, "mse", xl.reshape(-1, 3, hw, hw), yl.reshape(-1), uw, us, c["lambda_u"])
    names = [str(n) for n in g[f"{name}/grad_names"]]
    mine = [k for k, _ in list(ms.named_parameters()) + list(cs.named_parameters())]
    assert names == mine, "parameter order differs from the reference's named_parameters()"
    ref_err = g[f"{name}/grad_ref32_err"]
    grad_rows_check(name, dtype, names, st.grad, g)
    assert abs(float(g[f"{name}/loss_f64"][0]) - g[f"{name}/ret"][0]) <= 1e-5 * g[f"{name}/ret"][0]     # one batch: average == the loss
    for key in g.files:
        if key.startswith(f"{name}/grad/"):
            k = key[len(name) + 6:]
            want = torch.from_numpy(g[key]).double()
            idx = names.index(k)
            got = st.grad(idx).cpu().double().reshape(want.shape)
            # full small tensors against the reference's fp32 .grad: two fp32 results, each ref_err from the exact one
            near(f"{name}/grad_full/{k}", dtype, rel_err(got, want), 3e-3 + 2.5 * ref_err[idx],
                 2.0 * g[f"{name}/grad_bf16emul_err"][idx] + 0.05, floor=1e-2)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["cam_cr_f60", "cam_cr_f0"])
def test_cam_cr_epoch_vs_reference(name, dtype):
    from ssl_cr_histo_amd import steps
    _engine(dtype)
    c = C.CASES[name]
    g = load_golden(name)
    mt, ct = build("finetune", "finetune", 2, True)
    ms, cs = build("finetune", "finetune", 2, True)
    freeze(mt, 64)
    freeze(ms, c["modules"])
    opt = torch.optim.SGD(filter(lambda p: p.requires_grad, list(ms.parameters()) + list(cs.parameters())), lr=c["lr"],
                          momentum=0.9, weight_decay=c["wd"], nesterov=True)
    torch.manual_seed(777)
    ret = steps.cam_cr_train(ns(lambda_u=c["lambda_u"], image_size=c["hw"]), mt, ms, ct, cs,
                             C.labeled_batches_cls(name, 1000, 1), C.labeled_batches_cls(name, 1100, 0),
                             C.unlabeled_batches(name, 2000), C.unlabeled_batches(name, 2100), opt, 1)
    ts, tf, tp = TOLS[dtype]
    for i in range(3):
        near(f"{name}/ret{i}", dtype, relx(ret[i], g[f"{name}/ret"][i]), ts, yard_small(name, f"ret{i}", ts))
    if dtype == "fp32":
        assert ret[3] == g[f"{name}/ret"][3]
    near(f"{name}/feats", dtype, rel_err(ret[4].cpu(), g[f"{name}/feats"]), tf, yard_small(name, "feats", tf, scalar=False))
    assert torch.equal(ret[5].cpu(), torch.from_numpy(g[f"{name}/targets"]))
    if dtype == "fp32":
        check_snapshot(g, name, state_of(ms, cs), tp)
    torch.manual_seed(778)
    val = steps.cam_cr_validate(ns(), ms, cs, C.val_batches_cls(name, 4000, 1), C.val_batches_cls(name, 4100, 0), 1)
    near(f"{name}/val", dtype, relx(val[0], g[f"{name}/val"][0]), 2e-3, yard_small(name, "val", 6e-2))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_cam_wsi_probability_map_vs_reference(dtype):
    """'next' row f3: forward-only WSI tile classification (test_Camelyon16.test) against the reference's own map."""
    from ssl_cr_histo_amd.scripts import test_Camelyon16 as script
    _engine(dtype)
    name = "cam_wsi"
    g = load_golden(name)
    model, cls = build("finetune", "finetune", 2, True)
    loader = C.wsi_loader(name)
    pm = script.test(ns(), model, cls, loader)
    want = g[f"{name}/ret"]
    assert pm.shape == want.shape and pm.dtype == np.float64
    assert np.array_equal(pm == 0, ~g[f"{name}/mask"])
    # probabilities in [0,1]: absolute tolerance.  fp32 mode 1e-3 of the logit scale; bf16 measured, not 1e-3
    near(f"{name}/probability_map", dtype, np.abs(pm - want).max(), 1e-3, 6e-2)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_cam_cr_full_size_step_vs_reference(dtype):
    """One Camelyon SSL_CR iteration (cross-entropy + hard pseudo-label cross-entropy, SGD-Nesterov, the reference's three
    torch.randperm shuffles) at the size of the benchmark workload -- student 640 / teacher 448 images of 256x256 -- against
    reductions of the reference's own iteration: loss averages and accuracy, feature row norms / column sums, post-step
    snapshot, and every parameter gradient (norm + seeded +-1 projection of the reference's .grad; the engine keeps the
    gradients of the last backward, read here after the epoch function returned), against the float64 run of the same iteration.
    fp32: 1e-3 on losses/features, gradients within max(3e-3, 3 x the reference's own fp32 error); bf16: losses / features within yard16()'s multiples of the
    emulated bf16-storage error, gradients within 2 x / 3.5 x the emulated bf16-storage error + 0.05."""
    from ssl_cr_histo_amd import steps
    eng = _engine(dtype)
    name = "cam_cr_full"
    c = C.CASES[name]
    g = load_golden(name)
    mt, ct = build("finetune", "finetune", 2, True)
    ms, cs = build("finetune", "finetune", 2, True)
    freeze(mt, 64)
    freeze(ms, c["modules"])
    opt = torch.optim.SGD(filter(lambda p: p.requires_grad, list(ms.parameters()) + list(cs.parameters())), lr=c["lr"],
                          momentum=0.9, weight_decay=c["wd"], nesterov=True)
    torch.manual_seed(777)
    ret = steps.cam_cr_train(ns(lambda_u=c["lambda_u"], image_size=c["hw"]), mt, ms, ct, cs,
                             C.labeled_batches_cls(name, 1000, 1), C.labeled_batches_cls(name, 1100, 0),
                             C.unlabeled_batches(name, 2000), C.unlabeled_batches(name, 2100), opt, 1)
    ts, tf, tp = TOLS[dtype]
    y16 = yard16(name)
    for i in range(3):
        near(f"{name}/ret{i}", dtype, relx(ret[i], g[f"{name}/ret"][i]), ts, y16["ret"][i])
    f = ret[4].cpu().double()
    assert list(f.shape) == list(g[f"{name}/feats_shape"])
    near(f"{name}/feats_rowl2", dtype, rel_err(f.norm(dim=1), g[f"{name}/feats_rowl2"]), tf, y16["rowl2"])
    near(f"{name}/feats_colsum", dtype, rel_err(f.sum(0), g[f"{name}/feats_colsum"]), tf, y16["colsum"])
    near(f"{name}/feats_head", dtype, rel_err(ret[4][:4].cpu(), g[f"{name}/feats_head"]), tf, y16["head"])
    assert torch.equal(ret[5].cpu(), torch.from_numpy(g[f"{name}/targets"]))
    if dtype == "fp32":
        assert abs(ret[3] - g[f"{name}/ret"][3]) <= 1.0 / 192 + 1e-9            # accuracy (fraction) over 192 labeled images: at most one flip
        check_snapshot(g, name, state_of(ms, cs), tp)
    st = eng.bind(ms, cs)                      # the cached binding of the epoch function: gradients of its last backward
    names = [str(n) for n in g[f"{name}/grad_names"]]
    assert names == [k for k, _ in list(ms.named_parameters()) + list(cs.named_parameters())]
    # against the FLOAT64 run of the same iteration (same shuffles), like the BreastPathQ full-size case: fp32 within
    # max(3e-3, 3 x the reference's own fp32 error); bf16 within 2 x (norm) / 3.5 x (one +-1 projection) the error that bf16
    # storage alone causes in the CPU emulation of this iteration (oracle/bf16_emul.py) + 0.05
    grad_rows_check(name, dtype, names, st.grad, g)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_kather_cr_epoch_vs_reference(dtype):
    from ssl_cr_histo_amd import steps
    _engine(dtype)
    name = "kather_cr_f0"
    c = C.CASES[name]
    g = load_golden(name)
    mt, ct = build("finetune", "finetune", 9, True)
    ms, cs = build("finetune", "finetune", 9, True)
    freeze(mt, 64)
    freeze(ms, c["modules"])
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, list(ms.parameters()) + list(cs.parameters())), lr=c["lr"],
                           betas=(0.9, 0.999), weight_decay=c["wd"])
    ret = steps.kather_cr_train(ns(lambda_u=c["lambda_u"]), mt, ms, ct, cs, C.labeled_batches_kather(name),
                                C.unlabeled_batches(name), opt, 1)
    ts, tf, tp = TOLS[dtype]
    for i in range(3):
        near(f"{name}/ret{i}", dtype, relx(ret[i], g[f"{name}/ret"][i]), ts, yard_small(name, f"ret{i}", ts))
    if dtype == "fp32":
        assert ret[3] == g[f"{name}/ret"][3]
        check_snapshot(g, name, state_of(ms, cs), tp)
    val = steps.kather_cr_validate(ns(), ms, cs, C.val_batches_kather(name), 1)
    near(f"{name}/val", dtype, relx(val[0], g[f"{name}/val"][0]), 5e-3, yard_small(name, "val", 1e-1))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_rsp_epoch_and_lookahead_vs_reference(dtype):
    from ssl_cr_histo_amd import steps
    from ssl_cr_histo_amd.lookahead import Lookahead
    _engine(dtype)
    name = "rsp"
    c = C.CASES[name]
    g = load_golden(name)
    model, cls = build("triplet", "mlp", 6, False)
    opt = torch.optim.SGD(list(model.parameters()) + list(cls.parameters()), lr=c["lr"], momentum=0.9, weight_decay=c["wd"],
                          nesterov=True)
    la = Lookahead(opt, la_steps=5, la_alpha=0.5)               # the reference's call site, pretrain_BreastPathQ.py:247
    a = ns(tile_h=c["hw"], tile_w=c["hw"])
    ret = steps.rsp_train(a, model, cls, C.rsp_batches(name), torch.nn.CrossEntropyLoss(), opt, 1)
    ts, tf, tp = TOLS[dtype]
    near("rsp/ret0", dtype, relx(ret[0], g["rsp/ret"][0]), ts, yard_small("rsp", "ret0", ts))
    # lr 0.01 SGD on BN statistics of 16 elements (layer4 at 64x64, B=4): bf16 noise is amplified by the 2nd step
    near("rsp/feats", dtype, rel_err(ret[2].cpu(), g["rsp/feats"]), tf, yard_small("rsp", "feats", 0.2, scalar=False))
    assert torch.equal(ret[3].cpu(), torch.from_numpy(g["rsp/targets"]))
    if dtype == "fp32":
        assert ret[1] == g["rsp/ret"][1]
        check_snapshot(g, "rsp", state_of(model, cls), tp)
    val = steps.rsp_validate(a, model, cls, C.rsp_batches(name, 3500), torch.nn.CrossEntropyLoss(), 1)
    near("rsp/val", dtype, relx(val[0], g["rsp/val"][0]), 5e-3, yard_small("rsp", "val", 1e-1))
    if dtype == "fp32":
        for _ in range(5):            # pretrain_BreastPathQ.py:293: Lookahead stepped with the last batch's stale gradients
            la.step()
        check_snapshot(g, "rsp/la5", state_of(model, cls), 2e-2)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_rsp_full_size_step_vs_reference(dtype):
    """One RSP pre-training iteration at the reference's default size (pretrain_BreastPathQ.py --batch_size 128: 3 x 128
    images of 256x256 through the shared TripletNet backbone, 6-way CE, SGD-Nesterov) against reductions of the reference's
    own iteration: loss / accuracy, feature row norms and column sums, post-step snapshot, and every parameter gradient
    through its L2 norm and a seeded +-1 projection, against the float64 run of the same iteration.  fp32: 1e-3 on loss/features,
    gradients within max(3e-3, 3 x the reference's own fp32 error); bf16: loss / features within yard16()'s multiples of the emulated
    bf16-storage error, gradients within 2 x / 3.5 x the emulated bf16-storage error + 0.05 (oracle/bf16_emul.py)."""
    from ssl_cr_histo_amd import steps
    eng = _engine(dtype)
    name = "rsp_full"
    c = C.CASES[name]
    g = load_golden(name)
    model, cls = build("triplet", "mlp", 6, False)
    opt = torch.optim.SGD(list(model.parameters()) + list(cls.parameters()), lr=c["lr"], momentum=0.9, weight_decay=c["wd"],
                          nesterov=True)
    a = ns(tile_h=c["hw"], tile_w=c["hw"])
    ret = steps.rsp_train(a, model, cls, C.rsp_batches(name), torch.nn.CrossEntropyLoss(), opt, 1)
    ts, tf, tp = TOLS[dtype]
    y16 = yard16(name)
    near(f"{name}/ret0", dtype, relx(ret[0], g[f"{name}/ret"][0]), ts, y16["ret"][0])
    f = ret[2].cpu().double()
    assert list(f.shape) == list(g[f"{name}/feats_shape"])
    near(f"{name}/feats_rowl2", dtype, rel_err(f.norm(dim=1), g[f"{name}/feats_rowl2"]), tf, y16["rowl2"])
    near(f"{name}/feats_colsum", dtype, rel_err(f.sum(0), g[f"{name}/feats_colsum"]), tf, y16["colsum"])
    near(f"{name}/feats_head", dtype, rel_err(ret[2][:4].cpu(), g[f"{name}/feats_head"]), tf, y16["head"])
    assert torch.equal(ret[3].cpu(), torch.from_numpy(g[f"{name}/targets"]))
    if dtype == "fp32":
        assert abs(ret[1] - g[f"{name}/ret"][1]) <= 100.0 / c["b"] + 1e-9      # accuracy in percent: at most one flipped prediction
        check_snapshot(g, name, state_of(model, cls), tp)
    # gradients of that iteration: fresh weights, the engine's step without the update
    model, cls = build("triplet", "mlp", 6, False)
    net_ = eng.bind(model, cls)
    model.train()
    cls.train()
    (i1, i2, i3, tgt), = C.rsp_batches(name)
    hw = c["hw"]
    eng.step_supervised(net_, "ce", [v.reshape(-1, 3, hw, hw) for v in (i1, i2, i3)], tgt.long().reshape(-1), train=True)
    # bf16: the three branches ran as segments of one launch per layer (sslcr_conv_desc.seg_images); fp32: pass by pass
    assert net_.segments_used == (dtype == "bf16")
    names = [str(n) for n in g[f"{name}/grad_names"]]
    mine = [k for k, _ in list(model.named_parameters()) + list(cls.named_parameters())]
    assert names == mine
    # against the FLOAT64 run of the same iteration: fp32 within max(3e-3, 3 x the reference's own fp32 error); bf16 within
    # 2 x / 3.5 x the emulated bf16-storage error + 0.05 (norm / one +-1 projection) -- the rule of the SSL_CR full-size cases
    grad_rows_check(name, dtype, names, net_.grad, g)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_supervised_epochs_vs_reference(dtype):
    from ssl_cr_histo_amd import steps
    _engine(dtype)
    ts, tf, tp = TOLS[dtype]
    name = "cam_sup"
    c = C.CASES[name]
    g = load_golden(name)
    ms, cs = build("finetune", "finetune", 2, False)
    opt = torch.optim.SGD(list(ms.parameters()) + list(cs.parameters()), lr=c["lr"], momentum=0.9, weight_decay=c["wd"], nesterov=True)
    torch.manual_seed(779)
    ret = steps.cam_sup_train(ns(image_size=c["hw"]), ms, cs, C.labeled_batches_cls(name, 1000, 1),
                              C.labeled_batches_cls(name, 1100, 0), opt, 1)
    near(f"{name}/ret0", dtype, relx(ret[0], g[f"{name}/ret"][0]), ts, yard_small(name, "ret0", ts))
    near(f"{name}/feats", dtype, rel_err(ret[2].cpu(), g[f"{name}/feats"]), tf, yard_small(name, "feats", tf, scalar=False))
    if dtype == "fp32":
        check_snapshot(g, name, state_of(ms, cs), tp)
    name = "bpq_sup"
    c = C.CASES[name]
    g = load_golden(name)
    ms, cs = build("finetune", "finetune", 1, False)
    opt = torch.optim.Adam(list(ms.parameters()) + list(cs.parameters()), lr=c["lr"], betas=(0.9, 0.999), weight_decay=c["wd"])
    ret = steps.bpq_sup_train(ns(image_size=c["hw"]), ms, cs, C.labeled_batches(name), torch.nn.MSELoss(), opt, 1)
    near(f"{name}/ret0", dtype, relx(ret[0], g[f"{name}/ret"][0]), ts, yard_small(name, "ret0", ts))
    # Adam lr 1e-3 moves every weight by ~lr regardless of gradient size: bf16 gradient noise shows after step 1
    near(f"{name}/feats", dtype, rel_err(ret[1].cpu(), g[f"{name}/feats"]), tf, yard_small(name, "feats", 0.2, scalar=False))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("kind", ["mse", "ce"])
def test_gradients_vs_oracle(kind, dtype):
    """full fine-tune SSL_CR step: every parameter gradient of the engine against oracle autograd (same inputs).

    fp32 mode: 3e-3 relative L2 per parameter (measured ~1e-5).  bf16 mode: bf16 STORAGE alone moves these gradients by
    20-50 % on this tiny random-weight problem (oracle/bf16_emul.py reproduces that on CPU), so the engine is held to
    2x the emulated-bf16 error + 0.05 per parameter -- a kernel bug shows up far above that."""
    from oracle import bf16_emul as B
    eng = _engine(dtype)
    classes = 1 if kind == "mse" else 2
    hw, nx, nu, lam = 64, 4, 6, 0.7
    mt, ct = build("finetune", "finetune", classes, True)
    ms, cs = build("finetune", "finetune", classes, True)
    freeze(mt, 64)
    x, u_w, u_s = C.u8(7001, (nx, 3, hw, hw)), C.u8(7002, (nu, 3, hw, hw)), C.u8(7003, (nu, 3, hw, hw))
    y = C.f32(7004, (nx,)) if kind == "mse" else C.ints(7005, (nx,), classes)
    te, st = eng.bind(mt, ct), eng.bind(ms, cs)
    mt.eval()
    ms.train()
    r = eng.step_ssl_cr(te, st, kind, x, y, u_w, u_s, lam)
    # oracle: teacher logits (eval, no grad), then student loss + autograd
    pn_s, _, pc_s = oracle_state("finetune", classes, True)
    pn_t, bn_t, pc_t = oracle_state("finetune", classes, True)
    ps, pt = merged(pn_s, pc_s), merged(pn_t, pc_t)
    for v in ps.values():
        v.requires_grad_(True)
    with torch.no_grad():
        lt = OM.classifier_forward(pt, OM.finetune_forward(pt, bn_t, u_w.float(), False, True))
    g32, logits, loss = B.ssl_cr_grads(kind, ps, x.float(), y, u_s.float(), lt, lam, emulate=False)
    got = r["losses"].cpu()
    c_loss = c_logits = c_lt = 6e-2
    if dtype == "bf16":
        # independent ceilings: the same step on the CPU with bf16 STORAGE emulated in the student (train mode) and in the teacher
        # (BatchNorm folded into bf16 filters), oracle/bf16_emul.py -- 5 x its loss error (a scalar: partly cancelling), 2 x its
        # element-wise logit errors, never above the old flat 6e-2
        lt16 = B.teacher_logits(pt, bn_t, u_w.float(), True)
        _, logits16, loss16 = B.ssl_cr_grads(kind, ps, x.float(), y, u_s.float(), lt16, lam, emulate=True)
        c_loss = min(c_loss, 5 * max(relx(loss16, loss), 1e-3))
        c_logits = min(c_logits, 2 * float(rel_err(logits16, logits)))
        c_lt = min(c_lt, 2 * float(rel_err(lt16, lt)))
        g16, _, _ = B.ssl_cr_grads(kind, ps, x.float(), y, u_s.float(), lt, lam, emulate=True)
    near(f"grads_vs_oracle/{kind}/loss", dtype, relx(got[0], loss), 1e-3, c_loss)
    near(f"grads_vs_oracle/{kind}/logits", dtype, rel_err(r["logits"].cpu(), logits), 1e-3, c_logits)
    near(f"grads_vs_oracle/{kind}/logits_t", dtype, rel_err(r["logits_t"].cpu(), lt), 1e-3, c_lt)
    rows, bad = [], []
    for i, k in enumerate(ps.keys()):
        ref = g32[k].double()
        e = float((st.grad(i).cpu().double() - ref).norm() / (ref.norm() + 1e-30))
        bound = 3e-3 if dtype == "fp32" else 2.0 * float((g16[k].double() - ref).norm() / (ref.norm() + 1e-30)) + 0.05
        rows.append(f"   {i:2d} {k:40s} err {e:.3e}  ceiling {bound:.3e}")
        try:
            near(f"grads_vs_oracle/{kind}/grad/{i}", dtype, e, bound, bound, floor=1e-2)
        except AssertionError as ex:
            bad.append(rows[-1] + f"   <- {ex}")
    print(f"[{dtype}/{kind}] relative L2 gradient error per parameter:\n" + "\n".join(rows))
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_aux_stream_teacher_forward_is_the_same_step(dtype):
    """sslcr_set_aux_stream: the teacher forward on a second stream next to the student forward must not change a bit of
    the forward results (logits, teacher logits, losses); gradients equal up to the order of the fp32 atomics."""
    eng = _engine(dtype)
    hw, nx, nu = 64, 8, 12
    x, u_w, u_s = C.u8(7101, (nx, 3, hw, hw)), C.u8(7102, (nu, 3, hw, hw)), C.u8(7103, (nu, 3, hw, hw))
    y = C.f32(7104, (nx,))
    out = []
    try:
        for on in (False, True, True):
            eng.set_aux_stream(on)
            mt, ct = build("finetune", "finetune", 1, True)
            ms, cs = build("finetune", "finetune", 1, True)
            freeze(mt, 64)
            te, st = eng.bind(mt, ct), eng.bind(ms, cs)
            mt.eval()
            ms.train()
            r = eng.step_ssl_cr(te, st, "mse", x, y, u_w, u_s, 0.7)
            torch.cuda.synchronize()
            out.append((r["losses"].cpu(), r["logits"].cpu(), r["logits_t"].cpu(), [st.grad(i).cpu() for i in (0, 3, 30, 60)]))
    finally:
        eng.set_aux_stream(False)
    for o in out[1:]:
        assert torch.equal(o[0], out[0][0]) and torch.equal(o[1], out[0][1]) and torch.equal(o[2], out[0][2])
        for a, b in zip(o[3], out[0][3]):
            assert rel_err(a, b) < 1e-5


def test_deepcopy_teacher_and_state_dict_roundtrip():
    """teacher = copy.deepcopy(student) (eval_BreastPathQ_SSL_CR.py:515-516) and checkpoint key compatibility."""
    eng = _engine("fp32")
    ms, cs = build("finetune", "finetune", 2, True)
    x = C.u8(8001, (2, 3, 64, 64)).to(DEV)
    ms.eval()
    f0 = ms(x)
    mt = copy.deepcopy(ms)
    f1 = mt(x)
    assert torch.equal(f0, f1)
    sd = {"module." + k: v for k, v in ms.state_dict().items()}
    from ssl_cr_histo_amd.net import TripletNet_Finetune, strip_module_prefix
    m2 = TripletNet_Finetune("resnet18").to(DEV)
    m2.load_state_dict(strip_module_prefix(sd))
    m2.eval()
    assert torch.equal(m2(x), f0)
    ref_keys = list(OM.init_state(1, OM.net_param_specs()).keys())
    assert list(ms.state_dict().keys()) == ref_keys
    # in-place teacher refresh as EMA with decay 0 == deepcopy
    from ssl_cr_histo_amd import steps
    mt2, ct2 = build("finetune", "finetune", 2, False, seed=7)
    steps.teacher_refresh(mt2, ct2, ms, cs, 0.0)
    mt2.eval()
    assert torch.equal(mt2(x), f0)


def test_checkpoint_resume_roundtrip():
    """reference-style checkpoint dict ('model', 'classifier', 'optimizer' + `module.` prefix, eval_Camelyon_SSL_CR.py:575-590)
    saved after step 1 and resumed into fresh modules continues exactly like the uninterrupted run."""
    import io
    from ssl_cr_histo_amd.net import strip_module_prefix
    eng = _engine("fp32")
    hw, nx, nu = 64, 4, 4
    batches = [(C.u8(9000 + i, (nx, 3, hw, hw)), C.ints(9010 + i, (nx,), 2), C.u8(9020 + i, (nu, 3, hw, hw)),
                C.u8(9030 + i, (nu, 3, hw, hw))) for i in range(2)]

    def make():
        mt, ct = build("finetune", "finetune", 2, True)
        ms, cs = build("finetune", "finetune", 2, True)
        freeze(mt, 64)
        mt.eval(); ms.train()
        opt = torch.optim.Adam(list(ms.parameters()) + list(cs.parameters()), lr=1e-3, weight_decay=1e-4)
        return mt, ct, ms, cs, opt

    def step(mt, ct, ms, cs, opt, b):
        te, st = eng.bind(mt, ct), eng.bind(ms, cs)
        r = eng.step_ssl_cr(te, st, "ce", b[0], b[1], b[2], b[3], 1.0)
        st.optimizer_step(opt)
        return r["losses"].cpu()

    mt, ct, ms, cs, opt = make()
    step(mt, ct, ms, cs, opt, batches[0])
    buf = io.BytesIO()
    torch.save({"model": {"module." + k: v for k, v in ms.state_dict().items()},
                "classifier": {"module." + k: v for k, v in cs.state_dict().items()}, "optimizer": opt.state_dict(), "epoch": 1}, buf)
    l2 = step(mt, ct, ms, cs, opt, batches[1])
    buf.seek(0)
    ck = torch.load(buf, map_location=DEV, weights_only=False)
    mt2, ct2, ms2, cs2, opt2 = make()
    ms2.load_state_dict(strip_module_prefix(ck["model"]))
    cs2.load_state_dict(strip_module_prefix(ck["classifier"]))
    opt2.load_state_dict(ck["optimizer"])
    l2b = step(mt2, ct2, ms2, cs2, opt2, batches[1])
    assert torch.allclose(l2, l2b, rtol=1e-5, atol=1e-7), (l2, l2b)
    for (k, a), (_, b) in zip(ms.state_dict().items(), ms2.state_dict().items()):
        assert rel_err(b.float().cpu(), a.float().cpu()) < 2e-5, k


def test_collective_code_path_selftest_world1():
    """SSLCR_COMM_SELFTEST=1 builds the engine's two RCCL communicators with ONE rank and routes a full fine-tune step through
    every collective call site (synced-BN all-reduces forward and backward, bucketed gradient all-reduce on the side stream,
    event joins).  With one rank the collectives are identities, so the step must reproduce the plain path."""
    import subprocess
    import sys
    code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import cases as C
from ssl_cr_histo_amd import engine as E
import test_engine_gpu as T
eng = E.set_engine(E.Engine("cuda:0", "fp32"))
if os.environ.get("SSLCR_COMM_SELFTEST"):
    eng.init_comm(0, 1, lambda b: b)
    eng.set_bn_sync(os.environ.get("SELFTEST_BN_SYNC", "1") == "1")
mt, ct = T.build("finetune", "finetune", 2, True); ms, cs = T.build("finetune", "finetune", 2, True)
T.freeze(mt, 64); mt.eval(); ms.train()
te, st = eng.bind(mt, ct), eng.bind(ms, cs)
x, u_w, u_s = C.u8(1, (4, 3, 64, 64)), C.u8(2, (4, 3, 64, 64)), C.u8(3, (4, 3, 64, 64))
y = C.ints(4, (4,), 2)
opt = torch.optim.SGD(list(ms.parameters()) + list(cs.parameters()), lr=1e-2, momentum=0.9, nesterov=True)
for _ in range(2):
    r = eng.step_ssl_cr(te, st, "ce", x, y, u_w, u_s, 1.0); st.optimizer_step(opt)
torch.cuda.synchronize()
print("RESULT", " ".join(f"{v:.7e}" for v in r["losses"].cpu().tolist()), f"{float(dict(ms.named_parameters())['model.conv1.weight'].double().norm()):.9e}",
      f"{float(dict(ms.named_parameters())['model.layer4.1.bn2.weight'].double().sum()):.9e}")
'''
    outs = []
    for flag in ("", "1", "per-replica"):          # plain ; communicators + synced BN ; communicators + per-replica BN
        env = dict(os.environ)
        env.pop("SSLCR_COMM_SELFTEST", None)
        if flag:
            env["SSLCR_COMM_SELFTEST"] = "1"
            env["SELFTEST_BN_SYNC"] = "0" if flag == "per-replica" else "1"
        p = subprocess.run([sys.executable, "-c", code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][-1]
        outs.append([float(v) for v in line.split()[1:]])
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), outs
