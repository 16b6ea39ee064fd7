"""Round-2 step-level parity on the MI355X (through the C-ABI): BASELINE config 1 (Kather supervised, 224x224, 9 classes), the
EMA teacher against the oracle, reference-layout checkpoints (row f2: read the reference's file, continue like the reference),
multi-iteration trajectories (bf16 fidelity as a bounded, measured quantity), virtual-rank sharding."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cases as C  # noqa: E402
from oracle import model as OM  # noqa: E402
from oracle import steps as S  # noqa: E402

from _util import check_snapshot, held, load_golden, merged, oracle_state, rebuild_ckpt, rel_err, yard_small  # noqa: E402
from test_engine_gpu import DEV, TOLS, _engine, build, freeze, grad_rows_check, near, ns, relx, state_of  # noqa: E402


# ------------------------------------------------------------------------------------------------ config 1: Kather supervised
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_kather_supervised_epoch_vs_reference(dtype):
    """eval_Kather_SSL.train / validate (golden from the reference file's own functions) at 96x96: after the stem the maps are
    24/12/6/3 pixels -- none of them 16-tileable, so every conv runs on the engine's generic shapes."""
    from ssl_cr_histo_amd import steps
    _engine(dtype)
    name = "kather_sup"
    c = C.CASES[name]
    g = load_golden(name)
    ms, cs = build("finetune", "finetune", c["classes"], False)
    opt = torch.optim.Adam(list(ms.parameters()) + list(cs.parameters()), lr=c["lr"], betas=(0.9, 0.999), weight_decay=c["wd"])
    crit = torch.nn.CrossEntropyLoss()
    ret = steps.kather_sup_train(ns(image_size=c["hw"]), ms, cs, C.sup_batches_kather(name), crit, opt, 1)
    ts, tf, tp = TOLS[dtype]
    near(f"{name}/ret0", dtype, relx(ret[0], g[f"{name}/ret"][0]), ts, yard_small(name, "ret0", ts))
    if dtype == "fp32":
        assert ret[1] == g[f"{name}/ret"][1]
        check_snapshot(g, name, state_of(ms, cs), tp)
    val = steps.kather_sup_validate(ns(), ms, cs, C.val_batches_kather(name), crit, 1)
    near(f"{name}/val", dtype, relx(val[0], g[f"{name}/val"][0]), 5e-3, yard_small(name, "val", 1e-1))
    if dtype == "fp32":
        assert val[1] == g[f"{name}/val"][1]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_kather_config1_full_size_step_vs_reference(dtype):
    """BASELINE.json config 1 itself: ONE iteration of eval_Kather_SSL.train at --batch_size 32, 224x224, 9 classes (96 images;
    56/28/14/7 maps), Adam lr 1e-5 -- loss/accuracy, post-step snapshot, validate(), and every parameter gradient against the
    float64 run of the same iteration (fp32: max(3e-3, 3 x the reference's own fp32 error); bf16: 2 x / 3.5 x the emulated
    bf16-storage error + 0.05 for norm / projection -- the rule of the BreastPathQ full-size case)."""
    from ssl_cr_histo_amd import steps
    eng = _engine(dtype)
    name = "kather_sup_full"
    c = C.CASES[name]
    g = load_golden(name)
    ms, cs = build("finetune", "finetune", c["classes"], False)
    opt = torch.optim.Adam(list(ms.parameters()) + list(cs.parameters()), lr=c["lr"], betas=(0.9, 0.999), weight_decay=c["wd"])
    crit = torch.nn.CrossEntropyLoss()
    ret = steps.kather_sup_train(ns(image_size=c["hw"]), ms, cs, C.sup_batches_kather(name), crit, opt, 1)
    ts, tf, tp = TOLS[dtype]
    near(f"{name}/ret0", dtype, relx(ret[0], g[f"{name}/ret"][0]), ts, ts)
    assert abs(ret[1] - g[f"{name}/ret"][1]) <= (1.0 if dtype == "fp32" else 3.0) / 96 + 1e-9
    if dtype == "fp32":
        check_snapshot(g, name, state_of(ms, cs), tp)
    st = eng.bind(ms, cs)                      # the epoch function's binding: gradients of its (only) backward
    names = [str(n) for n in g[f"{name}/grad_names"]]
    assert names == [k for k, _ in list(ms.named_parameters()) + list(cs.named_parameters())]
    grad_rows_check(name, dtype, names, st.grad, g)
    val = steps.kather_sup_validate(ns(), ms, cs, C.val_batches_kather(name), crit, 1)
    near(f"{name}/val", dtype, relx(val[0], g[f"{name}/val"][0]), 5e-3, 1e-1)


# ------------------------------------------------------------------------------------------------ a11: EMA teacher
@pytest.mark.parametrize("decay", [0.99, 0.5, 0.0])
def test_ema_teacher_vs_oracle(decay):
    """north_star's EMA teacher: teacher <- decay * teacher + (1 - decay) * student on every parameter, BatchNorm buffers copied
    (decay 0 == the reference's copy.deepcopy, eval_BreastPathQ_SSL_CR.py:515-516) -- against oracle/steps.py:teacher_refresh, and
    the refreshed teacher's eval forward against the oracle's forward with the oracle's refreshed state."""
    from ssl_cr_histo_amd import steps
    _engine("fp32")
    mt, ct = build("finetune", "finetune", 2, True, seed=7)
    ms, cs = build("finetune", "finetune", 2, True, seed=C.PARAM_SEED)
    with torch.no_grad():                                        # a student that has trained: counters differ from the teacher's
        for b in ms.model.bn_modules():
            b.num_batches_tracked += 9

    def oracle_of(seed):
        sd = OM.init_state(seed, OM.net_param_specs(), random_running_stats=True)
        csd = OM.init_state(seed + 1, OM.classifier_param_specs("finetune", 2))
        p, b = OM.split_state(sd)
        pc, _ = OM.split_state(csd)
        return merged(p, pc), b
    pt, bt = oracle_of(7)
    ps, bs = oracle_of(C.PARAM_SEED)
    for k in bs:
        if k.endswith("num_batches_tracked"):
            bs[k] = bs[k] + 9
    S.teacher_refresh(ps, bs, pt, bt, decay)
    steps.teacher_refresh(mt, ct, ms, cs, decay)
    torch.cuda.synchronize()
    got = state_of(mt, ct)
    for k, v in list(pt.items()) + list(bt.items()):
        if k.endswith("num_batches_tracked"):
            assert int(got[k]) == int(v), k
        else:
            assert torch.allclose(got[k], v.detach(), rtol=2e-6, atol=1e-7), (k, float((got[k] - v.detach()).abs().max()))
    x = C.u8(8100, (3, 3, 64, 64))
    mt.eval(); ct.eval()
    logits = ct(mt(x.to(DEV)))
    with torch.no_grad():
        want = OM.classifier_forward(pt, OM.finetune_forward(pt, bt, x.float(), False, False))
    assert rel_err(logits.cpu(), want) < 1e-3


# ------------------------------------------------------------------------------------------------ f2: reference checkpoints
def _opt_for(c, ms, cs):
    prm = filter(lambda p: p.requires_grad, list(ms.parameters()) + list(cs.parameters()))
    if c["opt"] == "adam":
        return torch.optim.Adam(prm, lr=c["lr"], betas=(0.9, 0.999), weight_decay=c["wd"])
    return torch.optim.SGD(prm, lr=c["lr"], momentum=0.9, weight_decay=c["wd"], nesterov=True)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("source", ["reference_file", "own_file"])
def test_checkpoint_ssl_cr_resume_continues_like_the_reference(source, dtype, tmp_path):
    """eval_BreastPathQ_SSL_CR layout.  'reference_file': the .pt the REFERENCE wrote after its epoch 1 (rebuilt from the fixture)
    is resumed through ssl_cr_histo_amd.checkpoint and epoch 2 must equal the reference's own continued run.  'own_file': epoch 1
    on the engine, teacher refresh, save in the reference's layout, resume into fresh modules, epoch 2 -- same golden."""
    from ssl_cr_histo_amd import checkpoint as CK, steps
    _engine(dtype)
    name = "ckpt_bpq_cr"
    c = C.CASES[name]
    g = load_golden(name)
    ts, tf, tp = TOLS[dtype]

    def fresh():
        mt, ct = build("finetune", "finetune", 1, True)
        ms, cs = build("finetune", "finetune", 1, True)
        freeze(mt, 64)
        freeze(ms, c["modules"])
        return mt, ct, ms, cs, _opt_for(c, ms, cs)
    f = str(tmp_path / "fine_CR_trained_model_1.pt")
    a = ns(lambda_u=c["lambda_u"])
    if source == "reference_file":
        sd0 = OM.init_state(C.PARAM_SEED, OM.net_param_specs(), random_running_stats=True)
        cd0 = OM.init_state(C.PARAM_SEED + 1, OM.classifier_param_specs("finetune", 1))
        ref, _ = rebuild_ckpt(name, {"model_student": (sd0, False), "model_teacher": (sd0, False),
                                     "classifier_student": (cd0, False), "classifier_teacher": (cd0, False)})
        torch.save(ref, f)
    else:
        mt, ct, ms, cs, opt = fresh()
        r1 = steps.bpq_cr_train(a, mt, ms, ct, cs, C.labeled_batches(name), C.unlabeled_batches(name), opt, 1)
        for i in range(3):
            near(f"{name}/ret{i}", dtype, relx(r1[i], g[f"{name}/ret"][i]), ts, ts)
        mt, ct = copy.deepcopy(ms), copy.deepcopy(cs)                                   # :515-516
        CK.save_ssl_cr(f, None, ms, mt, ct, cs, opt, 1, r1[0], r1[1], r1[2])
    mt, ct, ms, cs, opt = fresh()
    start, _ = CK.resume(f, opt, map_location=DEV, model_student=ms, model_teacher=mt, classifier_teacher=ct, classifier_student=cs)
    assert start == 2
    r2 = steps.bpq_cr_train(a, mt, ms, ct, cs, C.labeled_batches(name, 1200), C.unlabeled_batches(name, 2200), opt, start)
    for i in range(3):
        near(f"{name}/{source}/ret2_{i}", dtype, relx(r2[i], g[f"{name}/ret2"][i]), ts, ts)
    near(f"{name}/{source}/feats2", dtype, rel_err(r2[3].cpu(), g[f"{name}/feats2"]), tf, tf)
    if dtype == "fp32":
        check_snapshot(g, name + "/e2", state_of(ms, cs), tp)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_checkpoint_finetune_resume_and_ssl_cr_start(dtype, tmp_path):
    """eval_Camelyon_SSL layout with DataParallel keys, as the reference wrote it: --resume -> epoch 2 equals the reference's; and the
    SSL_CR scripts' way in (teacher + student from 'model' / 'classifier', module. stripped) gives working nets."""
    from ssl_cr_histo_amd import checkpoint as CK, steps
    _engine(dtype)
    name = "ckpt_cam_sup"
    c = C.CASES[name]
    g = load_golden(name)
    ts, tf, tp = TOLS[dtype]
    sd0 = OM.init_state(C.PARAM_SEED, OM.net_param_specs())
    cd0 = OM.init_state(C.PARAM_SEED + 1, OM.classifier_param_specs("finetune", 2))
    ref, _ = rebuild_ckpt(name, {"model": (sd0, True), "classifier": (cd0, True)})
    f = str(tmp_path / "fine_tuned_model_1.pt")
    torch.save(ref, f)
    ms, cs = build("finetune", "finetune", 2, False)
    freeze(ms, c["modules"])
    opt = _opt_for(c, ms, cs)
    msw, csw = torch.nn.DataParallel(ms), torch.nn.DataParallel(cs)               # eval_Camelyon_SSL.py:355-356
    start, _ = CK.resume(f, opt, map_location=DEV, model=msw, classifier=csw)
    torch.manual_seed(782)
    r2 = steps.cam_sup_train(ns(image_size=c["hw"]), msw, csw, C.labeled_batches_cls(name, 1200, 1), C.labeled_batches_cls(name, 1300, 0),
                             opt, start)
    near(f"{name}/ret2_0", dtype, relx(r2[0], g[f"{name}/ret2"][0]), ts, ts)
    near(f"{name}/feats2", dtype, rel_err(r2[2].cpu(), g[f"{name}/feats2"]), tf, tf)
    if dtype == "fp32":
        assert r2[1] == g[f"{name}/ret2"][1]
        check_snapshot(g, name + "/e2", state_of(ms, cs), tp)
    mt, ct = build("finetune", "finetune", 2, True, seed=3)
    m2, c2 = build("finetune", "finetune", 2, True, seed=4)
    CK.load_finetuned(f, (mt, m2), (ct, c2), map_location=DEV)
    x = C.u8(8200, (2, 3, 64, 64)).to(DEV)
    mt.eval(); m2.eval()
    assert torch.equal(mt(x), m2(x))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_checkpoint_pretrain_resume_continues_like_the_reference(dtype, tmp_path):
    """pretrain_BreastPathQ layout: epoch 1 + the per-epoch Lookahead 'scheduler.step()' on the engine, saved as the reference saves
    it ('model' + 'optimizer', DataParallel keys, no classifier), --resume into fresh modules (classifier back at its seeded
    initialisation, like a fresh reference process), epoch 2 against the reference's own resumed run."""
    from ssl_cr_histo_amd import checkpoint as CK, steps
    from ssl_cr_histo_amd.lookahead import Lookahead
    _engine(dtype)
    name = "ckpt_rsp"
    c = C.CASES[name]
    g = load_golden(name)
    ts, tf, tp = TOLS[dtype]

    def fresh():
        model, cls = build("triplet", "mlp", 6, False)
        opt = torch.optim.SGD(list(model.parameters()) + list(cls.parameters()), lr=c["lr"], momentum=0.9, weight_decay=c["wd"], nesterov=True)
        return model, cls, opt, Lookahead(opt, la_steps=5, la_alpha=0.5)
    model, cls, opt, sched = fresh()
    a = ns(tile_h=c["hw"], tile_w=c["hw"])
    crit = torch.nn.CrossEntropyLoss()
    r1 = steps.rsp_train(a, model, cls, C.rsp_batches(name), crit, opt, 1)
    sched.step()                                                                   # pretrain_BreastPathQ.py:293
    near(f"{name}/ret0", dtype, relx(r1[0], g[f"{name}/ret"][0]), ts, ts)
    if dtype == "fp32":
        check_snapshot(g, name + "/e1", state_of(model, cls), 2e-2)
    f = str(tmp_path / "model_1.pt")
    CK.save_pretrain(f, None, model, sched, 1, r1[0], r1[1], data_parallel_keys=True)
    model, cls, opt, sched = fresh()
    start, ck = CK.resume(f, opt, map_location=DEV, model=model)
    assert start == 2 and "classifier" not in ck
    r2 = steps.rsp_train(a, model, cls, C.rsp_batches(name, 3200), crit, opt, start)
    if dtype == "fp32":
        # this run amplifies fp32 round-off (lr 0.01 SGD-Nesterov, stale-gradient Lookahead step, BatchNorm over 16-element maps):
        # the golden's float64 run says how far the REFERENCE's own fp32 epoch-2 features are from the exact ones (several
        # percent); the engine is held to twice that distance, against float64
        ref_err = float(g[f"{name}/feats2_ref32_err"][0])
        assert abs(r2[0] - g[f"{name}/ret2"][0]) <= 5e-3 * g[f"{name}/ret2"][0], (r2[0], g[f"{name}/ret2"][0])
        assert abs(r2[0] - g[f"{name}/ret2_f64"][0]) <= 5e-3 * g[f"{name}/ret2_f64"][0]
        e = rel_err(r2[2].cpu(), g[f"{name}/feats2_f64"])
        print(f"epoch-2 features vs float64: engine {e:.3e}, reference fp32 {ref_err:.3e}")
        assert e <= max(2e-3, 2.0 * ref_err), (e, ref_err)
        check_snapshot(g, name + "/e2", state_of(model, cls), 2e-2)
    else:
        held(f"{name}/ret2/{dtype}", relx(r2[0], g[f"{name}/ret2"][0]), 0.15, floor=2e-3)


# ------------------------------------------------------------------------------------------------ trajectories (bf16 fidelity)
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["traj_bpq_cr", "traj_cam_cr"])
def test_trajectory_vs_reference(name, dtype):
    """24 consecutive iterations of the reference's own train() (one call per iteration, one optimizer object, 256x256, full
    fine-tune; Adam / MSE for BreastPathQ, SGD-Nesterov / CE + pseudo-label CE for Camelyon), per-iteration returned losses,
    validate() every 4 iterations, final snapshot.  fp32 mode is held, against the FLOAT64 trajectory, to 3 x the distance of the
    reference's own fp32 runs from float64 (tests/golden/traj_*_yard.npz; never below the north-star 1e-3).
    bf16 mode (the mode the throughput is quoted in) is measured, printed and bounded: every iteration's losses within 3e-2 of the
    reference's, and the deviation must not keep growing -- the mean of the last eight iterations may not exceed 1.3 x the mean of
    iterations 9-16 (+0.2 %).  Measured: SGD-Nesterov / Camelyon stays at 0.3-2.5e-3 throughout; Adam / BreastPathQ (where every
    noisy gradient component becomes a full lr-sized step) climbs from 1e-3 to 1.4e-2 over the first twelve iterations while the
    loss itself falls by 30 %, then stays at 1.4-1.6e-2: a bounded offset, not a diverging run."""
    from ssl_cr_histo_amd import steps
    _engine(dtype)
    c = C.CASES[name]
    g = load_golden(name)
    cam = c["script"] == "cam_cr"
    mt, ct = build("finetune", "finetune", c["classes"], True)
    ms, cs = build("finetune", "finetune", c["classes"], True)
    freeze(mt, 64)
    freeze(ms, c["modules"])
    opt = _opt_for(c, ms, cs)
    want, wvals = g[f"{name}/ret"], g[f"{name}/vals"]
    dev, vdev, rets, vgot = [], [], [], []
    torch.manual_seed(780)
    for it in range(c["iters"]):
        if cam:
            r = steps.cam_cr_train(ns(lambda_u=c["lambda_u"], image_size=c["hw"]), mt, ms, ct, cs,
                                   C.labeled_batches_cls(name, 1000 + 7 * it, 1), C.labeled_batches_cls(name, 1100 + 7 * it, 0),
                                   C.unlabeled_batches(name, 2000 + 7 * it), C.unlabeled_batches(name, 2100 + 7 * it), opt, 1)
        else:
            r = steps.bpq_cr_train(ns(lambda_u=c["lambda_u"]), mt, ms, ct, cs, C.labeled_batches(name, 1000 + 7 * it),
                                   C.unlabeled_batches(name, 2000 + 7 * it), opt, 1)
        dev.append(max(abs(r[i] - want[it][i]) / abs(want[it][i]) for i in range(3)))
        rets.append([float(r[i]) for i in range(3)])
        if cam and dtype == "fp32":
            assert abs(r[3] - want[it][3]) <= 1.0 / 6 + 1e-9, (it, r[3], want[it][3])       # accuracy over 6 labeled images
        if (it + 1) % 4 == 0:
            if cam:
                v = steps.cam_cr_validate(ns(), ms, cs, C.val_batches_cls(name, 4000, 1), C.val_batches_cls(name, 4100, 0), 1)
            else:
                v = (steps.bpq_cr_validate(ns(), ms, cs, C.val_batches_reg(name), 1),)
            vdev.append(abs(v[0] - wvals[len(vdev)][0]) / abs(wvals[len(vdev)][0]))
            vgot.append(float(v[0]))
    print(f"[{dtype}] {name}: per-iteration max relative loss deviation from the reference:\n   " + " ".join(f"{d:.2e}" for d in dev) +
          "\n   validate() after every 4th iteration: " + " ".join(f"{d:.2e}" for d in vdev))
    if dtype == "fp32":
        # What may a correct fp32 implementation deviate by after 24 optimizer steps?  Measured, not asserted: traj_*_yard.npz
        # (tests/golden/make_golden.py:gen_traj_yard) holds the trajectory in FLOAT64 and the reference's own fp32 run at 8, 3
        # and 1 threads (three valid fp32 summation orders).  The engine is held, against float64, to 3 x the largest distance
        # of a reference fp32 run from float64 (never below the north-star 1e-3) -- on every iteration's losses, on validate()
        # and on the final state.  (Adam / BreastPathQ: the reference's own validate() sits 2.8e-3 from float64 after FOUR
        # iterations and its per-iteration losses 9.6e-4 after 24; until round 4 this test held the engine to a constant 5e-3
        # against ONE of those fp32 runs and failed at 5.6e-3 on the driver's box.)
        y = load_golden(name + "_yard")
        yn = name + "_yard"
        r64, v64 = y[f"{yn}/ret_f64"], y[f"{yn}/vals_f64"]
        assert np.array_equal(y[f"{yn}/ret_t8"], want[:, :3]) and np.array_equal(y[f"{yn}/vals_t8"], wvals[:, 0])   # same reference run
        yard_ret = max(float(np.abs(y[f"{yn}/ret_t{t}"] / r64 - 1).max()) for t in (8, 3, 1))
        yard_val = max(float(np.abs(y[f"{yn}/vals_t{t}"] / v64 - 1).max()) for t in (8, 3, 1))
        dev64 = [max(abs(rets[it][i] - r64[it][i]) / abs(r64[it][i]) for i in range(3)) for it in range(c["iters"])]
        vdev64 = [abs(vgot[i] - v64[i]) / abs(v64[i]) for i in range(len(vgot))]
        print(f"   vs float64: losses max {max(dev64):.2e} (reference fp32 runs: {yard_ret:.2e}), validate() max {max(vdev64):.2e} "
              f"(reference fp32 runs: {yard_val:.2e})")
        b_ret, b_val = max(3.0 * yard_ret, 1e-3), max(3.0 * yard_val, 1e-3)
        held(f"{name}/iter_max_f64/fp32", max(dev64), b_ret, floor=b_ret)           # (recorded; the bound is the yardstick's)
        held(f"{name}/val_max_f64/fp32", max(vdev64), b_val, floor=b_val)
        # ... and against the fp32 golden itself (two fp32 runs, each up to `yard` from float64)
        assert max(dev) <= max(1e-3, 4.0 * yard_ret), dev
        assert max(vdev) <= max(1e-3, 4.0 * yard_val), vdev
        # final state: per-tensor norms against float64, each within 3 x the reference fp32 runs' own relative distance
        # ||a - a64|| / ||a64|| for that tensor (a conv weight in front of a BatchNorm has directions the loss does not depend
        # on; their gradient is round-off, Adam turns it into lr-sized steps, and the pre-BatchNorm running means drift: 2.3e-2
        # on layer4.1.bn1.running_mean between the reference's own fp32 run and float64)
        names = [str(n) for n in y[f"{yn}/names"]]
        st = state_of(ms, cs)
        yard_state = np.max(np.stack([y[f"{yn}/state_err_t{t}"] for t in (8, 3, 1)]), axis=0)
        worst = 0.0
        for i, k in enumerate(names):
            got = float(st[k].double().norm())
            e = abs(got - y[f"{yn}/l2_f64"][i]) / (y[f"{yn}/l2_f64"][i] + 1e-300)
            assert e <= max(3.0 * yard_state[i], 1e-3), (k, e, yard_state[i])
            worst = max(worst, e / max(yard_state[i], 1e-12))
        for k in ("model.layer4.1.bn2.running_mean", "model.bn1.running_mean"):
            a64 = torch.from_numpy(y[f"{yn}/t_f64/{k}"])
            e = float((st[k].double() - a64).norm() / a64.norm())
            ref = max(float((torch.from_numpy(y[f"{yn}/t_t{t}/{k}"]) - a64).norm() / a64.norm()) for t in (8, 3, 1))
            print(f"   {k}: engine {e:.2e} from float64, reference fp32 runs {ref:.2e}")
            assert e <= max(3.0 * ref, 1e-3), (k, e, ref)
        check_snapshot(g, name, st, max(1e-3, 4.0 * float(yard_state.max())))
    else:
        # every iteration within 2 x the LARGEST per-iteration deviation measured over the trajectory (tests/measured_errors.json;
        # which iteration carries the maximum moves with any change of summation order, the maximum itself does not), ceiling 3e-2
        held(f"{name}/iter_max/{dtype}", max(dev), 3e-2, floor=2e-3)
        mid, tail = float(np.mean(dev[8:16])), float(np.mean(dev[-8:]))
        assert tail <= 1.3 * mid + 2e-3, (mid, tail, dev)
        held(f"{name}/val_max/{dtype}", max(vdev), 5e-2, floor=2e-3)


# ------------------------------------------------------------------------------------------------ e: virtual ranks on one GPU
def _run_ranks(world, fn):
    """fn(rank) on `world` host threads, one torch stream each; re-raises the first failure."""
    import threading
    out, err = [None] * world, [None] * world
    streams = [torch.cuda.Stream(device=DEV) for _ in range(world)]

    def body(r):
        try:
            torch.cuda.set_device(torch.device(DEV))
            with torch.cuda.stream(streams[r]):
                out[r] = fn(r)
                streams[r].synchronize()
        except BaseException as e:          # noqa: BLE001 -- reported below
            err[r] = e
    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("workload", ["ssl_cr_ce", "ssl_cr_mse", "rsp"])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_virtual_ranks_equal_the_single_device_step(world, workload, dtype):
    """The engine's SHARDED code path against its own 1-rank step on the concatenated batch: `world` contexts on this one GPU,
    each with 1/world of the batch, exchanging through the virtual communicator (include/sslcr.h: sslcr_vcomm -- same call sites
    as RCCL: per-layer (sum, sum^2) all-reduce + global count in every train-mode BatchNorm forward, (sum g, sum g(x - mean))
    in backward, loss scaled by the GLOBAL batch, five gradient buckets summed on the side stream and joined before the
    optimizer).  Every rank must end with the single-device gradient (all 66/68 parameters), losses that add up to the
    single-device loss, the single-device BatchNorm running statistics and, after the optimizer step, the same parameters.
    fp32: 2e-5 relative L2 per gradient (summation order only).  bf16 mode: a last-bit difference in a BatchNorm scale (fp64 sums
    added in another order) re-rounds some bf16 activations, and on this tiny random-weight problem bf16 rounding noise alone is
    worth tens of percent on the backbone gradients (oracle/bf16_emul.py; tools/bf16_sources.py; measured here: a flat 16 % from
    conv1 to fc.0 with losses equal to 5e-4) -- so bf16 is only held to 0.6 on the gradients, 1e-2 on the losses and 0.1 on the post-step state (lr 1e-2 SGD of those gradients); the arithmetic is pinned by the fp32 rows."""
    from ssl_cr_histo_amd import engine as E
    if world == 8 and (workload != "ssl_cr_mse" or dtype != "fp32"):
        pytest.skip("world 8 (the node size of BASELINE configs 4 and 5) is covered once, on the headline workload in the parity mode")
    hw = 64
    rsp = workload == "rsp"
    kind = "mse" if workload == "ssl_cr_mse" else "ce"
    classes = 6 if rsp else (1 if kind == "mse" else 2)
    nx, nu = 2 * world, 3 * world

    def nets():
        if rsp:
            return build("triplet", "mlp", 6, False)
        return build("finetune", "finetune", classes, True)
    if rsp:
        xs = [C.u8(8300 + j, (nx, 3, hw, hw)) for j in range(3)]
        y = C.ints(8310, (nx,), 6)
    else:
        x, u_w, u_s = C.u8(8320, (nx, 3, hw, hw)), C.u8(8321, (nu, 3, hw, hw)), C.u8(8322, (nu, 3, hw, hw))
        y = C.f32(8323, (nx,)) if kind == "mse" else C.ints(8323, (nx,), 2)

    def one_step(eng, r, w):
        lo_x, hi_x = r * nx // w, (r + 1) * nx // w
        lo_u, hi_u = r * nu // w, (r + 1) * nu // w
        ms, cs = nets()
        ms.train(); cs.train()
        st = eng.bind(ms, cs)
        opt = torch.optim.SGD(list(ms.parameters()) + list(cs.parameters()), lr=1e-2, momentum=0.9, weight_decay=1e-4, nesterov=True)
        if rsp:
            res = eng.step_supervised(st, "ce", [v[lo_x:hi_x] for v in xs], y[lo_x:hi_x], train=True, n_global=nx)
        else:
            mt, ct = nets()
            freeze(mt, 64)
            mt.eval(); ct.eval()
            te = eng.bind(mt, ct)
            res = eng.step_ssl_cr(te, st, kind, x[lo_x:hi_x], y[lo_x:hi_x], u_w[lo_u:hi_u], u_s[lo_u:hi_u], 0.7, nx_global=nx, nu_global=nu)
        grads = [st.grad(i).cpu().double() for i in range(len(st.params))]
        st.optimizer_step(opt)
        # the loss meters' collective (steps._Meters -> sslcr_comm_all_reduce_f32): per-rank shares -> the global losses, on every rank
        summed = eng.all_reduce_sum(res["losses"].clone())
        torch.cuda.current_stream().synchronize()
        return dict(losses=res["losses"].cpu().double(), summed=summed.cpu().double(), grads=grads, state=state_of(ms, cs))

    single = one_step(E.Engine(DEV, dtype), 0, 1)
    vc = E.VirtualComm(world)
    engines = [E.Engine(DEV, dtype) for _ in range(world)]
    for r, e in enumerate(engines):
        e.init_comm_virtual(vc, r, world)
        assert e.comm_info() == (r, world, "virtual")
    ranks = _run_ranks(world, lambda r: one_step(engines[r], r, world))
    tg, tl, tsn = (2e-5, 1e-5, 1e-5) if dtype == "fp32" else (0.6, 1e-2, 0.1)
    if world == 8 and dtype == "fp32":
        # 2 + 3 images per rank: the 8-term rank-ordered BatchNorm sums differ from the single pass in the last bit, one ReLU /
        # arg-max decision on these 2x2 top-level maps flips, and every layer below it sees that as ~1e-3 (measured 7e-4 .. 1.1e-3,
        # flat across conv1 .. layer4, losses equal to 1e-5); worlds 2 and 4 hold the arithmetic to 2e-5
        tg, tsn = 3e-3, 1e-3
    total = sum(o["losses"] for o in ranks)
    for o in ranks:
        assert torch.allclose(o["summed"], total, rtol=1e-6, atol=1e-7), (o["summed"], total)
    if dtype == "fp32":
        assert torch.allclose(total[:3], single["losses"][:3], rtol=tl, atol=1e-7), (total, single["losses"])
    else:
        held(f"vranks/{workload}/w{world}/losses/{dtype}", float(((total[:3] - single["losses"][:3]).abs() / (single["losses"][:3].abs() + 1e-7)).max()),
             tl, floor=5e-3)      # (bf16: the W-rank run and the 1-rank run round differently ordered sums into bf16 activations; the
                                  #  difference moved 5e-4 ... 4.2e-3 over this project's changes of summation order)
    assert abs(float(total[3] - single["losses"][3])) <= (0 if dtype == "fp32" else 2)           # correct-prediction counts add up
    worst = worst_state = 0.0
    prof = [float((a - b).norm() / (b.norm() + 1e-30)) for a, b in zip(ranks[0]["grads"], single["grads"])]
    print(f"[{dtype}] {workload} world {world}: losses {total.tolist()} vs {single['losses'].tolist()}\n   rank-0 gradient deviation by parameter: " +
          " ".join(f"{e:.1e}" for e in prof))
    for r, o in enumerate(ranks):
        for i, (a, b) in enumerate(zip(o["grads"], single["grads"])):
            e = float((a - b).norm() / (b.norm() + 1e-30))
            worst = max(worst, e)
            assert e <= tg, (r, i, e)
        for k, v in single["state"].items():
            if "num_batches" in k:
                assert int(o["state"][k]) == int(v), k
            else:
                es = float((o["state"][k].double() - v.double()).norm() / (v.double().norm() + 1e-30))
                worst_state = max(worst_state, es)
                assert es <= tsn, (r, k)
    if dtype != "fp32":
        # bf16: 2 x the measured worst per-parameter deviation (the ceilings 0.6 / 0.1 above only catch garbage)
        held(f"vranks/{workload}/w{world}/grad_worst/{dtype}", worst, tg, floor=5e-2)
        held(f"vranks/{workload}/w{world}/state_worst/{dtype}", worst_state, tsn, floor=5e-3)
    print(f"[{dtype}] {workload} world {world}: worst per-parameter gradient deviation from the single-device step {worst:.2e}")
    for e in engines:
        del e


def test_virtual_ranks_at_the_headline_per_rank_shape():
    """The sharded code path with REAL launch durations: two virtual ranks, each with the benchmark's per-GPU batch (192 labeled +
    448 strong-unlabeled student images, 448 weak-unlabeled teacher images of 256x256: BASELINE config 4 at world 2), fp32 mode,
    against the single-device step on the concatenated 2176-patch batch.  The tiny-shape tests above prove the arithmetic; here
    the five gradient buckets are hundreds of microseconds of all-reduce queued on the side stream while backward is still
    producing the next ones, the synced-BatchNorm sums of 20 layers cross between two contexts whose kernels take 100+ us each,
    and the optimizer has to wait for the last bucket -- the event / stream ordering is what is being exercised.  Both ranks must
    end with bit-identical gradients that match the single-device ones to fp32 summation order, losses to 1e-5."""
    from ssl_cr_histo_amd import engine as E
    dtype, world, hw, b, mu = "fp32", 2, 256, 64, 7
    nx, nu = 3 * b * world, mu * b * world
    x, u_w, u_s = C.u8(8400, (nx, 3, hw, hw)), C.u8(8401, (nu, 3, hw, hw)), C.u8(8402, (nu, 3, hw, hw))
    y = C.f32(8403, (nx,))

    def one_step(eng, r, w):
        lo_x, hi_x = r * nx // w, (r + 1) * nx // w
        lo_u, hi_u = r * nu // w, (r + 1) * nu // w
        ms, cs = build("finetune", "finetune", 1, True)
        mt, ct = build("finetune", "finetune", 1, True)
        ms.train(); cs.train()
        freeze(mt, 64)
        mt.eval(); ct.eval()
        te, st = eng.bind(mt, ct), eng.bind(ms, cs)
        opt = torch.optim.Adam(list(ms.parameters()) + list(cs.parameters()), lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-4)
        res = eng.step_ssl_cr(te, st, "mse", x[lo_x:hi_x], y[lo_x:hi_x], u_w[lo_u:hi_u], u_s[lo_u:hi_u], 1.0, nx_global=nx, nu_global=nu)
        grads = [st.grad(i).cpu().double() for i in range(len(st.params))]
        st.optimizer_step(opt)
        summed = eng.all_reduce_sum(res["losses"].clone())
        torch.cuda.current_stream().synchronize()
        out = dict(losses=res["losses"].cpu().double(), summed=summed.cpu().double(), grads=grads,
                   bn=ms.state_dict()["model.layer3.1.bn2.running_var"].cpu().double(), w=ms.state_dict()["model.layer2.0.conv1.weight"].cpu().double())
        del te, st
        return out

    single = one_step(E.Engine(DEV, dtype), 0, 1)
    torch.cuda.empty_cache()
    vc = E.VirtualComm(world)
    engines = [E.Engine(DEV, dtype) for _ in range(world)]
    for r, e in enumerate(engines):
        e.init_comm_virtual(vc, r, world)
    ranks = _run_ranks(world, lambda r: one_step(engines[r], r, world))
    total = sum(o["losses"] for o in ranks)
    assert torch.allclose(total[:3], single["losses"][:3], rtol=1e-5, atol=1e-7), (total, single["losses"])
    # fp32 sums over 1280 images x up to 16384 pixels in another order: the early-layer gradients are small residuals of
    # cancelling terms (the reference's own fp32 .grad sits 3.5e-3..4.8e-3 from the float64 gradient there, bpq_cr_full golden), so
    # the bound is 1e-2 per parameter (measured: 4e-3 at conv1 falling to 1e-5 at the heads) -- an ordering bug (a bucket reduced
    # before its last wgrad, a BatchNorm sum read before its peer wrote it) is an error of order one
    worst = 0.0
    prof = [float((a - bb).norm() / (bb.norm() + 1e-30)) for a, bb in zip(ranks[0]["grads"], single["grads"])]
    print("[fp32] headline shape, world 2: rank-0 gradient deviation from the single-device step by parameter: " + " ".join(f"{e:.1e}" for e in prof))
    for o in ranks:
        assert torch.allclose(o["summed"], total, rtol=1e-6, atol=1e-7)
        for i, (a, bb) in enumerate(zip(o["grads"], single["grads"])):
            e = float((a - bb).norm() / (bb.norm() + 1e-30))
            worst = max(worst, e)
            assert e <= 1e-2, (i, e)
        assert float((o["bn"] - single["bn"]).norm() / single["bn"].norm()) <= 1e-4
        # (first Adam step: every weight moves by ~lr whatever its gradient's size, so gradient noise shows as ~lr / |w|)
        assert float((o["w"] - single["w"]).norm() / single["w"].norm()) <= 1e-3
    for a, bb in zip(ranks[0]["grads"], ranks[1]["grads"]):
        assert torch.equal(a, bb)                                  # both ranks hold the same all-reduced bits
    print(f"[fp32] headline shape, world 2: worst per-parameter gradient deviation {worst:.2e}")
    del engines
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("workload", ["ssl_cr", "rsp"])
def test_wgrad_side_stream_is_the_same_step(workload, dtype):
    """sslcr_set_wgrad_stream: the weight-gradient launches of backward on a second stream (event-ordered behind their inputs, the
    scratch buffers' next writers ordered behind them, joined before the optimizer) must give the gradients of the in-line step:
    every parameter, three steps in a row on different batches with the parameters held fixed (the second and third steps
    exercise the cross-step ordering of the scratch buffers; without an update in between, step k's gradients are comparable
    to 1e-5 -- with updates a last-bit difference in the fp32 atomics' order flips ReLU masks on this tiny problem)."""
    eng = _engine(dtype)
    hw, nx, nu = 64, 8, 12
    out = []
    try:
        for on in (False, True):
            eng.set_wgrad_stream(on)
            if workload == "rsp":
                ms, cs = build("triplet", "mlp", 6, False)
                xs = [C.u8(7201 + j, (nx, 3, hw, hw)) for j in range(3)]
                y = C.ints(7210, (nx,), 6)
            else:
                mt, ct = build("finetune", "finetune", 1, True)
                ms, cs = build("finetune", "finetune", 1, True)
                freeze(mt, 64)
                mt.eval()
                te = eng.bind(mt, ct)
                x, u_w, u_s = C.u8(7221, (nx, 3, hw, hw)), C.u8(7222, (nu, 3, hw, hw)), C.u8(7223, (nu, 3, hw, hw))
                y = C.f32(7224, (nx,))
            ms.train()
            st = eng.bind(ms, cs)
            grads = []
            for k in range(3):
                if workload == "rsp":
                    eng.step_supervised(st, "ce", [torch.roll(v, k, 0) for v in xs], y, train=True)
                else:
                    eng.step_ssl_cr(te, st, "mse", torch.roll(x, k, 0), y, u_w, torch.roll(u_s, k, 0), 0.7)
                grads.append([st.grad(i).cpu() for i in range(len(st.params))])
            torch.cuda.synchronize()
            out.append(grads)
    finally:
        eng.set_wgrad_stream(False)
    for ga, gb in zip(out[0], out[1]):
        for i, (a, b) in enumerate(zip(ga, gb)):
            assert rel_err(b, a) < (1e-5 if dtype == "fp32" else 1e-3), i          # (fp32 atomics / slab folds in another order)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("workload", ["ssl_cr", "rsp"])
def test_fused_stem_backward_is_the_same_step(workload, dtype, monkeypatch):
    """The engine's stem backward (sslcr_stem_wgrad_pool: conv1's weight gradient straight from the pooled gradient, bn0's
    backward apply pass on the tile in LDS -- role-split kernel in bf16 mode, single-role in fp32) against the two-kernel path it
    replaced (SSLCR_FUSE_STEM_BWD=0: sslcr_bn_bwd_apply writes the un-pooled gradient, sslcr_stem_wgrad reads it): every gradient
    of the step, conv1 and bn1 in particular.  Two engines of one process, same batch, parameters held fixed."""
    from ssl_cr_histo_amd import engine as E
    hw, nx, nu = 64, 8, 12
    out = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("SSLCR_FUSE_STEM_BWD", fuse)
        eng = E.Engine(DEV, dtype)
        if workload == "rsp":
            ms, cs = build("triplet", "mlp", 6, False)
            xs = [C.u8(7301 + j, (nx, 3, hw, hw)) for j in range(3)]
            y = C.ints(7310, (nx,), 6)
            ms.train()
            st = eng.bind(ms, cs)
            eng.step_supervised(st, "ce", xs, y, train=True)
        else:
            mt, ct = build("finetune", "finetune", 1, True)
            ms, cs = build("finetune", "finetune", 1, True)
            freeze(mt, 64)
            mt.eval()
            te = eng.bind(mt, ct)
            x, u_w, u_s = C.u8(7321, (nx, 3, hw, hw)), C.u8(7322, (nu, 3, hw, hw)), C.u8(7323, (nu, 3, hw, hw))
            ms.train()
            st = eng.bind(ms, cs)
            eng.step_ssl_cr(te, st, "mse", x, C.f32(7324, (nx,)), u_w, u_s, 0.7)
        torch.cuda.synchronize()
        out.append([st.grad(i).cpu() for i in range(len(st.params))])
        del st, eng
    names = [k for k, _ in ms.named_parameters()]
    assert names[0].endswith("conv1.weight") and "bn1" in names[1]
    for i, (a, b) in enumerate(zip(out[0], out[1])):
        # the same dY bits feed the same MFMAs; what differs is the order of the fp32 folds / atomics into conv1.weight
        assert rel_err(b, a) <= (2e-6 if i > 2 else 1e-5), (i, names[i] if i < len(names) else "head", rel_err(b, a))


# ------------------------------------------------------------------------------------------------ f3: slide-sized WSI map
@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp8"])
def test_cam_wsi_slide_sized_map_vs_reference(dtype):
    """test_Camelyon16.test() (test_Camelyon16.py:30-70) on a 26 x 19 tissue mask: 220 tiles of 128x128 in batches of 32 with a ragged
    tail, eval forward + the softmax 'tumor' column (a HIP kernel, sslcr_softmax_col) scattered into the mask-shaped map, against the
    map the reference's own test() produced (probabilities spread over 0.23..0.33)."""
    from ssl_cr_histo_amd.scripts import test_Camelyon16 as script
    _engine(dtype)
    name = "cam_wsi_large"
    g = load_golden(name)
    model, cls = build("finetune", "finetune", 2, True)
    with torch.no_grad():
        cls.classifier[0].weight.mul_(C.CASES[name]["head_scale"])
        cls.classifier[0].bias.mul_(C.CASES[name]["head_scale"])
    loader = C.wsi_loader(name, 6500)
    pm = script.test(ns(), model, cls, loader)
    want = g[f"{name}/ret"]
    assert pm.shape == want.shape and pm.dtype == np.float64
    assert np.array_equal(pm == 0, ~g[f"{name}/mask"])
    err = float(np.abs(pm - want).max())
    print(f"[{dtype}] slide-sized probability map: max abs error {err:.2e} (values 0.23..0.33)")
    assert err <= {"fp32": 1e-4, "bf16": 2e-2, "fp8": 3e-2}[dtype], err


@pytest.mark.parametrize("script", ["bpq_cr", "rsp"])
def test_device_prefetch_of_host_batches_is_the_same_epoch(script):
    """steps._ahead: with ``args.device_prefetch`` the next batch's images go host -> device on a copy stream while the current
    step runs (event-ordered, record_stream'ed).  Three iterations from pinned host loaders must give the same results with and
    without it (same kernels, same inputs; only when the copy happens differs): equal losses and features, post-step state equal to
    the order of the weight gradients' fp32 atomics -- for a two-loader SSL_CR epoch and the four-tensor RSP epoch."""
    from ssl_cr_histo_amd import steps
    _engine("bf16")
    hw, b, mu = 64, 2, 3
    outs = []
    for pre in (False, True):
        a = ns(lambda_u=0.7, tile_h=hw, tile_w=hw, device_prefetch=pre)
        if script == "bpq_cr":
            lab = [(C.u8(9400 + i, (b, 3, 3, 256, 256)).pin_memory(), C.f32(9410 + i, (b, 3))) for i in range(3)]
            unl = [(C.u8(9420 + i, (b * mu, 3, 256, 256)).pin_memory(), C.u8(9430 + i, (b * mu, 3, 256, 256)).pin_memory()) for i in range(3)]
            mt, ct = build("finetune", "finetune", 1, True)
            ms, cs = build("finetune", "finetune", 1, True)
            freeze(mt, 64)
            opt = torch.optim.Adam(list(ms.parameters()) + list(cs.parameters()), lr=1e-4, weight_decay=1e-4)
            r = steps.bpq_cr_train(a, mt, ms, ct, cs, lab, unl, opt, 1)
            outs.append((r[0], r[1], r[2], r[3].cpu(), state_of(ms, cs)))
        else:
            batches = [tuple(C.u8(9440 + 4 * i + j, (b, 3, hw, hw)).pin_memory() for j in range(3)) + (C.ints(9460 + i, (b,), 6),) for i in range(3)]
            model, cls = build("triplet", "mlp", 6, False)
            opt = torch.optim.SGD(list(model.parameters()) + list(cls.parameters()), lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)
            r = steps.rsp_train(a, model, cls, batches, torch.nn.CrossEntropyLoss(), opt, 1)
            outs.append((r[0], r[1], 0.0, r[2].cpu(), state_of(model, cls)))
    x, y = outs
    assert x[0] == y[0] and x[1] == y[1] and x[2] == y[2]
    assert torch.equal(x[3], y[3])
    for k in x[4]:          # (weight gradients sum fp32 atomics in launch-dependent order: the state agrees to that, not to the bit)
        assert torch.allclose(x[4][k].float(), y[4][k].float(), rtol=1e-4, atol=1e-6), k


def test_bench_line_contract():
    """bench.py's ONE JSON line on a short run of the headline workload: the driver's fields, the roofline object with counters
    measured in this run (rocprofv3 is part of the image), and agreement between the HIP-event duration of the dominant kernel and
    the algorithmic work it books."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "2", "--no-also", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1088 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 2500.0 and r["unit"] == "TFLOP/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_gflop_per_launch"] / r["avg_launch_us"] * 1e3) <= 0.01 * r["achieved"]   # GFLOP / us = PFLOP/s
    assert 0.2 < r["frac"] < 1.0
    assert d["pmc_status"]["measured_in_run"] is True, d["pmc_status"]
    assert r["pmc"]["measured_in_run"] is True and 0.2 < r["pmc"]["mfma_busy"] < 1.0 and r["traffic"] > 1e8
    assert 20.0 < d["mfma_util_pct"] < 100.0 and 30.0 < d["hbm_traffic_gb_per_step"] < 120.0
    # what the headline number is held to: measured in bf16, the north star's 1e-3 is the fp32 mode's
    par = d["parity"]
    assert par["mode"] == "bf16" and par["north_star_1e-3_mode"] == "fp32"
    assert 0.0 < par["loss_rel_err_vs_reference"] < 6e-2 and "fp32_images_per_s" in par
    # every leg that only rank 0 runs sits in rank0_only_legs(), steps nothing and calls nothing collective; the legs that run the
    # workload again are child processes guarded by world == 1 (a rank-0-only step of a sharded job hangs all ranks: r04)
    import ast
    src = open(os.path.join(root, "bench.py")).read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "rank0_only_legs")
    body = ast.get_source_segment(src, fn)
    for banned in ("step(", "barrier(", "all_reduce", "eng."):
        assert banned not in body.replace("pmc_step", ""), banned
    for call in ("pmc_in_run(args)", "cpu_baseline(args)", "also_records(args)"):
        line = next(l for l in body.splitlines() if call in l)
        k = body[:body.index(line)].rfind("\n    if ") if line.startswith("        ") else -1
        region = body[k:body.index(line)] if k >= 0 else line
        assert "world == 1" in region, (call, region[-200:])


def test_bench_virtual_ranks_control_flow():
    """bench.py --virtual-ranks 2: the multi-rank control flow of the benchmark itself -- workload per rank, timed region, max over
    ranks, roofline leg on EVERY rank, rank-0-only legs, final barrier -- on one GPU, two engine contexts exchanging through the
    virtual communicator.  A leg that steps on rank 0 only (every --gpus N > 1 run before round 4 had one) hangs here as it would
    under RCCL: the run must finish inside its timeout with ONE line that saw both ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--virtual-ranks", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-also", "--no-pmc"], capture_output=True, text=True, timeout=240, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["ranks_seen"] == 2 and d["collective_transport"] == "virtual" and d["virtual_ranks"] == 2 and d["n_gpus"] == 1
    assert d["config"]["global_batch_patches"] == 2 * 1088 and d["config"]["bn_sync"] is True
    assert abs(d["value"] - 2 * 1088 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    assert "roofline" in d and d["roofline"]["launches"] > 0           # the roofline leg ran (on both ranks) and rank 0 read its table


def test_triplet_branches_as_segments_equal_pass_by_pass():
    """bf16 RSP step: the branches as segments of one launch per layer (default) against the same step pass by pass
    (SSLCR_SEGMENTS=0).  The conv outputs are bit-identical per image; what differs is the order in which BatchNorm partial sums
    are added (a workgroup walks other tiles), i.e. fp32 rounding of the statistics -- losses, running statistics and gradients
    must agree to that level."""
    import subprocess
    import sys
    code = r"""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import cases as C
from ssl_cr_histo_amd import engine as E
import test_engine_gpu as T
eng = E.set_engine(E.Engine("cuda:0", "bf16"))
model, cls = T.build("triplet", "mlp", 6, False)
net = eng.bind(model, cls)
model.train(); cls.train()
xs = [C.u8(31 + i, (12, 3, 256, 256)) for i in range(3)]
y = C.ints(34, (12,), 6)
r = eng.step_supervised(net, "ce", xs, y.long(), train=True)
torch.cuda.synchronize()
names = [k for k, _ in list(model.named_parameters()) + list(cls.named_parameters())]
gr = [net.grad(i).double() for i in range(len(names))]
print("SEG", int(net.segments_used))
print("LOSS", f"{float(r['losses'][0]):.9e}")
print("GN", " ".join(f"{float(g.norm()):.9e}" for g in gr))
print("GP", " ".join(f"{float(g.flatten()[::7].sum()):.9e}" for g in gr))
"""
    outs = {}
    for flag in ("1", "0"):
        env = dict(os.environ)
        env["SSLCR_SEGMENTS"] = flag
        p = subprocess.run([sys.executable, "-c", code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        d = {l.split()[0]: [float(v) for v in l.split()[1:]] for l in p.stdout.splitlines() if l.split() and l.split()[0] in ("SEG", "LOSS", "GN", "GP")}
        outs[flag] = d
    assert outs["1"]["SEG"] == [1.0] and outs["0"]["SEG"] == [0.0]
    assert abs(outs["1"]["LOSS"][0] - outs["0"]["LOSS"][0]) <= 2e-3 * abs(outs["0"]["LOSS"][0])
    # a statistics sum that rounds differently moves a BatchNorm scale by ~1e-7, which flips bf16 roundings downstream: the two
    # runs differ like two bf16 runs do (measured: <= 2.3 % on the smallest gradient norms, 1e-3 on the large ones)
    top = max(outs["0"]["GN"])
    for a, b in zip(outs["1"]["GN"], outs["0"]["GN"]):
        assert abs(a - b) <= 6e-2 * max(abs(b), 1e-3 * top), (a, b)
