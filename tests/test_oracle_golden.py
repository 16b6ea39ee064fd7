"""Pin the CPU oracle (oracle/) against the golden vectors captured from the REFERENCE's own
train()/validate() (tests/golden/make_golden.py).  CPU only."""
import os
import pytest
import numpy as np
import torch

from oracle import cases as C
from oracle import epochs as E
from oracle import model as OM
from oracle import steps as S

from _util import check_snapshot, load_golden, merged, oracle_state, rel_err, snapshot_dict

RT = 2e-4      # oracle and reference run the same torch CPU ops; slack covers summation order


def _student_teacher(classes, modules):
    pn_s, bn_s, pc_s = oracle_state("finetune", classes, True)
    pn_t, bn_t, pc_t = oracle_state("finetune", classes, True)
    S.apply_freeze(pn_s, modules)
    for v in pc_s.values():
        v.requires_grad_(True)
    return merged(pn_s, pc_s), bn_s, merged(pn_t, pc_t), bn_t


@pytest.mark.parametrize("faithful", [True, False])
@pytest.mark.parametrize("name", ["bpq_cr_f60", "bpq_cr_f0"])
def test_bpq_cr(name, faithful):
    c = C.CASES[name]
    g = load_golden(name)
    ps, bs, pt, bt = _student_teacher(1, c["modules"])
    opt = S.Adam(ps.values(), c["lr"], (0.9, 0.999), 1e-8, c["wd"])
    ret = E.bpq_cr_train(ps, bs, pt, bt, opt, C.labeled_batches(name), C.unlabeled_batches(name),
                         c["lambda_u"], faithful)
    rt = RT if faithful else 2e-3
    for i in range(3):
        assert abs(ret[i] - g[f"{name}/ret"][i]) <= rt * abs(g[f"{name}/ret"][i])
    assert rel_err(ret[3], g[f"{name}/feats"]) < rt
    assert torch.equal(ret[4], torch.from_numpy(g[f"{name}/targets"]))
    check_snapshot(g, name, snapshot_dict(ps, bs), rt)
    val = E.bpq_cr_validate(ps, bs, C.val_batches_reg(name), faithful)
    assert abs(val - g[f"{name}/val"][0]) <= rt * abs(g[f"{name}/val"][0])


@pytest.mark.parametrize("faithful", [True, False])
@pytest.mark.parametrize("name", ["cam_cr_f60", "cam_cr_f0"])
def test_cam_cr(name, faithful):
    c = C.CASES[name]
    g = load_golden(name)
    ps, bs, pt, bt = _student_teacher(2, c["modules"])
    opt = S.SGDNesterov(ps.values(), c["lr"], 0.9, c["wd"])
    torch.manual_seed(777)
    ret = E.cam_cr_train(ps, bs, pt, bt, opt, C.labeled_batches_cls(name, 1000, 1),
                         C.labeled_batches_cls(name, 1100, 0), C.unlabeled_batches(name, 2000),
                         C.unlabeled_batches(name, 2100), c["lambda_u"], c["hw"], faithful)
    rt = RT if faithful else 2e-3
    for i in range(4):
        assert abs(ret[i] - g[f"{name}/ret"][i]) <= rt * abs(g[f"{name}/ret"][i]) + 1e-9
    assert rel_err(ret[4], g[f"{name}/feats"]) < rt
    assert torch.equal(ret[5], torch.from_numpy(g[f"{name}/targets"]))
    check_snapshot(g, name, snapshot_dict(ps, bs), rt)
    torch.manual_seed(778)
    val = E.cam_cr_validate(ps, bs, C.val_batches_cls(name, 4000, 1), C.val_batches_cls(name, 4100, 0), faithful)
    assert abs(val[0] - g[f"{name}/val"][0]) <= rt * abs(g[f"{name}/val"][0])
    assert val[1] == g[f"{name}/val"][1]


def test_kather_cr():
    name = "kather_cr_f0"
    c = C.CASES[name]
    g = load_golden(name)
    ps, bs, pt, bt = _student_teacher(9, c["modules"])
    opt = S.Adam(ps.values(), c["lr"], (0.9, 0.999), 1e-8, c["wd"])
    ret = E.kather_cr_train(ps, bs, pt, bt, opt, C.labeled_batches_kather(name), C.unlabeled_batches(name), c["lambda_u"])
    for i in range(3):
        assert abs(ret[i] - g[f"{name}/ret"][i]) <= RT * abs(g[f"{name}/ret"][i])
    assert ret[3] == g[f"{name}/ret"][3]
    check_snapshot(g, name, snapshot_dict(ps, bs), RT)
    val = E.kather_cr_validate(ps, bs, C.val_batches_kather(name))
    assert abs(val[0] - g[f"{name}/val"][0]) <= RT * g[f"{name}/val"][0] and val[1] == g[f"{name}/val"][1]


def test_rsp_and_lookahead():
    name = "rsp"
    c = C.CASES[name]
    g = load_golden(name)
    pn, bn, pc = oracle_state("mlp", 6, False)
    p = merged(pn, pc)
    for v in p.values():
        v.requires_grad_(True)
    opt = S.SGDNesterov(p.values(), c["lr"], 0.9, c["wd"])
    la = S.Lookahead(opt, 5, 0.5)
    ret = E.rsp_epoch(p, bn, opt, C.rsp_batches(name), c["hw"], True)
    assert abs(ret[0] - g["rsp/ret"][0]) <= RT * g["rsp/ret"][0]
    assert ret[1] == g["rsp/ret"][1]
    assert rel_err(ret[2], g["rsp/feats"]) < RT
    assert torch.equal(ret[3], torch.from_numpy(g["rsp/targets"]))
    check_snapshot(g, "rsp", snapshot_dict(p, bn), RT)
    val = E.rsp_epoch(p, bn, None, C.rsp_batches(name, 3500), c["hw"], False)
    assert abs(val[0] - g["rsp/val"][0]) <= RT * g["rsp/val"][0]
    assert val[1] == g["rsp/val"][1]
    for _ in range(5):          # pretrain_BreastPathQ.py:293 -- Lookahead stepped with stale grads
        la.step()
    check_snapshot(g, "rsp/la5", snapshot_dict(p, bn), RT)


def test_supervised():
    name = "cam_sup"
    c = C.CASES[name]
    g = load_golden(name)
    pn, bn, pc = oracle_state("finetune", 2, False)
    p = merged(pn, pc)
    for v in p.values():
        v.requires_grad_(True)
    opt = S.SGDNesterov(p.values(), c["lr"], 0.9, c["wd"])
    torch.manual_seed(779)
    ret = E.cam_sup_train(p, bn, opt, C.labeled_batches_cls(name, 1000, 1), C.labeled_batches_cls(name, 1100, 0), c["hw"])
    assert abs(ret[0] - g[f"{name}/ret"][0]) <= RT * g[f"{name}/ret"][0]
    assert ret[1] == g[f"{name}/ret"][1]
    assert rel_err(ret[2], g[f"{name}/feats"]) < RT
    check_snapshot(g, name, snapshot_dict(p, bn), RT)

    name = "bpq_sup"
    c = C.CASES[name]
    g = load_golden(name)
    pn, bn, pc = oracle_state("finetune", 1, False)
    p = merged(pn, pc)
    for v in p.values():
        v.requires_grad_(True)
    opt = S.Adam(p.values(), c["lr"], (0.9, 0.999), 1e-8, c["wd"])
    ret = E.bpq_sup_train(p, bn, opt, C.labeled_batches(name), c["hw"])
    assert abs(ret[0] - g[f"{name}/ret"][0]) <= RT * g[f"{name}/ret"][0]
    assert rel_err(ret[1], g[f"{name}/feats"]) < RT
    check_snapshot(g, name, snapshot_dict(p, bn), RT)


def test_cam_wsi_probability_map():
    """'next' row f3: the oracle's restatement of test_Camelyon16.test() against the map the reference itself produced."""
    name = "cam_wsi"
    g = load_golden(name)
    pn, bn, pc = oracle_state("finetune", 2, True)
    loader = C.wsi_loader(name)
    assert np.array_equal(loader.dataset.mask, g[f"{name}/mask"])
    for faithful in (True, False):
        pm = E.cam_wsi_test(merged(pn, pc), bn, loader, faithful)
        assert pm.shape == g[f"{name}/ret"].shape and pm.dtype == np.float64
        assert np.array_equal(pm == 0, ~g[f"{name}/mask"])                 # untouched where there is no tissue
        assert np.abs(pm - g[f"{name}/ret"]).max() <= 1e-5


def test_cam_wsi_slide_sized_map():
    """the same function on a 26 x 19 mask (220 tissue tiles of 128x128, batches of 32, ragged tail) with a head scaled so that the
    probabilities spread over (0.23, 0.33) instead of saturating."""
    name = "cam_wsi_large"
    g = load_golden(name)
    pn, bn, pc = oracle_state("finetune", 2, True)
    for k in pc:
        pc[k] = pc[k] * C.CASES[name]["head_scale"]
    loader = C.wsi_loader(name, 6500)
    assert np.array_equal(loader.dataset.mask, g[f"{name}/mask"]) and int(loader.dataset.mask.sum()) == 220
    pm = E.cam_wsi_test(merged(pn, pc), bn, loader, False)
    assert np.array_equal(pm == 0, ~g[f"{name}/mask"])
    assert np.abs(pm - g[f"{name}/ret"]).max() <= 1e-5


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_stage_activations(mode):
    g = load_golden("stages")
    pn, bn, _ = oracle_state("finetune", 1, True)
    x = C.u8(5000, (2, 3, 64, 64)).float()
    taps = {}
    with torch.no_grad():
        e = OM.backbone_forward(pn, bn, x, mode == "train", taps=taps)
    for k, v in taps.items():
        assert rel_err(v, g[f"stages/{mode}/{k}"]) < 1e-5, k
    bn2 = {k: v.clone() for k, v in oracle_state("finetune", 1, True)[1].items()}
    with torch.no_grad():
        feats = OM.finetune_forward(pn, bn2, x, mode == "train", faithful=False)
    assert rel_err(feats, g[f"stages/{mode}/feats"]) < 1e-5
    if mode == "train":     # the x3 running-stat replay of TripletNet_Finetune (models/net.py:88-90)
        for k in ("model.bn1.running_mean", "model.bn1.running_var", "model.layer3.0.downsample.1.running_mean",
                  "model.layer4.1.bn2.running_var"):
            assert rel_err(bn2[k], g[f"stages/train/{k}"]) < 1e-5, k
        assert int(bn2["model.bn1.num_batches_tracked"]) == int(g["stages/train/nbt"]) == 3


def test_forward_only_config2_full_size():
    """BASELINE config 2 at its own size: the oracle's TripletNet_Finetune forward on 256 images of 256x256 (eval mode; one backbone
    pass per image) against reductions of the reference module's own output (tests/golden/make_golden.py:gen_fwd_full)."""
    g = load_golden("fwd_full")
    pn, bn, _ = oracle_state("finetune", 1, True)
    x = C.u8(5100, (256, 3, 256, 256)).float()
    with torch.no_grad():
        feats = OM.finetune_forward(pn, bn, x, False, faithful=False).double()
    assert rel_err(feats.norm(dim=1), g["fwd_full/eval/feats_rowl2"]) < 2e-4
    assert rel_err(feats.sum(0), g["fwd_full/eval/feats_colsum"]) < 2e-4
    assert rel_err(feats[:4].float(), g["fwd_full/eval/feats_head"]) < 2e-4


def test_kather_supervised():
    """config 1's script, eval_Kather_SSL.train/validate (the slice of the reference file above ``def parse_args``), 96x96."""
    name = "kather_sup"
    c = C.CASES[name]
    g = load_golden(name)
    pn, bn, pc = oracle_state("finetune", c["classes"], False)
    p = merged(pn, pc)
    for v in p.values():
        v.requires_grad_(True)
    opt = S.Adam(p.values(), c["lr"], (0.9, 0.999), 1e-8, c["wd"])
    ret = E.kather_sup_train(p, bn, opt, C.sup_batches_kather(name), c["hw"])
    assert abs(ret[0] - g[f"{name}/ret"][0]) <= RT * g[f"{name}/ret"][0] and ret[1] == g[f"{name}/ret"][1]
    check_snapshot(g, name, snapshot_dict(p, bn), RT)
    val = E.kather_sup_validate(p, bn, C.val_batches_kather(name))
    assert abs(val[0] - g[f"{name}/val"][0]) <= RT * g[f"{name}/val"][0] and val[1] == g[f"{name}/val"][1]


def _load_into(p, b, sd):
    """the oracle's load_state_dict: values of a (module.-stripped) reference state_dict into the functional dicts."""
    with torch.no_grad():
        for k, v in sd.items():
            k = k[7:] if k.startswith("module.") else k
            if k in p:
                p[k].copy_(v)
            else:
                b[k] = v.clone()


def test_checkpoint_continuation_ssl_cr():
    """row f2: the oracle, started from the checkpoint the REFERENCE wrote after epoch 1 (rebuilt from the fixture), reproduces
    the reference's epoch 2 -- parameters, BatchNorm buffers and Adam moments all come from the file."""
    from _util import rebuild_ckpt
    name = "ckpt_bpq_cr"
    c = C.CASES[name]
    g = load_golden(name)
    sd0 = OM.init_state(C.PARAM_SEED, OM.net_param_specs(), random_running_stats=True)
    cd0 = OM.init_state(C.PARAM_SEED + 1, OM.classifier_param_specs("finetune", 1))
    ck, _ = rebuild_ckpt(name, {"model_student": (sd0, False), "model_teacher": (sd0, False),
                                "classifier_student": (cd0, False), "classifier_teacher": (cd0, False)})
    ps, bs, pt, bt = _student_teacher(1, c["modules"])
    _load_into(ps, bs, ck["model_student"]); _load_into(ps, bs, ck["classifier_student"])
    _load_into(pt, bt, ck["model_teacher"]); _load_into(pt, bt, ck["classifier_teacher"])
    opt = S.Adam(ps.values(), c["lr"], (0.9, 0.999), 1e-8, c["wd"])
    st = ck["optimizer"]["state"]
    assert len(st) == len(opt.params)
    for i in range(len(opt.params)):
        opt.m[i].copy_(st[i]["exp_avg"]); opt.v[i].copy_(st[i]["exp_avg_sq"])
    opt.t = int(st[0]["step"])
    ret = E.bpq_cr_train(ps, bs, pt, bt, opt, C.labeled_batches(name, 1200), C.unlabeled_batches(name, 2200), c["lambda_u"])
    for i in range(3):
        assert abs(ret[i] - g[f"{name}/ret2"][i]) <= RT * abs(g[f"{name}/ret2"][i])
    assert rel_err(ret[3], g[f"{name}/feats2"]) < RT
    check_snapshot(g, name + "/e2", snapshot_dict(ps, bs), RT)


@pytest.mark.parametrize("name", ["traj_bpq_cr", "traj_cam_cr"])
def test_trajectory_yardstick_is_pinned_to_the_reference(name):
    """tests/golden/traj_*_yard.npz (the fp32-tolerance yardstick of the GPU trajectory test): (a) its 8-thread reference run IS the
    trajectory golden; (b) its float64 leg comes from the oracle restatement, so the oracle's fp32 run of the first four iterations
    (same seeds, same shuffles, through the epoch functions) must reproduce the reference's losses and its first validate();
    (c) the float64 leg starts where the fp32 runs start (iteration 1 agrees to fp32 round-off) and the recorded distances of
    the reference's own fp32 runs from float64 are what the GPU test's bounds are built from: they are finite and ordered as
    expected (losses < validate() < state for Adam)."""
    c = C.CASES[name]
    cam = c["script"] == "cam_cr"
    g, y = load_golden(name), load_golden(name + "_yard")
    yn = name + "_yard"
    assert np.array_equal(y[f"{yn}/ret_t8"], g[f"{name}/ret"][:, :3]) and np.array_equal(y[f"{yn}/vals_t8"], g[f"{name}/vals"][:, 0])
    r64 = y[f"{yn}/ret_f64"]
    assert np.abs(y[f"{yn}/ret_t8"][0] / r64[0] - 1).max() < 2e-6
    for t in (8, 3, 1):
        e = np.abs(y[f"{yn}/ret_t{t}"] / r64 - 1).max()
        assert 1e-6 < e < 3e-3, (t, e)                                     # three different fp32 trajectories, all near float64
    assert not np.array_equal(y[f"{yn}/ret_t8"], y[f"{yn}/ret_t1"])
    ps, bs, pt, bt = _student_teacher(c["classes"], c["modules"])
    opt = S.SGDNesterov(ps.values(), c["lr"], 0.9, c["wd"]) if cam else S.Adam(ps.values(), c["lr"], (0.9, 0.999), 1e-8, c["wd"])
    torch.manual_seed(780)
    for it in range(4):
        if cam:
            r = E.cam_cr_train(ps, bs, pt, bt, opt, C.labeled_batches_cls(name, 1000 + 7 * it, 1), C.labeled_batches_cls(name, 1100 + 7 * it, 0),
                               C.unlabeled_batches(name, 2000 + 7 * it), C.unlabeled_batches(name, 2100 + 7 * it), c["lambda_u"], c["hw"])
        else:
            r = E.bpq_cr_train(ps, bs, pt, bt, opt, C.labeled_batches(name, 1000 + 7 * it), C.unlabeled_batches(name, 2000 + 7 * it),
                               c["lambda_u"])
        for i in range(3):
            assert abs(r[i] - g[f"{name}/ret"][it][i]) <= RT * abs(g[f"{name}/ret"][it][i]), (it, i)
    if cam:
        v = E.cam_cr_validate(ps, bs, C.val_batches_cls(name, 4000, 1), C.val_batches_cls(name, 4100, 0))[0]
    else:
        v = E.bpq_cr_validate(ps, bs, C.val_batches_reg(name))
    # validate() after four Adam steps already amplifies fp32 round-off (the reference's own runs at 8 / 3 / 1 threads differ by
    # this much from each other): held to the yardstick's spread, not to RT
    spread = max(abs(y[f"{yn}/vals_t{a}"][0] / y[f"{yn}/vals_t{b}"][0] - 1) for a, b in ((8, 3), (8, 1), (3, 1)))
    assert abs(v / g[f"{name}/vals"][0][0] - 1) <= max(RT, 3 * spread), (v, g[f"{name}/vals"][0][0], spread)


def test_flip_yardstick_names_a_head_unit_fp32_cannot_resolve():
    """tests/golden/*_full_flip.npz (make_golden.py:gen_flip_yard): the head ReLU pre-activations that sit within 1e-5 x rms of zero in
    the float64 run of the full-size iterations, and how much of every parameter's gradient flows through them.  Data checks on all
    three files; for rsp_full the oracle's own fp32 forward must find the named unit inside that margin too (pair 0, sample 54, unit
    384: z = -1.08e-6 at rms 0.52 -- any fp32 summation order decides its sign)."""
    import torch.nn.functional as F
    from collections import OrderedDict
    from oracle import bf16_emul as B
    for name in ("rsp_full", "bpq_cr_full", "cam_cr_full"):
        y = load_golden(name + "_flip")
        g = load_golden(name)
        u = y[f"{name}_flip/units"]
        assert 1 <= u.shape[0] <= 4 and u.shape[1] == 5
        assert np.all(np.abs(u[:, 3]) < 1e-5 * u[:, 4])
        n = len(g[f"{name}/grad_names"])
        fl, fp = y[f"{name}_flip/flip_l2"], y[f"{name}_flip/flip_pr"]
        assert fl.shape == (n,) and fp.shape == (n,) and np.all(fl >= 0) and np.all(fp >= 0)
        names = [str(k) for k in g[f"{name}/grad_names"]]
        for k in ("fc.2.weight", "fc.2.bias", "classifier.0.weight"):       # nothing behind the unit depends on its mask
            if k in names:
                assert fl[names.index(k)] == 0.0
        assert fl[names.index("fc.0.bias")] > 0.0 and fl.max() < 5e-2
    name = "rsp_full"
    u = load_golden(name + "_flip")[f"{name}_flip/units"][0]
    pn, _, pc = oracle_state("mlp", 6, False)
    p = merged(pn, pc)
    (i1, i2, i3, _t), = C.rsp_batches(name)
    pair = ((0, 1), (1, 2), (0, 2))[int(u[0])]
    with torch.no_grad():
        e = {i: B.backbone_train(p, (i1, i2, i3)[i].float(), lambda t: t) for i in pair}
        z = F.linear(torch.cat((e[pair[0]], e[pair[1]]), 1), p["fc.0.weight"], p["fc.0.bias"])
    assert abs(float(z[int(u[1]), int(u[2])])) < 1e-5 * float(z.pow(2).mean().sqrt())


def test_bf16_teacher_emulation_is_the_eval_forward_when_nothing_is_rounded():
    """oracle/bf16_emul.py:backbone_eval (BatchNorm folded into the filters, the engine's teacher path) with q = identity is the
    eval-mode backbone of oracle/model.py, so the only difference of the emulated run is the rounding."""
    from oracle import bf16_emul as B
    p_net, b_net, p_cls = oracle_state("finetune", 1, True)
    p = {k: v.double() for k, v in merged(p_net, p_cls).items()}
    b = {k: v.double() if v.is_floating_point() else v for k, v in b_net.items()}
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    with torch.no_grad():
        want = OM.backbone_forward(p, dict(b), x, False)
        got = B.backbone_eval(p, b, x, lambda t: t)
        lt = B.teacher_logits(p, b, x, False)
        assert rel_err(got, want) < 1e-12
        assert rel_err(lt, OM.classifier_forward(p, OM.finetune_forward(p, dict(b), x, False, True))) < 1e-12
        assert 1e-4 < rel_err(B.backbone_eval({k: v.float() for k, v in p.items()}, {k: v.float() if v.is_floating_point() else v for k, v in b.items()},
                                              x.float(), B.rnd).double(), want) < 3e-2


@pytest.mark.parametrize("name", ["bpq_cr_full", "cam_cr_full"])
def test_bf16_yardstick_is_pinned_to_the_reference(name):
    """tests/golden/bf16_yard.npz (make_bf16_yard.py, oracle only): its float64 run returns the losses the REFERENCE's own
    iteration returned (golden `ret`), so the emulated run next to it measures bf16 storage on that very iteration; the
    emulation covers the teacher too (its error dominates the BreastPathQ consistency loss: 1.4e-3 on the teacher's logits)."""
    y = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_yard.npz"))
    g = load_golden(name)
    assert np.allclose(y[f"{name}/ret_f64"], g[f"{name}/ret"][:3], rtol=2e-6, atol=0)
    assert np.allclose(y[f"{name}/ret_bf16emul"], y[f"{name}/ret_f64"], rtol=5e-3, atol=0)
    assert int(y[f"{name}/argmax_flips"][0]) == 0
    assert 1e-4 < float(y[f"{name}/teacher_logits_err"][0]) < 5e-3 and 1e-3 < float(y[f"{name}/feats_err"][0]) < 2e-2


def test_bf16_yardstick_rsp_and_forward_only_are_pinned_to_the_reference():
    """the float64 runs behind the rsp_full / fwd_full yardsticks reproduce the reference's own loss and feature row norms"""
    y = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_yard.npz"))
    assert abs(float(y["rsp_full/ret_f64"][0]) - load_golden("rsp_full")["rsp_full/ret"][0]) <= 2e-6 * float(y["rsp_full/ret_f64"][0])
    g = load_golden("fwd_full")
    for mode in ("eval", "train"):
        assert rel_err(torch.from_numpy(y[f"fwd_full/{mode}/feats_rowl2_f64"]), g[f"fwd_full/{mode}/feats_rowl2"]) < 2e-6
        assert 1e-3 < float(y[f"fwd_full/{mode}/feats_err"][0]) < 2e-2


def test_small_epoch_bf16_yardstick_is_pinned_to_the_reference():
    """tests/golden/bf16_yard_small.npz (make_bf16_yard_small.py, oracle only): the fp32 leg of every small-epoch case returns the
    losses / validate() values the REFERENCE's own epoch returned (the goldens; 2e-3: the de-triplicated oracle form), so the distance
    of its bf16-storage leg from it is a distance from the reference's numbers; and the emulating() context leaves the oracle as it was."""
    from oracle import bf16_emul as B
    y = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_yard_small.npz"))
    idx = {"ret0": ("ret", 0), "ret1": ("ret", 1), "ret2": ("ret", 2), "val": ("val", 0)}
    n = 0
    for k in y.files:
        if not k.endswith("_fp32"):
            continue
        name, q = k[:-5].split("/")
        g = load_golden(name)
        key, i = idx[q]
        want = float(g[f"{name}/{key}"][i])
        assert abs(float(y[k][0]) - want) <= 2e-3 * abs(want) + 1e-9, (k, float(y[k][0]), want)
        assert 0.0 < float(y[f"{name}/{q}_err"][0]) < 0.2
        n += 1
    assert n >= 25
    f = OM.backbone_forward
    with B.emulating():
        assert OM.backbone_forward is B.backbone_forward_emulated
    assert OM.backbone_forward is f
