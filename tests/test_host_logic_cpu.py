"""CPU-only: the host-side mirror of the reference interface (module trees, parameter order, state_dict keys, meters,
freeze-by-index, sharding helpers)."""
import copy
import types

import torch

from oracle import model as OM


def test_parameter_order_and_state_dict_keys_match_reference():
    from ssl_cr_histo_amd import net
    m = net.TripletNet_Finetune("resnet18")
    specs = OM.net_param_specs()
    named = list(m.named_parameters())
    assert len(named) == 64 == len(specs)
    for (k, p), (sk, shape, _) in zip(named, specs):
        assert k == sk and tuple(p.shape) == tuple(shape), (k, sk)
    # index landmarks quoted by the reference's --modules help (eval_Kather_SSL.py:229): layer1 3, layer2 15, layer3 30, layer4 45, fc 60
    idx = {k: i for i, (k, _) in enumerate(named)}
    assert idx["model.layer1.0.conv1.weight"] == 3 and idx["model.layer2.0.conv1.weight"] == 15
    assert idx["model.layer3.0.conv1.weight"] == 30 and idx["model.layer4.0.conv1.weight"] == 45 and idx["fc.0.weight"] == 60
    assert list(m.state_dict().keys()) == list(OM.init_state(0, specs).keys())
    t = net.TripletNet("resnet18")
    assert [k for k, _ in t.named_parameters()] == [k for k, _ in named]
    assert len(m.model.bn_modules()) == 20 and [b.num_features for b in m.model.bn_modules()] == [c for _, c in OM.bn_names()]
    for kind, n in (("finetune", 9), ("mlp", 6)):
        c = net.FinetuneResNet(n) if kind == "finetune" else net.Classifier(768, n)
        sp = OM.classifier_param_specs(kind, n)
        assert [(k, tuple(p.shape)) for k, p in c.named_parameters()] == [(k, tuple(s)) for k, s, _ in sp]


def test_load_reference_style_checkpoint_and_deepcopy():
    from ssl_cr_histo_amd import net
    m = net.TripletNet_Finetune("resnet18")
    sd = OM.init_state(3, OM.net_param_specs(), random_running_stats=True)
    wrapped = {"module." + k: v for k, v in sd.items()}        # the reference saves DataParallel-wrapped modules
    m.load_state_dict(net.strip_module_prefix(wrapped))
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k])
    for i, (_, p) in enumerate(m.named_parameters()):
        p.requires_grad = i >= 60                               # freeze-by-index, eval_BreastPathQ_SSL_CR.py:433-441
    t = copy.deepcopy(m)                                        # teacher refresh keeps values AND requires_grad flags
    assert [p.requires_grad for p in t.parameters()] == [p.requires_grad for p in m.parameters()]
    assert all(torch.equal(a, b) for a, b in zip(t.state_dict().values(), m.state_dict().values()))
    try:
        net.TripletNet("resnet50")
        assert False
    except NotImplementedError:
        pass


def test_average_meter_and_device_meters():
    from ssl_cr_histo_amd.steps import _Meters
    from ssl_cr_histo_amd.util import AverageMeter
    a = AverageMeter()
    a.update(2.0, 3)
    a.update(4.0, 1)
    assert a.val == 4.0 and a.count == 4 and abs(a.avg - 2.5) < 1e-12
    m = _Meters(["loss", "loss_x", "loss_u", "acc"])
    m.add(torch.tensor([3.0, 1.0, 2.0, 2.0]), 4)                # losses + #correct
    m.add(torch.tensor([1.0, 0.5, 0.5, 6.0]), 6)
    out = m.meters()
    assert abs(out["loss"].avg - (3 * 4 + 1 * 6) / 10) < 1e-6
    assert abs(out["acc"].avg - (0.5 * 4 + 1.0 * 6) / 10) < 1e-6


def test_shard_ranges_cover_the_batch():
    from ssl_cr_histo_amd.dist import shard_range
    for n in (1, 7, 64, 65, 512):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_script_modules_expose_reference_names():
    import importlib
    for mod in ("eval_BreastPathQ_SSL_CR", "eval_Camelyon_SSL_CR", "eval_Kather_SSL_CR", "pretrain_BreastPathQ",
                "pretrain_Camelyon16", "pretrain_RSP", "eval_Camelyon_SSL", "eval_BreastPathQ_SSL", "eval_Kather_SSL"):
        m = importlib.import_module("ssl_cr_histo_amd.scripts." + mod)
        assert callable(m.train) and callable(m.validate)
    # the modules are synthesised from one table (no file per script): `from ... import` forms work as well, unknown names do not
    from ssl_cr_histo_amd import scripts, steps
    from ssl_cr_histo_amd.scripts import test_Camelyon16 as t
    from ssl_cr_histo_amd.scripts.eval_Kather_SSL import train as ktrain
    assert t.test is steps.camelyon16_test and ktrain is steps.kather_sup_train
    assert set(scripts.SCRIPTS) == set(scripts.__all__) and all(hasattr(steps, f) for d in scripts.SCRIPTS.values() for f in d.values())
    import pytest
    with pytest.raises(ImportError):
        importlib.import_module("ssl_cr_histo_amd.scripts.eval_Nothing")
    import inspect
    from ssl_cr_histo_amd import steps
    assert list(inspect.signature(steps.bpq_cr_train).parameters) == [
        "args", "model_teacher", "model_student", "classifier_teacher", "classifier_student", "labeled_train_loader",
        "unlabeled_train_loader", "optimizer", "epoch"]
    assert list(inspect.signature(steps.rsp_train).parameters) == ["args", "model", "classifier", "train_loader", "criterion",
                                                                   "optimizer", "epoch"]
    # the synthesised modules carry a location (tracebacks, inspect), survive a reload without a second finder, and are listed
    import sys
    m = importlib.import_module("ssl_cr_histo_amd.scripts.eval_Kather_SSL")
    assert m.__spec__.origin == scripts.__file__ and m.__file__ == scripts.__file__ and callable(m.teacher_refresh)
    n_finders = len(sys.meta_path)
    assert importlib.reload(m).train is steps.kather_sup_train and len(sys.meta_path) == n_finders
    assert "eval_Kather_SSL" in dir(scripts)


def test_only_the_default_cross_entropy_criterion_is_accepted():
    """the reference calls criterion(output, target) with nn.CrossEntropyLoss(); the engine computes plain mean CE, so a criterion
    that carries class weights, label smoothing, a non-default ignore_index or reduction must raise instead of being ignored."""
    import pytest
    from ssl_cr_histo_amd import steps
    steps._plain_ce(None, "x")
    steps._plain_ce(torch.nn.CrossEntropyLoss(), "x")
    for bad in (torch.nn.CrossEntropyLoss(weight=torch.ones(6)), torch.nn.CrossEntropyLoss(label_smoothing=0.1),
                torch.nn.CrossEntropyLoss(reduction="sum"), torch.nn.CrossEntropyLoss(ignore_index=3), torch.nn.MSELoss()):
        with pytest.raises(NotImplementedError):
            steps._plain_ce(bad, "x")
        with pytest.raises(NotImplementedError):
            steps.kather_sup_validate(None, None, None, [], bad, 1)
        with pytest.raises(NotImplementedError):
            steps.rsp_validate(None, None, None, [], bad, 1)


def test_weak_augment_params_follow_the_reference_draw_order():
    """'next' row f1 host side: per sample flip = rand(1) < p, then randint(top), randint(left); no crop draws when the
    source already has the target size (RandomCrop.get_params).  Checked against the oracle's restatement."""
    import pytest
    from oracle import augment_ref as AR
    from ssl_cr_histo_amd import augment as A
    for n, hw, size in ((7, (300, 280), 256), (4, (64, 64), 64), (3, (40, 52), (30, 32))):
        g1, g2 = torch.Generator().manual_seed(11), torch.Generator().manual_seed(11)
        p1, p2 = A.weak_params(n, hw, size, g1), AR.draw_params(n, hw, size, g2)
        assert p1.dtype == torch.int32 and torch.equal(p1, p2)
        th, tw = (size, size) if isinstance(size, int) else size
        assert int(p1[:, 1].max()) <= hw[0] - th and int(p1[:, 2].max()) <= hw[1] - tw and int(p1.min()) >= 0
        # the generators are left in the same state: the NEXT draw agrees too
        assert torch.equal(torch.rand(3, generator=g1), torch.rand(3, generator=g2))
    with pytest.raises(ValueError):
        A.weak_params(1, (32, 32), 64)


def test_bench_spawns_ranks_itself_and_refuses_more_gpus_than_visible(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-executes itself through torch.distributed.run (the driver's form for N = 1
    is the bare script); on a box with fewer GPUs it must stop with a clear message, not hang in a rendezvous."""
    import importlib
    import subprocess
    import sys
    import pytest
    import types
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    args = types.SimpleNamespace(gpus=8)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(args)
    assert "only 1 GPU(s) visible" in str(e.value)
    # with enough GPUs: one torch.distributed.run child with --nproc-per-node N, loopback rendezvous, the original arguments
    seen = {}
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3"])

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=0)
    monkeypatch.setattr(subprocess, "run", fake_run)
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(args)
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_colour_op_restatements_and_draw_orders():
    """row f4 oracle (no GPU): the stain round trip rgb2hed -> hed2rgb of the restated scikit-image 0.15.0 functions is the identity
    up to the final truncation, a zero shift therefore changes no byte by more than one LSB, the product's parameter helpers draw
    exactly like the oracle's, and the brightness / contrast restatement clips at the image's own maximum."""
    import random

    import numpy as np
    from oracle import augment_ref as AR
    from ssl_cr_histo_amd import augment as A
    img = np.random.RandomState(1).randint(0, 256, (32, 24, 3), dtype=np.uint8)
    back = AR.hed2rgb(AR.rgb2hed(img))
    assert np.abs(back - img / 255.0).max() < 1e-12
    same = AR.colour_augmentation(img, 0.0, 0.0, 0.0)
    assert np.abs(same.astype(int) - img.astype(int)).max() <= 1
    assert np.allclose(np.dot(AR.HED_FROM_RGB, AR.RGB_FROM_HED), np.eye(3), atol=1e-12)
    inv, fwd = A._hed_matrices()
    assert inv == AR.HED_FROM_RGB.reshape(-1).tolist() and fwd == AR.RGB_FROM_HED.reshape(-1).tolist()
    ra, rb = random.Random(4), random.Random(4)
    assert [A.colour_shifts(ra) for _ in range(5)] == [AR.draw_colour_shifts(rb) for _ in range(5)]
    assert [A.brightness_contrast_params(ra, contrast_limit=0.1) for _ in range(8)] == \
           [AR.draw_brightness_contrast(rb, contrast_limit=0.1) for _ in range(8)]
    # a NEGATIVE limit (RandAugment's val for v < 15, models/randaugment.py:93-103): albumentations 0.1.8 keeps the tuple
    # (-limit, limit) unsorted, so the draw is uniform(|v|, -|v|) = |v| (1 - 2 r) -- the mirror image of the sorted draw
    rc = random.Random(9)
    on, alpha, beta = A.brightness_contrast_params(rc, contrast_limit=-0.1)
    rd = random.Random(9)
    rd.random(); rd.random()
    r = rd.random()
    assert on and abs(alpha - (1.0 + 0.1 * (1 - 2 * r))) < 1e-15 and A.brightness_contrast_params(random.Random(9), contrast_limit=-0.1) == \
        AR.draw_brightness_contrast(random.Random(9), contrast_limit=-0.1)
    dark = (img // 2).astype(np.uint8)
    out = AR.brightness_contrast_adjust(dark, 1.2, 0.2)
    assert out.max() == dark.max() and out.dtype == np.uint8
    assert [p[0] for p in A.RandAugmentDevice.POOL] == ["HSV", "Noise", "Scale_Resize_Crop", "Shift_Scale_Rotate", "Color", "Blur_img",
                                                         "Brightness", "Contrast", "Rotate_Crop"]      # models/randaugment.py:105-117
