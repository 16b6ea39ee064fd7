"""Row f2 on CPU: ssl_cr_histo_amd.checkpoint reads the files the REFERENCE writes and writes files the reference reads.

The fixtures (tests/golden/ckpt_*.npz, made by tests/golden/make_golden.py) hold the reference-written checkpoint of each
layout -- the exact pickle structure (key order, container types, dtypes, shapes, python scalars, the argparse.Namespace) and,
for the frozen-backbone cases, every tensor that moved away from the seeded initial state.  Here the file is rebuilt,
written with torch.save, read back through the product's loader into the product's modules, and the product's writer is
held to the same structure.  The engine-side continuation (epoch 2 equals the reference's) is tests/test_engine_gpu.py."""
import copy

import pytest
import torch

from oracle import cases as C
from oracle import model as OM

from _util import ckpt_tree, load_golden, rebuild_ckpt, strip_values


def _mods(kind_net, kind_cls, classes, rand_stats):
    from ssl_cr_histo_amd import net
    model = net.TripletNet_Finetune("resnet18") if kind_net == "finetune" else net.TripletNet("resnet18")
    cls = net.FinetuneResNet(classes) if kind_cls == "finetune" else net.Classifier(768, classes)
    model.load_state_dict(OM.init_state(C.PARAM_SEED, OM.net_param_specs(), random_running_stats=rand_stats))
    cls.load_state_dict(OM.init_state(C.PARAM_SEED + 1, OM.classifier_param_specs(kind_cls, classes)))
    return model, cls


def _freeze(model, modules):
    for i, (_, p) in enumerate(model.named_parameters()):
        p.requires_grad = i >= modules


def _same_sd(module, sd, prefixed):
    for k, v in module.state_dict().items():
        assert torch.equal(v, sd[("module." if prefixed else "") + k]), k


def test_reads_reference_ssl_cr_file_and_writes_the_same_layout(tmp_path):
    from ssl_cr_histo_amd import checkpoint as CK
    name = "ckpt_bpq_cr"
    c = C.CASES[name]
    sd0 = OM.init_state(C.PARAM_SEED, OM.net_param_specs(), random_running_stats=True)
    cd0 = OM.init_state(C.PARAM_SEED + 1, OM.classifier_param_specs("finetune", 1))
    ref, tree = rebuild_ckpt(name, {"model_student": (sd0, False), "model_teacher": (sd0, False),
                                    "classifier_student": (cd0, False), "classifier_teacher": (cd0, False)})
    f = tmp_path / "fine_CR_trained_model_1.pt"
    torch.save(ref, f)
    ck = CK.load_file(f)
    assert CK.detect_layout(ck) == "ssl_cr" and ckpt_tree(ck) == tree          # what we read IS what the reference wrote
    mt, ct = _mods("finetune", "finetune", 1, False)
    ms, cs = _mods("finetune", "finetune", 1, False)
    _freeze(mt, 64)
    _freeze(ms, c["modules"])
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, list(ms.parameters()) + list(cs.parameters())), lr=c["lr"],
                           betas=(0.9, 0.999), weight_decay=c["wd"])
    start, _ = CK.resume(str(f), opt, model_student=ms, model_teacher=mt, classifier_teacher=ct, classifier_student=cs)
    assert start == 2
    _same_sd(ms, ref["model_student"], False)
    _same_sd(mt, ref["model_teacher"], False)
    _same_sd(cs, ref["classifier_student"], False)
    _same_sd(ct, ref["classifier_teacher"], False)
    assert int(ms.state_dict()["model.bn1.num_batches_tracked"]) == 6          # two train() iterations x3 (models/net.py:88-90)
    got = opt.state_dict()
    assert got["param_groups"] == ref["optimizer"]["param_groups"]
    for i, st in ref["optimizer"]["state"].items():
        for k, v in st.items():
            assert torch.equal(torch.as_tensor(got["state"][i][k]), torch.as_tensor(v)), (i, k)
    # the writer: same pickle structure as the reference's file (values of the logged scalars aside)
    f2 = tmp_path / "ours.pt"
    CK.save_ssl_cr(str(f2), ref["args"], ms, mt, ct, cs, opt, 1, 0.1, 0.2, 0.3)
    assert strip_values(ckpt_tree(CK.load_file(f2))) == strip_values(tree)
    # DataParallel-keyed variant (multi-GPU reference runs) loads into bare modules too
    f3 = tmp_path / "ours_dp.pt"
    CK.save_ssl_cr(str(f3), ref["args"], ms, mt, ct, cs, opt, 1, 0.1, 0.2, 0.3, data_parallel_keys=True)
    ck3 = CK.load_file(f3)
    assert all(k.startswith("module.") for k in ck3["model_student"])
    ms2, cs2 = _mods("finetune", "finetune", 1, False)
    mt2, ct2 = _mods("finetune", "finetune", 1, False)
    CK.resume(str(f3), None, model_student=ms2, model_teacher=mt2, classifier_teacher=ct2, classifier_student=cs2)
    _same_sd(ms2, ref["model_student"], False)


def test_reads_reference_finetune_file_both_ways(tmp_path):
    """the eval_Camelyon_SSL.py file (DataParallel keys): --resume into (wrapped or bare) modules, and the SSL_CR scripts' way
    of consuming it -- teacher and student from 'model' / 'classifier' with the module. prefix stripped."""
    from ssl_cr_histo_amd import checkpoint as CK
    name = "ckpt_cam_sup"
    c = C.CASES[name]
    sd0 = OM.init_state(C.PARAM_SEED, OM.net_param_specs())
    cd0 = OM.init_state(C.PARAM_SEED + 1, OM.classifier_param_specs("finetune", 2))
    ref, tree = rebuild_ckpt(name, {"model": (sd0, True), "classifier": (cd0, True)})
    assert all(k.startswith("module.") for k in ref["model"])
    f = tmp_path / "fine_tuned_model_1.pt"
    torch.save(ref, f)
    ck = CK.load_file(f)
    assert CK.detect_layout(ck) == "finetune" and ckpt_tree(ck) == tree
    ms, cs = _mods("finetune", "finetune", 2, False)
    _freeze(ms, c["modules"])
    opt = torch.optim.SGD(filter(lambda p: p.requires_grad, list(ms.parameters()) + list(cs.parameters())), lr=c["lr"], momentum=0.9,
                          weight_decay=c["wd"], nesterov=True)
    start, _ = CK.resume(str(f), opt, model=torch.nn.DataParallel(ms), classifier=cs)
    assert start == 2
    _same_sd(ms, ref["model"], True)
    _same_sd(cs, ref["classifier"], True)
    for i, st in ref["optimizer"]["state"].items():
        assert torch.equal(opt.state_dict()["state"][i]["momentum_buffer"], st["momentum_buffer"])
    mt, ct = _mods("finetune", "finetune", 2, True)
    ms2, cs2 = _mods("finetune", "finetune", 2, True)
    CK.load_finetuned(str(f), (mt, ms2), (ct, cs2))
    _same_sd(mt, ref["model"], True)
    _same_sd(cs2, ref["classifier"], True)
    f2 = tmp_path / "ours.pt"
    CK.save_finetune(str(f2), ref["args"], ms, cs, opt, 1, 0.5, data_parallel_keys=True, train_acc=0.5, val_acc=0.5, val_loss=0.7)
    assert strip_values(ckpt_tree(CK.load_file(f2))) == strip_values(tree)


def test_writes_reference_pretrain_layout(tmp_path):
    """pretrain_BreastPathQ.py:298-305: 'model' + 'optimizer' (no classifier); the fixture carries the structure of the
    reference-written file.  The written file loads into the fine-tuning net the way eval_Camelyon_SSL.py:319-331 does."""
    from ssl_cr_histo_amd import checkpoint as CK
    from ssl_cr_histo_amd.lookahead import Lookahead
    import json
    name = "ckpt_rsp"
    c = C.CASES[name]
    tree = json.loads(str(load_golden(name)[f"{name}/ckpt_tree"]))
    model, cls = _mods("triplet", "mlp", 6, False)
    opt = torch.optim.SGD(list(model.parameters()) + list(cls.parameters()), lr=c["lr"], momentum=0.9, weight_decay=c["wd"], nesterov=True)
    for p in opt.param_groups[0]["params"]:                      # one optimizer step's worth of state, without an engine
        opt.state[p]["momentum_buffer"] = torch.zeros_like(p)
    import argparse
    args = argparse.Namespace(**{k[1]: node["v"] for k, node in tree["k"][0][1]["v"]["k"]})
    f = tmp_path / "model_1.pt"
    CK.save_pretrain(str(f), args, model, Lookahead(opt, la_steps=5, la_alpha=0.5), 1, 1.8, 0.2, data_parallel_keys=True)
    ck = CK.load_file(f)
    assert CK.detect_layout(ck) == "pretrain" and "classifier" not in ck
    assert strip_values(ckpt_tree(ck)) == strip_values(tree)
    ft, _ = _mods("finetune", "finetune", 2, True)
    with torch.no_grad():
        model.fc[0].bias.add_(1.0)
    CK.save_pretrain(str(f), args, model, opt, 1, 1.8, 0.2, data_parallel_keys=True)
    CK.load_pretrained(ft, str(f))
    _same_sd(ft, CK.load_file(f)["model"], True)
    m2, c2 = _mods("triplet", "mlp", 6, False)
    o2 = torch.optim.SGD(list(m2.parameters()) + list(c2.parameters()), lr=c["lr"], momentum=0.9, weight_decay=c["wd"], nesterov=True)
    start, _ = CK.resume(str(f), o2, model=m2)
    assert start == 2 and torch.equal(m2.fc[0].bias, model.fc[0].bias)


def test_rejects_foreign_files(tmp_path):
    from ssl_cr_histo_amd import checkpoint as CK
    f = tmp_path / "x.pt"
    torch.save({"weights": torch.zeros(3)}, f)
    with pytest.raises(KeyError):
        CK.detect_layout(CK.load_file(f))
    with pytest.raises(FileNotFoundError):
        CK.resume(str(tmp_path / "missing.pt"), None)
