"""Checkpoint I/O in the reference's own ``.pt`` layouts, so files move freely between the reference
scripts and this engine.  Parameters, BatchNorm buffers and optimizer moments live in the caller's ``nn.Module`` /
``torch.optim`` objects (the engine updates them in place through raw pointers), so a checkpoint is exactly what the
reference writes: plain ``state_dict()``s under the reference's keys, pickled with ``torch.save``.

layout        written by (reference)                              keys
``ssl_cr``    eval_{BreastPathQ,Camelyon,Kather}_SSL_CR.py        args, model_student, model_teacher, classifier_teacher,
              (eval_BreastPathQ_SSL_CR.py:519-533,                 classifier_student, optimizer, epoch, train_loss, train_losses_x,
               eval_Camelyon_SSL_CR.py:575-590)                    train_losses_u [, train_acc, val_acc, val_loss]
``finetune``  eval_{BreastPathQ,Camelyon,Kather}_SSL.py           args, model, classifier, optimizer, epoch, train_loss
              (eval_Camelyon_SSL.py:424-435)                       [, train_acc, val_acc, val_loss]
``pretrain``  pretrain_{BreastPathQ,Camelyon16}.py, pretrain_RSP  args, model, optimizer, epoch, train_loss, train_acc
              (pretrain_BreastPathQ.py:298-305)                    (the RSP classifier is NOT saved by the reference)

Multi-GPU reference runs wrap the modules in ``nn.DataParallel`` before saving (eval_BreastPathQ_SSL_CR.py:474-477), so
their keys carry a ``module.`` prefix, which the consuming scripts strip with ``k[7:]`` (eval_Camelyon_SSL_CR.py:405-412).
Readers here accept both spellings; writers take ``data_parallel_keys=True`` to produce the prefixed form.
"""
import argparse
import os
from collections import OrderedDict

import torch

from .net import strip_module_prefix, unwrap

LAYOUT_KEYS = {
    "ssl_cr": ("args", "model_student", "model_teacher", "classifier_teacher", "classifier_student", "optimizer", "epoch"),
    "finetune": ("args", "model", "classifier", "optimizer", "epoch"),
    "pretrain": ("args", "model", "optimizer", "epoch"),
}


def _sd(module, data_parallel_keys):
    """state_dict of the bare module, CPU-side keys exactly as the reference's (optionally DataParallel-prefixed)."""
    sd = unwrap(module).state_dict()
    if data_parallel_keys:
        return OrderedDict(("module." + k, v) for k, v in sd.items())
    return sd


def _inner(optimizer):
    """Lookahead wraps the real optimizer; the reference saves the inner one (pretrain_BreastPathQ.py:301)."""
    return getattr(optimizer, "optimizer", optimizer)


def load_file(path, map_location=None):
    """torch.load of a reference checkpoint.  The reference pickles its ``argparse.Namespace`` next to the tensors; that
    one class is allow-listed for the weights-only unpickler (nothing else is executed)."""
    with torch.serialization.safe_globals([argparse.Namespace]):
        return torch.load(path, map_location=map_location, weights_only=True)


def detect_layout(ckpt):
    for name in ("ssl_cr", "finetune", "pretrain"):
        if all(k in ckpt for k in LAYOUT_KEYS[name]):
            return name
    raise KeyError("not a reference checkpoint layout: keys = %s" % sorted(ckpt.keys()))


def _load_module(module, sd):
    """load_state_dict into the bare module whether or not the file / the module are DataParallel-wrapped (the reference strips
    ``module.`` with k[7:], eval_Camelyon_SSL_CR.py:405-412, or loads prefixed keys into wrapped modules, :526-529)."""
    unwrap(module).load_state_dict(strip_module_prefix(sd))


# ------------------------------------------------------------------------------------------------------------ writers
def save_ssl_cr(path, args, model_student, model_teacher, classifier_teacher, classifier_student, optimizer, epoch,
                train_loss, train_losses_x, train_losses_u, data_parallel_keys=False, **extra):
    """eval_BreastPathQ_SSL_CR.py:519-533 (extra = train_acc / val_acc / val_loss of the Camelyon and Kather scripts)."""
    state = {"args": args,
             "model_student": _sd(model_student, data_parallel_keys), "model_teacher": _sd(model_teacher, data_parallel_keys),
             "classifier_teacher": _sd(classifier_teacher, data_parallel_keys),
             "classifier_student": _sd(classifier_student, data_parallel_keys),
             "optimizer": _inner(optimizer).state_dict(), "epoch": epoch, "train_loss": train_loss,
             "train_losses_x": train_losses_x, "train_losses_u": train_losses_u}
    state.update(extra)
    torch.save(state, path)
    return state


def save_finetune(path, args, model, classifier, optimizer, epoch, train_loss, data_parallel_keys=False, **extra):
    """eval_Camelyon_SSL.py:424-435 / eval_BreastPathQ_SSL.py (extra = train_acc, val_acc, val_loss)."""
    state = {"args": args, "model": _sd(model, data_parallel_keys), "classifier": _sd(classifier, data_parallel_keys),
             "optimizer": _inner(optimizer).state_dict(), "epoch": epoch, "train_loss": train_loss}
    state.update(extra)
    torch.save(state, path)
    return state


def save_pretrain(path, args, model, optimizer, epoch, train_loss, train_acc, data_parallel_keys=False):
    """pretrain_BreastPathQ.py:298-305 -- 'model' + 'optimizer' only, like the reference."""
    state = {"args": args, "model": _sd(model, data_parallel_keys), "optimizer": _inner(optimizer).state_dict(),
             "epoch": epoch, "train_loss": train_loss, "train_acc": train_acc}
    torch.save(state, path)
    return state


# ------------------------------------------------------------------------------------------------------------ readers
def load_pretrained(model, path_or_ckpt, map_location=None):
    """the fine-tuning scripts' start: ``state_dict['model']`` of an RSP pre-training file, ``module.`` stripped
    (eval_Camelyon_SSL.py:319-331, eval_BreastPathQ_SSL.py:341-353).  A TripletNet file loads into TripletNet_Finetune: the
    two share every key (models/net.py:25-48,70-84)."""
    ckpt = path_or_ckpt if isinstance(path_or_ckpt, dict) else load_file(path_or_ckpt, map_location)
    _load_module(model, ckpt["model"])
    return ckpt


def load_finetuned(path_or_ckpt, models, classifiers, map_location=None):
    """the SSL_CR scripts' start: teacher AND student from the fine-tuned file's 'model' / 'classifier'
    (eval_BreastPathQ_SSL_CR.py:394-402, eval_Camelyon_SSL_CR.py:405-412,449-464)."""
    ckpt = path_or_ckpt if isinstance(path_or_ckpt, dict) else load_file(path_or_ckpt, map_location)
    for m in models:
        _load_module(m, ckpt["model"])
    for c in classifiers:
        _load_module(c, ckpt["classifier"])
    return ckpt


def _load_optimizer(optimizer, sd):
    """optimizer.load_state_dict, then make sure every state tensor sits with its parameter and is contiguous fp32: the engine
    updates the moments in place through raw pointers (torch's own load already casts/moves; this is the assertion of it)."""
    inner = _inner(optimizer)
    inner.load_state_dict(sd)
    for p, st in inner.state.items():
        for k, v in st.items():
            if torch.is_tensor(v) and v.dim() > 0 and (v.device != p.device or not v.is_contiguous()):
                st[k] = v.to(p.device).contiguous()


def resume(path, optimizer=None, map_location=None, **modules):
    """the reference's ``--resume`` block for whichever layout the file has:
      ssl_cr   (eval_Camelyon_SSL_CR.py:522-538): model_student, model_teacher, classifier_teacher, classifier_student, optimizer
      finetune (eval_Camelyon_SSL.py:378-393):     model, classifier, optimizer
      pretrain (pretrain_BreastPathQ.py:256-266):  model, optimizer
    ``modules`` are passed by those names.  Returns ``(start_epoch, checkpoint)``; start_epoch = checkpoint['epoch'] + 1."""
    if not os.path.isfile(path):
        raise FileNotFoundError("=> no checkpoint found at '{}'".format(path))
    ckpt = load_file(path, map_location)
    layout = detect_layout(ckpt)
    want = [k for k in LAYOUT_KEYS[layout] if k not in ("args", "optimizer", "epoch")]
    missing = [k for k in want if k not in modules]
    if missing:
        raise TypeError(f"resume of a '{layout}' checkpoint needs modules {want}; missing {missing}")
    for k in want:
        _load_module(modules[k], ckpt[k])
    if optimizer is not None:
        _load_optimizer(optimizer, ckpt["optimizer"])
    return ckpt["epoch"] + 1, ckpt
