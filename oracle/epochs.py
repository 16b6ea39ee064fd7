"""Restatement of the reference's ``train()``/``validate()`` EPOCH loops on top of oracle.steps
(test oracle).  Same loader contracts, casts, reshapes, shuffles, meters and return tuples as

  * eval_BreastPathQ_SSL_CR.py:37-128 (train) / :131-175 (validate)
  * eval_Camelyon_SSL_CR.py:33-157 (train) / :160-225 (validate)
  * pretrain_BreastPathQ.py:27-92 (train) / :95-148 (validate)
  * eval_Camelyon_SSL.py:31-119, eval_BreastPathQ_SSL.py:35-103 (supervised train)
"""
import torch

from . import steps as S


def bpq_cr_train(ps, bs, pt, bt, opt, labeled, unlabeled, lambda_u, faithful=True):
    losses, losses_x, losses_u = S.AverageMeter(), S.AverageMeter(), S.AverageMeter()
    feats, targets = [], []
    for (x, y), (u_w, u_s) in zip(labeled, unlabeled):
        x, u_w, u_s, y = x.float(), u_w.float(), u_s.float(), y.float()
        x = x.reshape(-1, 3, 256, 256)                                   # :74 (hard-coded 256)
        r = S.ssl_cr_step("mse", ps, bs, pt, bt, opt, x, y.reshape(-1), u_w, u_s, lambda_u, faithful)
        n = x.shape[0]
        losses.update(r["loss"], n); losses_x.update(r["loss_x"], n); losses_u.update(r["loss_u"], n)
        feats.append(r["feats"]); targets.append(y)
    return losses.avg, losses_x.avg, losses_u.avg, torch.cat(feats), torch.cat(targets)


def bpq_cr_validate(ps, bs, val_loader, faithful=True):
    losses = S.AverageMeter()
    for x, y in val_loader:
        r = S.supervised_step("mse", ps, bs, None, x.float(), y.float(), faithful, train=False)
        losses.update(r["loss"], y.size(0))
    return losses.avg


def _cat_shuffle(t_x, n_x, perm):
    return torch.cat([t_x, n_x])[perm]


def cam_cr_train(ps, bs, pt, bt, opt, tum_l, nor_l, tum_u, nor_u, lambda_u, image_size, faithful=True):
    losses, losses_x, losses_u, acc = (S.AverageMeter() for _ in range(4))
    feats, targets = [], []
    for (tx, ty), (nx_, ny), (tuw, tus), (nuw, nus) in zip(tum_l, nor_l, tum_u, nor_u):
        tx = tx.reshape(-1, 3, image_size, image_size).float(); ty = ty.reshape(-1).long()
        nx_ = nx_.reshape(-1, 3, image_size, image_size).float(); ny = ny.reshape(-1).long()
        tuw, tus, nuw, nus = tuw.float(), tus.float(), nuw.float(), nus.float()
        p_x = torch.randperm(2 * len(tx))                                   # :79-81, same order
        p_uw = torch.randperm(2 * len(tuw))
        p_us = torch.randperm(2 * len(tus))
        x = _cat_shuffle(tx, nx_, p_x); y = _cat_shuffle(ty, ny, p_x)
        u_w = _cat_shuffle(tuw, nuw, p_uw); u_s = _cat_shuffle(tus, nus, p_us)
        r = S.ssl_cr_step("ce", ps, bs, pt, bt, opt, x, y, u_w, u_s, lambda_u, faithful)
        n = x.shape[0]
        losses_x.update(r["loss_x"], n); losses_u.update(r["loss_u"], n); losses.update(r["loss"], n)
        acc.update(r["acc"], n)
        feats.append(r["feats"][:n]); targets.append(y)
    return losses.avg, losses_x.avg, losses_u.avg, acc.avg, torch.cat(feats), torch.cat(targets)


def cam_cr_validate(ps, bs, val_tumor, val_normal, faithful=True):
    losses, acc = S.AverageMeter(), S.AverageMeter()
    for (tx, ty), (nx_, ny) in zip(val_tumor, val_normal):
        perm = torch.randperm(2 * len(tx))
        x = torch.cat([tx, nx_])[perm].float(); y = torch.cat([ty, ny])[perm].long()
        r = S.supervised_step("ce", ps, bs, None, x, y, faithful, train=False)
        losses.update(r["loss"], y.size(0)); acc.update(r["acc"], y.size(0))
    return losses.avg, acc.avg


def kather_cr_train(ps, bs, pt, bt, opt, labeled, unlabeled, lambda_u, faithful=True):
    """eval_Kather_SSL_CR.py:37-127 -> (loss, loss_x, loss_u, acc)."""
    losses, losses_x, losses_u, acc = (S.AverageMeter() for _ in range(4))
    for (x, y), (u_w, u_s) in zip(labeled, unlabeled):
        x = x.float().reshape(-1, 3, 256, 256)                               # :68
        y = y.long().reshape(-1)
        r = S.ssl_cr_step("ce", ps, bs, pt, bt, opt, x, y, u_w.float(), u_s.float(), lambda_u, faithful)
        n = x.shape[0]
        losses_x.update(r["loss_x"], n); losses_u.update(r["loss_u"], n); losses.update(r["loss"], n); acc.update(r["acc"], n)
    return losses.avg, losses_x.avg, losses_u.avg, acc.avg


def kather_cr_validate(ps, bs, val_loader, faithful=True):
    losses, acc = S.AverageMeter(), S.AverageMeter()
    for x, y in val_loader:
        r = S.supervised_step("ce", ps, bs, None, x.float(), y.long(), faithful, train=False)
        losses.update(r["loss"], y.size(0)); acc.update(r["acc"], y.size(0))
    return losses.avg, acc.avg


def cam_wsi_test(p, b, test_loader, faithful=True):
    """test_Camelyon16.py:30-70 -> probs_map (float64, shape of the tissue mask): eval forward, softmax, last column
    ('tumor'), scattered to the mask coordinates of each tile."""
    import numpy as np
    probs_map = np.zeros(test_loader.dataset.mask.shape)
    for inp, x_mask, y_mask in test_loader:
        with torch.no_grad():
            feats = S.M.finetune_forward(p, b, inp, False, faithful)
            output = S.M.classifier_forward(p, feats)
            probs = torch.softmax(output, dim=1)[:, -1]
        probs_map[x_mask.numpy(), y_mask.numpy()] = probs.numpy()
    return probs_map


def rsp_epoch(p, b, opt, loader, tile, train=True):
    losses, acc = S.AverageMeter(), S.AverageMeter()
    feats, targets = [], []
    for i1, i2, i3, t in loader:
        i1, i2, i3 = (v.float().reshape(-1, 3, tile, tile) for v in (i1, i2, i3))
        t = t.long().view(-1, 1).reshape(-1)
        r = S.rsp_step(p, b, opt, i1, i2, i3, t, train)
        losses.update(r["loss"], t.size(0)); acc.update(r["acc"], t.size(0))
        feats.append(r["feats"]); targets.append(t)
    if train:
        return losses.avg, acc.avg, torch.cat(feats), torch.cat(targets)
    return losses.avg, acc.avg


def cam_sup_train(p, b, opt, tum_l, nor_l, image_size, faithful=True):
    losses, acc = S.AverageMeter(), S.AverageMeter()
    feats, targets = [], []
    for (tx, ty), (nx_, ny) in zip(tum_l, nor_l):
        tx = tx.reshape(-1, 3, image_size, image_size).float(); ty = ty.reshape(-1).long()
        nx_ = nx_.reshape(-1, 3, image_size, image_size).float(); ny = ny.reshape(-1).long()
        perm = torch.randperm(2 * len(tx))
        x = _cat_shuffle(tx, nx_, perm); y = _cat_shuffle(ty, ny, perm)
        r = S.supervised_step("ce", p, b, opt, x, y, faithful)
        losses.update(r["loss"], x.shape[0]); acc.update(r["acc"], x.shape[0])
        feats.append(r["feats"]); targets.append(y)
    return losses.avg, acc.avg, torch.cat(feats), torch.cat(targets)


def bpq_sup_train(p, b, opt, loader, image_size, faithful=True):
    losses = S.AverageMeter()
    feats, targets = [], []
    for x, y in loader:
        x = x.float().reshape(-1, 3, image_size, image_size); y = y.float().reshape(-1)
        r = S.supervised_step("mse", p, b, opt, x, y, faithful)
        losses.update(r["loss"], y.size(0))
        feats.append(r["feats"]); targets.append(y)
    return losses.avg, torch.cat(feats), torch.cat(targets)


def kather_sup_train(p, b, opt, loader, image_size, faithful=True):
    """eval_Kather_SSL.py:32-99 -> (loss_avg, acc_avg)."""
    losses, acc = S.AverageMeter(), S.AverageMeter()
    for x, y in loader:
        x = x.float().reshape(-1, 3, image_size, image_size); y = y.long().reshape(-1)          # :54-57
        r = S.supervised_step("ce", p, b, opt, x, y, faithful)
        losses.update(r["loss"], y.size(0)); acc.update(r["acc"], y.size(0))
    return losses.avg, acc.avg


def kather_sup_validate(p, b, val_loader, faithful=True):
    """eval_Kather_SSL.py:102-151 -> (loss_avg, acc_avg)."""
    return kather_cr_validate(p, b, val_loader, faithful)
