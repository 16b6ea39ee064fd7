"""Drop-in ``train()`` / ``validate()`` for the reference scripts, same signatures and return tuples, running every
iteration as ONE native engine step (teacher fwd + student fwd + losses + backward + all-reduce) plus one fused
optimizer launch.  Loaders are any iterables yielding the reference's batch tuples, uint8 or float,
CPU or GPU.  Differences from the reference loops that do not change results: no per-iteration ``.item()`` host syncs
(meters are filled from device scalars at print/epoch boundaries) and features are concatenated once, not per step.

reference                                   here
eval_BreastPathQ_SSL_CR.train/validate      bpq_cr_train / bpq_cr_validate      (:37-128 / :131-175)
eval_Camelyon_SSL_CR.train/validate         cam_cr_train / cam_cr_validate      (:33-157 / :160-225)
eval_Kather_SSL_CR.train/validate           kather_cr_train / kather_cr_validate(:37-127 / :130-179)
pretrain_BreastPathQ|Camelyon16|RSP.train   rsp_train / rsp_validate            (:27-92 / :95-148)
eval_Camelyon_SSL.train                     cam_sup_train                       (:31-119)
eval_BreastPathQ_SSL.train                  bpq_sup_train                       (:35-103)
eval_Kather_SSL.train/validate              kather_sup_train / kather_sup_validate (:32-99 / :102-151)
"""
import time

import torch

from .engine import get_engine
from .util import AverageMeter


class _Meters:
    """AverageMeter semantics (util.py:26-46) fed from device scalars without a sync per step.

    Sharded runs (one process per GPU): a step's losses are THIS rank's share -- already scaled by 1/global-count, so the SUM
    over ranks is the global value -- and stay local until somebody looks: meters() issues ONE all-reduce of all rows gathered so
    far ("one small all-reduce per print_freq"), on the engine's own communicator, and only then syncs to the
    host.  `reduce` is that in-place SUM over ranks (None on one rank), `world` scales the meter weights to global counts."""

    def __init__(self, names, reduce=None, world=1, device=None):
        self.names = names
        self.device = device                # where the row-count header lives when no row has been added yet
        self.rows, self.weights = [], []
        self.reduce, self.world = reduce, world
        self.done = []                      # rows already reduced and fetched: python lists

    def add(self, losses, n):
        self.rows.append(losses)
        self.weights.append(n * self.world)

    def meters(self, acc_denoms=None):
        """COLLECTIVE on sharded runs: every rank must call it at the same points with the same number of rows gathered since
        its last call (the step functions do: print_freq boundaries and epoch end of equally long loaders).  A rank whose loader
        is shorter would otherwise reduce a buffer of another length -- a hang or mismatched sums -- so a fixed-size header with
        the row count is reduced first and a disagreement raises on every rank."""
        out = {k: AverageMeter() for k in self.names}
        if self.reduce is not None:
            # the count's three base-256 DIGITS and their squares: all ranks hold the same n  <=>  W * sum(d^2) == (sum d)^2 for every
            # digit (Cauchy-Schwarz with equality).  A square is at most 255^2 and a sum over ranks at most W * 255^2 < 2^24 for
            # W <= 256, so every partial sum of any reduction order is an exact fp32 integer (with base-4096 digits the cross-rank
            # sums of squares passed 2^24 and could round: a false mismatch at world >= 3)
            n = len(self.rows)
            if self.world > 256:
                raise RuntimeError("_Meters.meters(): the row-count header is exact in fp32 for world <= 256 only")
            if n >= 1 << 24:
                raise RuntimeError("_Meters.meters(): more than 2^24 rows gathered between two reads")
            d = [float((n >> (8 * i)) & 255) for i in range(3)]
            dev = self.rows[0].device if self.rows else self.device
            hdr = torch.tensor(d + [x * x for x in d], dtype=torch.float32, device=dev)
            self.reduce(hdr)
            h = hdr.cpu().tolist()
            if any(h[3 + i] * self.world != h[i] * h[i] for i in range(3)):
                mean = sum(h[i] * 256 ** i for i in range(3)) / self.world
                raise RuntimeError(f"_Meters.meters(): the ranks gathered different numbers of steps since the last read (this rank "
                                   f"{n}, mean over ranks {mean:g}): the loaders' lengths differ, or "
                                   f"meters() was not called on every rank")
        if self.rows:
            vals = torch.stack(self.rows)
            if self.reduce is not None:
                self.reduce(vals)                                  # the one data collective per print / epoch boundary
            self.done += vals.cpu().tolist()                       # the device->host sync
            self.rows = []
        for r, n in zip(self.done, self.weights):
            for k in self.names:
                if k == "acc":
                    out[k].update(r[3] / n, n)
                else:
                    out[k].update(r[{"loss": 0, "loss_x": 1, "loss_u": 2}[k]], n)
        return out


def _meters(eng, names):
    return _Meters(names, eng.all_reduce_sum if eng.world > 1 else None, eng.world, eng.device)


def _prefetch_on(args):
    import os
    return bool(getattr(args, "device_prefetch", False)) or os.environ.get("SSLCR_PREFETCH", "0") not in ("", "0")


def _ahead(batches, eng, enabled):
    """Host-fed loaders: move batch k+1's image tensors (dim >= 4, still in host memory) to the device on a copy stream while
    step k runs, so the PCIe copy (214 MB per benchmark step = 4.3 ms at 50 GB/s, tools/pcie_inclusive.py) hides under compute.
    Everything else in a batch (targets, WSI tile coordinates) is left where it is.  OFF unless ``args.device_prefetch`` or
    SSLCR_PREFETCH=1: it calls ``next(loader)`` one step early, which reorders the loader's random draws against the step's own
    (``torch.randperm`` in the Camelyon loop) when both use the main process's generators -- harmless with DataLoader workers
    (they draw from their own), a different random stream than the reference's with ``num_workers=0``."""
    if not enabled:
        yield from batches
        return
    dev = eng.device
    copy = eng.copy_stream()

    def move(obj, moved):
        if torch.is_tensor(obj):
            if obj.dim() >= 4 and not obj.is_cuda:
                t = obj.to(dev, non_blocking=True)
                moved.append(t)
                return t
            return obj
        if isinstance(obj, (tuple, list)):
            return type(obj)(move(o, moved) for o in obj)
        return obj

    it = iter(batches)

    def fetch():
        try:
            b = next(it)
        except StopIteration:
            return None
        moved = []
        with torch.cuda.stream(copy):
            mb = move(b, moved)
            ev = torch.cuda.Event()
            ev.record(copy)
        return mb, ev, moved

    nxt = fetch()
    while nxt is not None:
        cur, ev, moved = nxt
        nxt = fetch()                                   # batch k+1's copies are queued before step k is launched
        st = torch.cuda.current_stream(dev)
        st.wait_event(ev)
        for t in moved:
            t.record_stream(st)                         # allocated on the copy stream, consumed on the compute stream
        yield cur


def _device_of(model):
    return next(model.parameters()).device


def _maybe_print(args, batch_idx, tag, epoch, total, t0, meters):
    pf = getattr(args, "print_freq", 0)
    if pf and (batch_idx + 1) % pf == 0:
        m = meters.meters()
        body = "\t".join(f"{k} {v.val:.3f} ({v.avg:.3f})" for k, v in m.items())
        print(f"{tag}: [{epoch}][{batch_idx + 1}/{total}]\tBT {(time.time() - t0) / (batch_idx + 1):.3f}\t{body}")


def _len(x):
    try:
        return len(x)
    except TypeError:
        return -1


# ------------------------------------------------------------------------------------------------ BreastPathQ SSL_CR
def bpq_cr_train(args, model_teacher, model_student, classifier_teacher, classifier_student, labeled_train_loader,
                 unlabeled_train_loader, optimizer, epoch):
    """eval_BreastPathQ_SSL_CR.train: returns (loss_avg, loss_x_avg, loss_u_avg, final_feats, final_targets)."""
    eng = get_engine(_device_of(model_student))
    for m in (model_teacher, classifier_teacher):
        m.eval()
    for m in (model_student, classifier_student):
        m.train()
    te, st = eng.bind(model_teacher, classifier_teacher), eng.bind(model_student, classifier_student)
    meters = _meters(eng, ["loss", "loss_x", "loss_u"])
    feats, targets = [], []
    t0 = time.time()
    for batch_idx, (data_x, data_u) in enumerate(_ahead(zip(labeled_train_loader, unlabeled_train_loader), eng, _prefetch_on(args))):
        inputs_x, targets_x = data_x
        inputs_u_w, inputs_u_s = data_u
        inputs_x = inputs_x.reshape(-1, 3, 256, 256)                         # :74 (hard-coded by the reference)
        targets_x = targets_x.float().to(eng.device)
        r = eng.step_ssl_cr(te, st, "mse", inputs_x, targets_x.reshape(-1), inputs_u_w, inputs_u_s, args.lambda_u)
        st.optimizer_step(optimizer)
        meters.add(r["losses"], inputs_x.shape[0])
        feats.append(r["feats"])
        targets.append(targets_x)
        _maybe_print(args, batch_idx, "Train", epoch, _len(labeled_train_loader), t0, meters)
    m = meters.meters()
    return m["loss"].avg, m["loss_x"].avg, m["loss_u"].avg, torch.cat(feats).detach(), torch.cat(targets).detach()


def bpq_cr_validate(args, model_student, classifier_student, val_loader, epoch):
    """eval_BreastPathQ_SSL_CR.validate -> loss_avg."""
    eng = get_engine(_device_of(model_student))
    model_student.eval()
    classifier_student.eval()
    st = eng.bind(model_student, classifier_student)
    meters = _meters(eng, ["loss"])
    t0 = time.time()
    for batch_idx, (input, target) in enumerate(_ahead(val_loader, eng, _prefetch_on(args))):
        r = eng.step_supervised(st, "mse", [input], target.float().reshape(-1), train=False)
        meters.add(r["losses"], target.size(0))
        _maybe_print(args, batch_idx, "Val", epoch, _len(val_loader), t0, meters)
    return meters.meters()["loss"].avg


# ------------------------------------------------------------------------------------------------ Camelyon16 SSL_CR
def _cat_shuffle(a, b, perm):
    return torch.cat([a, b])[perm]


def cam_cr_train(args, model_teacher, model_student, classifier_teacher, classifier_student, tumor_labeled_train_loader,
                 normal_labeled_train_loader, tumor_unlabeled_train_loader, normal_unlabeled_train_loader, optimizer, epoch):
    """eval_Camelyon_SSL_CR.train: returns (loss, loss_x, loss_u, acc, final_feats[:labeled], final_targets)."""
    eng = get_engine(_device_of(model_student))
    for m in (model_teacher, classifier_teacher):
        m.eval()
    for m in (model_student, classifier_student):
        m.train()
    te, st = eng.bind(model_teacher, classifier_teacher), eng.bind(model_student, classifier_student)
    meters = _meters(eng, ["loss", "loss_x", "loss_u", "acc"])
    feats, targets = [], []
    t0 = time.time()
    S = args.image_size
    loaders = zip(tumor_labeled_train_loader, normal_labeled_train_loader, tumor_unlabeled_train_loader,
                  normal_unlabeled_train_loader)
    for batch_idx, (tumor_data_x, normal_data_x, tumor_data_u, normal_data_u) in enumerate(_ahead(loaders, eng, _prefetch_on(args))):
        t_x, t_y = tumor_data_x
        n_x, n_y = normal_data_x
        t_x, t_y = t_x.reshape(-1, 3, S, S), t_y.reshape(-1)
        n_x, n_y = n_x.reshape(-1, 3, S, S), n_y.reshape(-1)
        t_uw, t_us = tumor_data_u
        n_uw, n_us = normal_data_u
        p_x = torch.randperm(2 * len(t_x))                       # same three draws, same order as :79-81
        p_uw = torch.randperm(2 * len(t_uw))
        p_us = torch.randperm(2 * len(t_us))
        x, y = _cat_shuffle(t_x, n_x, p_x.to(t_x.device)), _cat_shuffle(t_y, n_y, p_x.to(t_y.device)).long()
        u_w, u_s = _cat_shuffle(t_uw, n_uw, p_uw.to(t_uw.device)), _cat_shuffle(t_us, n_us, p_us.to(t_us.device))
        r = eng.step_ssl_cr(te, st, "ce", x, y, u_w, u_s, args.lambda_u)
        st.optimizer_step(optimizer)
        n = x.shape[0]
        meters.add(r["losses"], n)
        feats.append(r["feats"][:n])
        targets.append(y.to(eng.device))
        _maybe_print(args, batch_idx, "Train", epoch, _len(tumor_labeled_train_loader) * 2, t0, meters)
    m = meters.meters()
    return m["loss"].avg, m["loss_x"].avg, m["loss_u"].avg, m["acc"].avg, torch.cat(feats).detach(), torch.cat(targets).detach()


def cam_cr_validate(args, model_student, classifier_student, val_tumor_loader, val_normal_loader, epoch):
    """eval_Camelyon_SSL_CR.validate -> (loss_avg, acc_avg)."""
    eng = get_engine(_device_of(model_student))
    model_student.eval()
    classifier_student.eval()
    st = eng.bind(model_student, classifier_student)
    meters = _meters(eng, ["loss", "acc"])
    t0 = time.time()
    for batch_idx, (data_tumor, data_normal) in enumerate(_ahead(zip(val_tumor_loader, val_normal_loader), eng, _prefetch_on(args))):
        t_x, t_y = data_tumor
        n_x, n_y = data_normal
        perm = torch.randperm(2 * len(t_x))
        x = torch.cat([t_x, n_x])[perm.to(t_x.device)]
        y = torch.cat([t_y, n_y])[perm.to(t_y.device)].long()
        r = eng.step_supervised(st, "ce", [x], y, train=False)
        meters.add(r["losses"], y.size(0))
        _maybe_print(args, batch_idx, "Val", epoch, 2 * _len(val_tumor_loader), t0, meters)
    m = meters.meters()
    return m["loss"].avg, m["acc"].avg


# ------------------------------------------------------------------------------------------------ Kather SSL_CR
def kather_cr_train(args, model_teacher, model_student, classifier_teacher, classifier_student, labeled_train_loader,
                    unlabeled_train_loader, optimizer, epoch):
    """eval_Kather_SSL_CR.train: CE + hard-pseudo-label CE, single labeled/unlabeled loader pair;
    returns (loss, loss_x, loss_u, acc)."""
    eng = get_engine(_device_of(model_student))
    for m in (model_teacher, classifier_teacher):
        m.eval()
    for m in (model_student, classifier_student):
        m.train()
    te, st = eng.bind(model_teacher, classifier_teacher), eng.bind(model_student, classifier_student)
    meters = _meters(eng, ["loss", "loss_x", "loss_u", "acc"])
    t0 = time.time()
    for batch_idx, (data_x, data_u) in enumerate(_ahead(zip(labeled_train_loader, unlabeled_train_loader), eng, _prefetch_on(args))):
        inputs_x, targets_x = data_x
        inputs_u_w, inputs_u_s = data_u
        inputs_x = inputs_x.reshape(-1, 3, 256, 256)                          # :68
        targets_x = targets_x.reshape(-1).long()                              # :69
        r = eng.step_ssl_cr(te, st, "ce", inputs_x, targets_x, inputs_u_w, inputs_u_s, args.lambda_u)
        st.optimizer_step(optimizer)
        meters.add(r["losses"], inputs_x.shape[0])
        _maybe_print(args, batch_idx, "Train", epoch, _len(labeled_train_loader), t0, meters)
    m = meters.meters()
    return m["loss"].avg, m["loss_x"].avg, m["loss_u"].avg, m["acc"].avg


def kather_cr_validate(args, model_student, classifier_student, val_loader, epoch):
    """eval_Kather_SSL_CR.validate -> (loss_avg, acc_avg)."""
    eng = get_engine(_device_of(model_student))
    model_student.eval()
    classifier_student.eval()
    st = eng.bind(model_student, classifier_student)
    meters = _meters(eng, ["loss", "acc"])
    for batch_idx, (input, target) in enumerate(_ahead(val_loader, eng, _prefetch_on(args))):
        r = eng.step_supervised(st, "ce", [input], target.reshape(-1).long(), train=False)
        meters.add(r["losses"], target.size(0))
    m = meters.meters()
    return m["loss"].avg, m["acc"].avg


# ------------------------------------------------------------------------------------------------ RSP pretraining
def _plain_ce(criterion, where):
    """the engine computes plain mean cross-entropy (what the reference's nn.CrossEntropyLoss() is): anything else -- class
    weights, label smoothing, a non-default ignore_index or reduction -- must not be silently dropped."""
    if criterion is None:
        return
    if not isinstance(criterion, torch.nn.CrossEntropyLoss):
        raise NotImplementedError(f"{where}: the reference uses nn.CrossEntropyLoss")
    if criterion.weight is not None or getattr(criterion, "label_smoothing", 0.0) != 0.0 or criterion.reduction != "mean" or \
            criterion.ignore_index != -100:
        raise NotImplementedError(f"{where}: only the default nn.CrossEntropyLoss() (no weight / label_smoothing / ignore_index, "
                                  "reduction='mean') is implemented by the engine")


def _rsp_epoch(args, model, classifier, loader, criterion, optimizer, epoch, train):
    _plain_ce(criterion, "RSP pretraining (pretrain_BreastPathQ.py:56)")
    eng = get_engine(_device_of(model))
    model.train(train)
    classifier.train(train)
    net = eng.bind(model, classifier)
    meters = _meters(eng, ["loss", "acc"])
    feats, targets = [], []
    t0 = time.time()
    for batch_idx, (input1, input2, input3, target) in enumerate(_ahead(loader, eng, _prefetch_on(args))):
        i1, i2, i3 = (v.reshape(-1, 3, args.tile_h, args.tile_w) for v in (input1, input2, input3))
        target = target.long().view(-1, 1).reshape(-1)
        r = eng.step_supervised(net, "ce", [i1, i2, i3], target, train=train)
        if train:
            net.optimizer_step(optimizer)
        meters.add(r["losses"], target.size(0))
        if train:
            feats.append(r["feats"])
            targets.append(target.to(eng.device))
        _maybe_print(args, batch_idx, "Train" if train else "Val", epoch, _len(loader), t0, meters)
    m = meters.meters()
    if train:
        return m["loss"].avg, m["acc"].avg, torch.cat(feats).detach(), torch.cat(targets).detach()
    return m["loss"].avg, m["acc"].avg


def rsp_train(args, model, classifier, train_loader, criterion, optimizer, epoch):
    """pretrain_BreastPathQ.train (= pretrain_Camelyon16, Pretraining_v2/pretrain_RSP): (loss, acc, feats, targets)."""
    return _rsp_epoch(args, model, classifier, train_loader, criterion, optimizer, epoch, True)


def rsp_validate(args, model, classifier, val_loader, criterion, epoch):
    return _rsp_epoch(args, model, classifier, val_loader, criterion, None, epoch, False)


# ------------------------------------------------------------------------------------------------ supervised fine-tune
def cam_sup_train(args, model, classifier, tumor_labeled_train_loader, normal_labeled_train_loader, optimizer, epoch):
    """eval_Camelyon_SSL.train -> (loss, acc, feats, targets)."""
    eng = get_engine(_device_of(model))
    model.train()
    classifier.train()
    net = eng.bind(model, classifier)
    meters = _meters(eng, ["loss", "acc"])
    feats, targets = [], []
    S = args.image_size
    for batch_idx, (tumor_data_x, normal_data_x) in enumerate(_ahead(zip(tumor_labeled_train_loader, normal_labeled_train_loader), eng,
                                                                     _prefetch_on(args))):
        t_x, t_y = tumor_data_x
        n_x, n_y = normal_data_x
        t_x, t_y = t_x.reshape(-1, 3, S, S), t_y.reshape(-1)
        n_x, n_y = n_x.reshape(-1, 3, S, S), n_y.reshape(-1)
        perm = torch.randperm(2 * len(t_x))
        x, y = _cat_shuffle(t_x, n_x, perm.to(t_x.device)), _cat_shuffle(t_y, n_y, perm.to(t_y.device)).long()
        r = eng.step_supervised(net, "ce", [x], y, train=True)
        net.optimizer_step(optimizer)
        meters.add(r["losses"], x.shape[0])
        feats.append(r["feats"])
        targets.append(y.to(eng.device))
    m = meters.meters()
    return m["loss"].avg, m["acc"].avg, torch.cat(feats).detach(), torch.cat(targets).detach()


def bpq_sup_train(args, model, classifier, train_loader, criterion, optimizer, epoch):
    """eval_BreastPathQ_SSL.train -> (loss, feats, targets)."""
    if criterion is not None and not isinstance(criterion, torch.nn.MSELoss):
        raise NotImplementedError("the reference fine-tunes BreastPathQ with nn.MSELoss")
    eng = get_engine(_device_of(model))
    model.train()
    classifier.train()
    net = eng.bind(model, classifier)
    meters = _meters(eng, ["loss"])
    feats, targets = [], []
    for batch_idx, (input1, target) in enumerate(_ahead(train_loader, eng, _prefetch_on(args))):
        x = input1.reshape(-1, 3, args.image_size, args.image_size)
        y = target.float().reshape(-1)
        r = eng.step_supervised(net, "mse", [x], y, train=True)
        net.optimizer_step(optimizer)
        meters.add(r["losses"], y.size(0))
        feats.append(r["feats"])
        targets.append(y.to(eng.device))
    m = meters.meters()
    return m["loss"].avg, torch.cat(feats).detach(), torch.cat(targets).detach()


def kather_sup_train(args, model, classifier, train_loader, criterion, optimizer, epoch):
    """eval_Kather_SSL.train (:32-99; the reference file does not parse as a whole, :243): student-only CE -> (loss, acc)."""
    _plain_ce(criterion, "Kather fine-tuning (eval_Kather_SSL.py:410)")
    eng = get_engine(_device_of(model))
    model.train()
    classifier.train()
    net = eng.bind(model, classifier)
    meters = _meters(eng, ["loss", "acc"])
    for batch_idx, (input, target) in enumerate(_ahead(train_loader, eng, _prefetch_on(args))):
        x = input.reshape(-1, 3, args.image_size, args.image_size)                   # :57
        y = target.reshape(-1).long()
        r = eng.step_supervised(net, "ce", [x], y, train=True)
        net.optimizer_step(optimizer)
        meters.add(r["losses"], y.size(0))
    m = meters.meters()
    return m["loss"].avg, m["acc"].avg


def kather_sup_validate(args, model, classifier, val_loader, criterion, epoch):
    """eval_Kather_SSL.validate (:102-151) -> (loss_avg, acc_avg): eval-mode forward + CE + accuracy."""
    _plain_ce(criterion, "Kather validation (eval_Kather_SSL.py:102-151)")
    return kather_cr_validate(args, model, classifier, val_loader, epoch)


# ------------------------------------------------------------------------------------------------ WSI inference (f3)
def camelyon16_test(args, model, classifier, test_loader):
    """test_Camelyon16.test (test_Camelyon16.py:30-70) -> probs_map: every tissue pixel of ``test_loader.dataset.mask``
    gets the softmax 'tumor' probability of its tile; forward-only (BatchNorm folded, all epilogues fused)."""
    import numpy as np
    from . import kernels as K
    eng = get_engine(_device_of(model))
    model.eval()
    classifier.eval()
    net = eng.bind(model, classifier)
    probs_map = np.zeros(test_loader.dataset.mask.shape)
    t0 = time.time()
    for batch_idx, (input, x_mask, y_mask) in enumerate(_ahead(test_loader, eng, _prefetch_on(args))):
        _, output = net.forward((input,), train=False)
        probs = K.softmax_col(output.contiguous(), -1).cpu().numpy()        # second column 'tumor' (:58-60)
        probs_map[x_mask.numpy(), y_mask.numpy()] = probs
        if (batch_idx + 1) % 10 == 0 and getattr(args, "print_freq", 0):
            print("Test: [{0}/{1}]\tBT {2:.3f}".format(batch_idx, _len(test_loader), (time.time() - t0) / (batch_idx + 1)))
    return probs_map


def teacher_refresh(model_teacher, classifier_teacher, model_student, classifier_student, ema_decay=0.0):
    """In-place form of the reference's per-epoch ``teacher = copy.deepcopy(student)`` (eval_BreastPathQ_SSL_CR.py:515-516),
    generalised to an EMA (decay 0 == the reference).  ``copy.deepcopy`` itself also works on these modules."""
    eng = get_engine(_device_of(model_student))
    te, st = eng.bind(model_teacher, classifier_teacher), eng.bind(model_student, classifier_student)
    te.ema_from(st, ema_decay)
