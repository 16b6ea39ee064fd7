"""Seeded inputs + the golden case table shared by tests/golden/make_golden.py (which drives
the REFERENCE's train()/validate()) and the tests (which drive the oracle and the HIP engine).

Inputs follow the reference's batch contract: uint8 NCHW in [0,255], no
normalisation; labeled batches ``[b,3,3,H,W]`` + ``[b,3]`` (dataset.py:487-536), unlabeled
``([mu*b,3,H,W],[mu*b,3,H,W])`` (dataset.py:624-677), RSP ``3x[B,3,H,W] + [B,1]``
(dataset.py:166-213).  numpy's legacy RandomState is bit-stable across versions.
"""
import numpy as np
import torch


def u8(seed, shape):
    return torch.from_numpy(np.random.RandomState(seed).randint(0, 256, size=shape, dtype=np.uint8))


def f32(seed, shape):
    return torch.from_numpy(np.random.RandomState(seed).uniform(0.0, 1.0, size=shape).astype(np.float32))


def ints(seed, shape, hi):
    return torch.from_numpy(np.random.RandomState(seed).randint(0, hi, size=shape).astype(np.int64))


# name -> config.  hw = image side; b = --batch_size; mu = --mu; nb = batches per epoch
CASES = {
    # eval_BreastPathQ_SSL_CR.train: MSE/MSE, Adam lr 1e-4 wd 1e-4 (:268-272,481); 256 is hard-coded (:74)
    "bpq_cr_f60": dict(script="bpq_cr", hw=256, b=1, mu=2, nb=2, modules=60, classes=1, lr=1e-4, wd=1e-4,
                       lambda_u=1.0, opt="adam"),
    "bpq_cr_f0": dict(script="bpq_cr", hw=256, b=1, mu=2, nb=2, modules=0, classes=1, lr=1e-4, wd=1e-4,
                      lambda_u=1.0, opt="adam"),
    # the BASELINE.json / bench.py workload itself, ONE iteration: per-GPU --batch_size 64 --mu 7, full fine-tune
    # (student 192 + 448 images, teacher 448).  Its golden holds reductions only (losses, feature sums, per-parameter
    # gradient norms and seeded +-1 projections), see tests/golden/make_golden.py:gen_bpq_cr_full
    "bpq_cr_full": dict(script="bpq_cr", hw=256, b=64, mu=7, nb=1, modules=0, classes=1, lr=1e-4, wd=1e-4,
                        lambda_u=1.0, opt="adam"),
    # eval_Camelyon_SSL_CR.train: CE + hard pseudo-label CE, SGD-Nesterov lr 5e-4 (:251-256,514)
    "cam_cr_f60": dict(script="cam_cr", hw=64, b=2, mu=2, nb=2, modules=60, classes=2, lr=5e-4, wd=1e-4,
                       lambda_u=1.0, opt="sgd"),
    "cam_cr_f0": dict(script="cam_cr", hw=64, b=2, mu=2, nb=2, modules=0, classes=2, lr=5e-4, wd=1e-4,
                      lambda_u=0.5, opt="sgd"),
    # eval_Kather_SSL_CR.train/validate: 9-class CE + hard pseudo-label CE, Adam (:527); 256 is hard-coded (:68)
    "kather_cr_f0": dict(script="kather_cr", hw=256, b=1, mu=2, nb=2, modules=0, classes=9, lr=1e-4, wd=1e-4,
                         lambda_u=1.0, opt="adam"),
    # pretrain_BreastPathQ.train/validate: RSP 6-way CE, SGD-Nesterov lr .01 + Lookahead(5,.5) (:245-247)
    "rsp": dict(script="rsp", hw=64, b=4, nb=2, classes=6, lr=0.01, wd=1e-4, opt="sgd"),
    # one Camelyon SSL_CR iteration (CE + hard pseudo-label CE, SGD-Nesterov) at the size of the benchmark workload:
    # 2 x 32 x 3 labeled + 2 x 224 weak/strong unlabeled 256x256 patches (student 640, teacher 448); reductions only
    "cam_cr_full": dict(script="cam_cr", hw=256, b=32, mu=7, nb=1, modules=0, classes=2, lr=5e-4, wd=1e-4,
                        lambda_u=1.0, opt="sgd"),
    # one RSP iteration at full size (pretrain_BreastPathQ.py defaults: --batch_size 128, 256x256 tiles; 3 x 128 images);
    # golden = reductions only, tests/golden/make_golden.py:gen_rsp_full
    "rsp_full": dict(script="rsp", hw=256, b=128, nb=1, classes=6, lr=0.01, wd=1e-4, opt="sgd"),
    # eval_Camelyon_SSL.train: supervised CE (student only), SGD-Nesterov (:371)
    "cam_sup": dict(script="cam_sup", hw=64, b=2, nb=2, modules=0, classes=2, lr=1e-3, wd=1e-4, opt="sgd"),
    # eval_BreastPathQ_SSL.train: supervised MSE, Adam (:396); image side is args.image_size (:58)
    "bpq_sup": dict(script="bpq_sup", hw=64, b=2, nb=2, modules=0, classes=1, lr=1e-3, wd=1e-4, opt="adam"),
    # eval_Kather_SSL.train/validate (:32-99 / :102-151): supervised 9-class CE, Adam lr 1e-5 (:234,419), image side is
    # args.image_size (:57).  96x96 is NOT 16-tileable past the stem (24/12/6/3 maps): the engine's fallback conv shapes
    "kather_sup": dict(script="kather_sup", hw=96, b=2, nb=2, modules=0, classes=9, lr=1e-5, wd=1e-4, opt="adam"),
    # BASELINE.json config 1 itself: --batch_size 32, 224x224, 9 classes -> [32,3,3,224,224] -> 96 images (56/28/14/7 maps);
    # ONE iteration, reductions only (tests/golden/make_golden.py:gen_kather_sup_full)
    "kather_sup_full": dict(script="kather_sup", hw=224, b=32, nb=1, modules=0, classes=9, lr=1e-5, wd=1e-4, opt="adam"),
    # multi-iteration trajectories (bf16-fidelity yardstick): the reference's train() called once per iteration with ONE batch,
    # same optimizer object throughout, 256x256, full fine-tune; per-iteration returned losses + final snapshot
    "traj_bpq_cr": dict(script="bpq_cr", hw=256, b=2, mu=3, nb=1, iters=24, modules=0, classes=1, lr=1e-4, wd=1e-4,
                        lambda_u=1.0, opt="adam"),
    "traj_cam_cr": dict(script="cam_cr", hw=256, b=1, mu=3, nb=1, iters=24, modules=0, classes=2, lr=5e-4, wd=1e-4,
                        lambda_u=1.0, opt="sgd"),
    # checkpoint layouts (row f2): epoch 1 -> the reference's save dict -> torch.save -> fresh modules -> the reference's
    # --resume / load code -> epoch 2.  Frozen backbones (--modules 62: fc.2 + the classifier train) keep the fixture small (only those tensors, the BatchNorm buffers and their optimizer state move).
    "ckpt_bpq_cr": dict(script="bpq_cr", hw=256, b=1, mu=2, nb=2, modules=62, classes=1, lr=1e-4, wd=1e-4,
                        lambda_u=1.0, opt="adam"),
    "ckpt_cam_sup": dict(script="cam_sup", hw=64, b=2, nb=2, modules=62, classes=2, lr=1e-3, wd=1e-4, opt="sgd"),
    # (one iteration per epoch, 128x128: with lr 0.01 SGD-Nesterov momentum and BatchNorm over a handful of values, ReLU-mask
    # flips between two fp32 implementations get amplified by every further iteration -- measured 1e-5 -> 5e-2 over four
    # iterations at 64x64 with torch-CPU fp32 itself showing the same flips against float64)
    "ckpt_rsp": dict(script="rsp", hw=128, b=4, nb=1, classes=6, lr=0.01, wd=1e-4, opt="sgd"),
    # test_Camelyon16.test: forward-only WSI tile classification -> tumour-probability map ("next" row f3); the loader
    # yields (float32 RGB tile batch, x_mask, y_mask) for the tissue pixels of a mask, last batch ragged (:41-66)
    "cam_wsi": dict(script="cam_wsi", hw=64, b=4, classes=2, mask=(6, 5)),
    # a slide-sized run of the same function: 26 x 19 mask (~220 tissue pixels) of 128x128 tiles in batches of 32, ragged tail
    "cam_wsi_large": dict(script="cam_wsi", hw=128, b=32, classes=2, mask=(26, 19), head_scale=0.05),
}

PARAM_SEED = 42        # the reference's default --seed (eval_BreastPathQ_SSL_CR.py:253)


def grad_probe(idx, numel):
    """seeded +-1 vector for parameter idx: <grad, probe> pins the DIRECTION of a gradient in one float"""
    return torch.from_numpy(np.random.RandomState(9000 + idx).randint(0, 2, numel).astype(np.float64) * 2.0 - 1.0)


def labeled_batches(case, seed0=1000):
    """BreastPathQ-style labeled loader: [(x u8 [b,3,3,H,W], y f32 [b,3])] * nb."""
    c = CASES[case]
    return [(u8(seed0 + i, (c["b"], 3, 3, c["hw"], c["hw"])), f32(seed0 + 50 + i, (c["b"], 3)))
            for i in range(c["nb"])]


def labeled_batches_cls(case, seed0, label):
    """Camelyon-style class loader: x u8 [b,3,3,H,W], y int64 [b,3] all == label."""
    c = CASES[case]
    return [(u8(seed0 + i, (c["b"], 3, 3, c["hw"], c["hw"])),
             torch.full((c["b"], 3), label, dtype=torch.int64)) for i in range(c["nb"])]


def labeled_batches_kather(case, seed0=1000):
    """Kather labeled loader: (x u8 [b,3,3,H,W], y int64 [b,3] in 0..8)."""
    c = CASES[case]
    return [(u8(seed0 + i, (c["b"], 3, 3, c["hw"], c["hw"])), ints(seed0 + 50 + i, (c["b"], 3), c["classes"]))
            for i in range(c["nb"])]


def sup_batches_kather(case, seed0=1000):
    """eval_Kather_SSL labeled loader (DatasetKather_Supervised_train, dataset.py): x u8 [b,3,3,H,W], y int64 [b,3]."""
    return labeled_batches_kather(case, seed0)


def val_batches_kather(case, seed0=4000):
    c = CASES[case]
    return [(u8(seed0 + i, (2, 3, c["hw"], c["hw"])), ints(seed0 + 50 + i, (2,), c["classes"])) for i in range(2)]


def unlabeled_batches(case, seed0=2000):
    c = CASES[case]
    n = c["b"] * c["mu"]
    return [(u8(seed0 + i, (n, 3, c["hw"], c["hw"])), u8(seed0 + 50 + i, (n, 3, c["hw"], c["hw"])))
            for i in range(c["nb"])]


def rsp_batches(case, seed0=3000):
    c = CASES[case]
    return [(u8(seed0 + i, (c["b"], 3, c["hw"], c["hw"])), u8(seed0 + 20 + i, (c["b"], 3, c["hw"], c["hw"])),
             u8(seed0 + 40 + i, (c["b"], 3, c["hw"], c["hw"])), ints(seed0 + 60 + i, (c["b"], 1), 6).to(torch.uint8))
            for i in range(c["nb"])]


def val_batches_reg(case, seed0=4000):
    """BreastPathQ validate(): (input [n,3,H,W] u8, target [n] f32) (dataset.py eval contract)."""
    c = CASES[case]
    return [(u8(seed0 + i, (2, 3, c["hw"], c["hw"])), f32(seed0 + 50 + i, (2,))) for i in range(2)]


def val_batches_cls(case, seed0, label):
    c = CASES[case]
    return [(u8(seed0 + i, (2, 3, c["hw"], c["hw"])), torch.full((2,), label, dtype=torch.int64))
            for i in range(2)]


class WsiLoader:
    """What test_Camelyon16.test() needs of its DataLoader: iteration over (input, x_mask, y_mask) batches, ``len``, and
    ``.dataset.mask`` (the tissue mask whose shape the probability map takes; dataset.py:958-972)."""

    def __init__(self, mask, batches):
        self.dataset = type("WsiDataset", (), {"mask": mask})()
        self._batches = batches

    def __iter__(self):
        return iter(self._batches)

    def __len__(self):
        return len(self._batches)


def wsi_loader(case="cam_wsi", seed0=6000):
    """seeded tissue mask + tile batches in the order DatasetCamelyon16_test enumerates them (np.where(mask), row-major);
    tiles are float32 RGB 0..255 like ``np.array(img, dtype=np.float32).transpose(2,0,1)`` (dataset.py:991-993)."""
    c = CASES[case]
    rs = np.random.RandomState(seed0)
    mask = rs.rand(*c["mask"]) < 0.45
    mask[0, 0] = True                           # at least one tissue pixel, and an edge one
    xs, ys = np.where(mask)
    batches = []
    for i in range(0, len(xs), c["b"]):
        n = min(c["b"], len(xs) - i)
        tiles = u8(seed0 + 10 + i, (n, 3, c["hw"], c["hw"])).float()
        batches.append((tiles, torch.from_numpy(xs[i:i + n].copy()), torch.from_numpy(ys[i:i + n].copy())))
    return WsiLoader(mask, batches)
