"""Device-side weak augmentation ("next" row f1): the reference's ``TransformFix.weak`` -- ``RandomHorizontalFlip()`` then
``RandomCrop(size=image_size)`` (dataset.py:663-677) -- applied to a uint8 batch that already sits in HBM.

The random draws are made on the host in the order torchvision makes them for ONE sample (``torch.rand(1) < p`` for the
flip, then ``torch.randint(0, h - th + 1)`` and ``torch.randint(0, w - tw + 1)`` for the crop), sample after sample, so a
seeded run reproduces what the CPU transform would have produced sample by sample; the HIP kernel is the deterministic gather.
The output is the NCHW uint8 batch the stem kernel ingests directly.
"""
import torch

from . import _lib as L


def weak_params(n, src_hw, size, generator=None, p=0.5):
    """-> int32 [n, 3] (flip, top, left), drawn like n successive calls of TransformFix.weak."""
    sh, sw = src_hw
    th, tw = (size, size) if isinstance(size, int) else size
    if th > sh or tw > sw:
        raise ValueError(f"Required crop size {(th, tw)} is larger than input image size {(sh, sw)}")
    out = torch.empty((n, 3), dtype=torch.int32)
    for k in range(n):
        flip = bool(torch.rand(1, generator=generator) < p)                         # RandomHorizontalFlip.forward
        if sh == th and sw == tw:                                                   # RandomCrop.get_params
            top = left = 0
        else:
            top = int(torch.randint(0, sh - th + 1, size=(1,), generator=generator).item())
            left = int(torch.randint(0, sw - tw + 1, size=(1,), generator=generator).item())
        out[k, 0], out[k, 1], out[k, 2] = int(flip), top, left
    return out


def weak_augment(src_u8, params, size, *, src_hwc=False, out=None):
    """src uint8 [N,3,SH,SW] (or [N,SH,SW,3] with src_hwc) on the GPU, params int32 [N,3] -> uint8 [N,3,size,size]."""
    if not src_u8.is_cuda or src_u8.dtype != torch.uint8 or not src_u8.is_contiguous():
        raise ValueError("weak_augment: contiguous uint8 CUDA tensor expected")
    n = src_u8.shape[0]
    sh, sw = (src_u8.shape[1], src_u8.shape[2]) if src_hwc else (src_u8.shape[2], src_u8.shape[3])
    th, tw = (size, size) if isinstance(size, int) else size
    prm = params.to(device=src_u8.device, dtype=torch.int32).contiguous()
    if prm.shape != (n, 3):
        raise ValueError("params must be [N, 3]")
    dst = out if out is not None else torch.empty((n, 3, th, tw), dtype=torch.uint8, device=src_u8.device)
    d = L.WeakAugDesc(L.ptr(src_u8), L.ptr(dst), L.ptr(prm), n, sh, sw, th, tw, int(src_hwc))
    L.check(L.lib().sslcr_weak_augment(d, L.stream_ptr()))
    return dst


class TransformFixWeak:
    """Batched, device-side counterpart of ``TransformFix(image_size, N).weak`` (dataset.py:669)."""

    def __init__(self, image_size, generator=None):
        self.image_size, self.generator = image_size, generator

    def __call__(self, batch_u8, src_hwc=False):
        n = batch_u8.shape[0]
        hw = (batch_u8.shape[1], batch_u8.shape[2]) if src_hwc else (batch_u8.shape[2], batch_u8.shape[3])
        return weak_augment(batch_u8, weak_params(n, hw, self.image_size, self.generator), self.image_size, src_hwc=src_hwc)


def fix_params(n, src_hw, size, generator=None, p=0.5):
    """The torch-RNG draws of n successive ``TransformFix.__call__`` (dataset.py:663-677): per sample the weak branch's
    (flip, top, left) and then the strong branch's (flip, top, left) -- ``RandAugment`` draws from Python's ``random`` and numpy
    (models/randaugment.py:53,133-135), so it does not move the torch stream.  -> (weak int32 [n,3], strong int32 [n,3])."""
    weak = torch.empty((n, 3), dtype=torch.int32)
    strong = torch.empty((n, 3), dtype=torch.int32)
    for k in range(n):
        weak[k] = weak_params(1, src_hw, size, generator, p)[0]
        strong[k] = weak_params(1, src_hw, size, generator, p)[0]
    return weak, strong


class TransformFixGeometric:
    """Batched, device-side counterpart of the DETERMINISTIC part of ``TransformFix(image_size, N)`` (row f4, first half): both
    branches' ``RandomHorizontalFlip`` + ``RandomCrop`` with the reference's per-sample draw order, as two launches of the
    gather kernel over the source batch in HBM.  -> (weak uint8 [N,3,S,S], strong_geometric uint8 [N,3,S,S]); the histology
    ``RandAugment`` ops of the strong branch (models/randaugment.py:51-144) remain a host-side stage on the second output."""

    def __init__(self, image_size, generator=None):
        self.image_size, self.generator = image_size, generator

    def __call__(self, batch_u8, src_hwc=False):
        n = batch_u8.shape[0]
        hw = (batch_u8.shape[1], batch_u8.shape[2]) if src_hwc else (batch_u8.shape[2], batch_u8.shape[3])
        pw, ps = fix_params(n, hw, self.image_size, self.generator)
        return (weak_augment(batch_u8, pw, self.image_size, src_hwc=src_hwc),
                weak_augment(batch_u8, ps, self.image_size, src_hwc=src_hwc))


# ------------------------------------------------------------------------------------------------ strong branch, colour ops (row f4)
# skimage.color's stain matrices (scikit-image 0.15.0, requirements.txt:369): Ruifrok & Johnston's rgb_from_hed and its inverse
_RGB_FROM_HED = ((0.65, 0.70, 0.29), (0.07, 0.99, 0.11), (0.27, 0.57, 0.78))


def _hed_matrices():
    import numpy as np
    m = np.array(_RGB_FROM_HED, dtype=np.float64)
    return np.linalg.inv(m).reshape(-1).tolist(), m.reshape(-1).tolist()      # hed_from_rgb = linalg.inv(rgb_from_hed), as skimage builds it


def colour_shifts(rng):
    """the draws of ONE ``Color(img, v)`` call (models/randaugment.py:81-84 -> :30-32) from a ``random.Random``-like ``rng``: three
    ``uniform(-0.035, 0.035)`` standard deviations, then ``normalvariate(0, std)`` for h, d, e.  -> (hmod, dmod, emod)"""
    hs, ds, es = rng.uniform(-0.035, 0.035), rng.uniform(-0.035, 0.035), rng.uniform(-0.035, 0.035)
    return rng.normalvariate(0, hs), rng.normalvariate(0, ds), rng.normalvariate(0, es)


def _apply_mask(apply, n, device):
    if apply is None:
        return None
    a = torch.as_tensor(apply).to(device=device, dtype=torch.uint8).contiguous()
    if a.shape != (n,):
        raise ValueError("apply must be [N]")
    return a


def hed_colour_augment(batch_u8, shifts, apply=None, *, hwc=False, out=None):
    """``colour_augmentation`` (models/randaugment.py:17-48) on a uint8 batch in HBM: [N,3,H,W] (or [N,H,W,3] with hwc);
    shifts [N,3] = (hmod, dmod, emod) per image; apply [N] bool or None.  float64 arithmetic, skimage 0.15.0 order."""
    if not batch_u8.is_cuda or batch_u8.dtype != torch.uint8 or not batch_u8.is_contiguous() or batch_u8.dim() != 4:
        raise ValueError("hed_colour_augment: contiguous uint8 CUDA batch [N,3,H,W] or [N,H,W,3] expected")
    n = batch_u8.shape[0]
    h, w = (batch_u8.shape[1], batch_u8.shape[2]) if hwc else (batch_u8.shape[2], batch_u8.shape[3])
    if (batch_u8.shape[3] if hwc else batch_u8.shape[1]) != 3:
        raise ValueError("three colour channels expected")
    sh = torch.as_tensor(shifts, dtype=torch.float64).to(batch_u8.device).contiguous()
    if sh.shape != (n, 3):
        raise ValueError("shifts must be [N, 3]")
    ap = _apply_mask(apply, n, batch_u8.device)
    dst = out if out is not None else torch.empty_like(batch_u8)
    inv, fwd = _hed_matrices()
    d = L.ColourAugDesc(L.ptr(batch_u8), L.ptr(dst), L.ptr(sh), L.ptr(ap), (L.f64 * 9)(*inv), (L.f64 * 9)(*fwd), n, h, w, int(hwc))
    L.check(L.lib().sslcr_hed_colour_augment(d, L.stream_ptr()))
    return dst


def brightness_contrast_params(rng, brightness_limit=0.2, contrast_limit=0.2, p=0.5):
    """the draws of ONE ``Compose([RandomBrightnessContrast(...)])(image=img)`` call of albumentations 0.1.8 (what ``Brightness`` /
    ``Contrast`` of models/randaugment.py:93-103 build): Compose's ``random.random() < 1``, the transform's ``random.random() < p``,
    and only then alpha = 1 + uniform(contrast), beta = uniform(brightness).  -> (applied, alpha, beta)"""
    rng.random()
    if not rng.random() < p:
        return False, 1.0, 0.0
    # to_tuple(limit) of 0.1.8 is (-limit, limit) UNSORTED and get_params draws random.uniform(limit[0], limit[1]) = a + (b - a) r:
    # RandAugment's val = v / 30 * 0.4 - 0.2 is negative for v < 15, and the draw is then |val| (1 - 2 r), not |val| (2 r - 1)
    alpha = 1.0 + rng.uniform(-contrast_limit, contrast_limit)
    beta = 0.0 + rng.uniform(-brightness_limit, brightness_limit)
    return True, alpha, beta


def brightness_contrast(batch_u8, alpha_beta, apply=None, *, out=None):
    """albumentations 0.1.8 ``brightness_contrast_adjust`` on a uint8 batch in HBM ([N,3,H,W] or [N,H,W,3]; the map does not care):
    clip(float32(img) * alpha + beta * mean(img), 0, max(img)).astype(uint8) with per-image mean / max.  alpha_beta [N,2]."""
    if not batch_u8.is_cuda or batch_u8.dtype != torch.uint8 or not batch_u8.is_contiguous() or batch_u8.dim() != 4:
        raise ValueError("brightness_contrast: contiguous uint8 CUDA batch expected")
    n = batch_u8.shape[0]
    ab = torch.as_tensor(alpha_beta, dtype=torch.float64).to(batch_u8.device).contiguous()
    if ab.shape != (n, 2):
        raise ValueError("alpha_beta must be [N, 2]")
    ap = _apply_mask(apply, n, batch_u8.device)
    dst = out if out is not None else torch.empty_like(batch_u8)
    stats = torch.empty((n, 2), dtype=torch.int64, device=batch_u8.device)
    pix = batch_u8.numel() // (3 * n)
    d = L.BrightnessContrastDesc(L.ptr(batch_u8), L.ptr(dst), L.ptr(ab), L.ptr(ap), L.ptr(stats), n, pix, 1)
    L.check(L.lib().sslcr_brightness_contrast(d, L.stream_ptr()))
    return dst


class RandAugmentDevice:
    """Batched counterpart of ``RandAugment(n, m)`` (models/randaugment.py:119-144) for a uint8 NCHW batch that sits in HBM.

    Per image the reference draws ``ops = random.choices(augment_pool, k=n)`` and, per op, ``v = np.random.randint(1, m)``; the op
    then makes its own draws from ``random``.  Here the same draws are made on the host, image after image and op after op, from
    the generators passed in; the ops with restatable arithmetic -- Color, Brightness, Contrast -- run on the device, batched per op
    slot with an apply mask.  The six geometric / blur / noise / HSV ops are albumentations 0.1.8 + OpenCV code that is not
    installed here and cannot be pinned: they go through ``host_ops[name](img_hwc_uint8_numpy, val) -> numpy`` if given
    (the image makes a round trip through host memory for that slot), else NotImplementedError names the op.  A host op draws from
    the module-level generators when it RUNS (after the whole batch has been planned), so a batch that mixes host ops in does not
    consume the ``random`` stream in the reference's image-by-image order; batches served by the device ops alone do."""

    POOL = (("HSV", -1, 1), ("Noise", 0, 0.15), ("Scale_Resize_Crop", 0.8, 1.2), ("Shift_Scale_Rotate", 0.01, 0.1),
            ("Color", -0.035, 0.035), ("Blur_img", 0, 2), ("Brightness", -0.2, 0.2), ("Contrast", -0.2, 0.2), ("Rotate_Crop", -90, 90))

    def __init__(self, n, m, rng, np_rng, host_ops=None):
        self.n, self.m, self.rng, self.np_rng, self.host_ops = n, m, rng, np_rng, host_ops or {}

    def __call__(self, batch_u8):
        N = batch_u8.shape[0]
        cur = batch_u8
        # the reference finishes one image (all its ops, in order) before it draws for the next: draw everything first, in that order
        plan = []
        for _ in range(N):
            ops = self.rng.choices(self.POOL, k=self.n)
            row = []
            for name, lo, hi in ops:
                v = int(self.np_rng.randint(1, self.m))
                val = (float(v) / 30) * float(hi - lo) + lo
                if name == "Color":
                    row.append((name, colour_shifts(self.rng)))
                elif name == "Brightness":
                    row.append((name, brightness_contrast_params(self.rng, brightness_limit=val)))
                elif name == "Contrast":
                    row.append((name, brightness_contrast_params(self.rng, contrast_limit=val)))
                else:
                    if name not in self.host_ops:
                        raise NotImplementedError(f"RandAugment op {name} (albumentations 0.1.8) has no device kernel; pass host_ops[{name!r}]")
                    row.append((name, val))      # the host op makes its own draws from the module-level generators when it runs
            plan.append(row)
        for slot in range(self.n):
            names = [plan[i][slot][0] for i in range(N)]
            if "Color" in names:
                sh = [plan[i][slot][1] if names[i] == "Color" else (0.0, 0.0, 0.0) for i in range(N)]
                cur = hed_colour_augment(cur, sh, [nm == "Color" for nm in names])
            if "Brightness" in names or "Contrast" in names:
                bc = [nm in ("Brightness", "Contrast") and plan[i][slot][1][0] for i, nm in enumerate(names)]
                ab = [plan[i][slot][1][1:] if bc[i] else (1.0, 0.0) for i in range(N)]
                if any(bc):
                    cur = brightness_contrast(cur, ab, bc)
            host = [i for i, nm in enumerate(names) if nm not in ("Color", "Brightness", "Contrast")]
            if host:
                if cur is batch_u8:
                    cur = batch_u8.clone()
                for i in host:
                    img = cur[i].permute(1, 2, 0).contiguous().cpu().numpy()
                    res = self.host_ops[names[i]](img, plan[i][slot][1])
                    if isinstance(res, dict):
                        res = res["image"]
                    cur[i] = torch.from_numpy(res).permute(2, 0, 1).to(cur.device)
        return cur
