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
