"""CPU restatement of the reference's weak augmentation (test oracle -- only tests/, smoke() and bench's cpu_baseline may
import this package).  ``TransformFix.weak`` = ``transforms.Compose([RandomHorizontalFlip(), RandomCrop(size)])``
(dataset.py:663-677); torchvision's functional forms are ``hflip(img) = img.flip(-1)`` and
``crop(img, top, left, h, w) = img[..., top:top+h, left:left+w]``, applied in that order.

Parity note: torchvision is not installed in the build container, so the draw ORDER (flip: ``torch.rand(1) < p``; crop:
``randint(0, h-th+1)`` then ``randint(0, w-tw+1)``) is restated from its published source rather than pinned against it;
the pixel mapping given (flip, top, left) is exact.
"""
import torch


def weak_one(img_chw, flip, top, left, size):
    """img uint8 [3,SH,SW] -> [3,size,size]: hflip (if flip) then crop, like Compose([RandomHorizontalFlip, RandomCrop])."""
    th, tw = (size, size) if isinstance(size, int) else size
    if flip:
        img_chw = img_chw.flip(-1)
    return img_chw[..., top:top + th, left:left + tw]


def weak_batch(src_nchw, params, size):
    return torch.stack([weak_one(src_nchw[k], bool(params[k, 0]), int(params[k, 1]), int(params[k, 2]), size)
                        for k in range(src_nchw.shape[0])])


def draw_params(n, src_hw, size, generator=None, p=0.5):
    """The draws n successive ``TransformFix.weak`` calls make: per sample rand(1) < p, then randint for top, then for left
    (no crop draws when the source already has the target size -- RandomCrop.get_params returns (0, 0) without drawing)."""
    sh, sw = src_hw
    th, tw = (size, size) if isinstance(size, int) else size
    rows = []
    for _ in range(n):
        flip = int(torch.rand(1, generator=generator).item() < p)
        if (sh, sw) == (th, tw):
            top = left = 0
        else:
            top = int(torch.randint(0, sh - th + 1, size=(1,), generator=generator).item())
            left = int(torch.randint(0, sw - tw + 1, size=(1,), generator=generator).item())
        rows.append((flip, top, left))
    return torch.tensor(rows, dtype=torch.int32)


def draw_fix_params(n, src_hw, size, generator=None, p=0.5):
    """The torch-RNG draws of n successive ``TransformFix.__call__`` calls (dataset.py:672-677: ``self.weak(x)`` then
    ``self.strong(x)``, whose RandAugment stage draws from Python's ``random`` / numpy, not torch): per sample
    (flip, top, left) of the weak branch, then of the strong branch."""
    weak, strong = [], []
    for _ in range(n):
        weak.append(draw_params(1, src_hw, size, generator, p)[0])
        strong.append(draw_params(1, src_hw, size, generator, p)[0])
    return torch.stack(weak), torch.stack(strong)
