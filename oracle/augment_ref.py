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


# ------------------------------------------------------------------------------------------------ strong branch, colour ops (row f4)
# The reference's RandAugment pool (models/randaugment.py:105-117) holds nine ops.  Two families run arithmetic that can be
# restated exactly: colour_augmentation (:17-48, the reference's own code on top of scikit-image 0.15.0's rgb2hed / hed2rgb) and
# Brightness / Contrast (:93-103, albumentations 0.1.8 RandomBrightnessContrast).  Neither library is installed in the build
# container (no network), so both restatements follow the PUBLISHED source of the pinned versions (requirements.txt:369 and :10):
# PARITY UNPINNED against the libraries themselves; the reference's own lines (:19-47) are followed statement by statement.
import numpy as np

# skimage/color/colorconv.py (0.15.0): "Haematoxylin-Eosin-DAB colorspace", Ruifrok & Johnston
RGB_FROM_HED = np.array([[0.65, 0.70, 0.29],
                         [0.07, 0.99, 0.11],
                         [0.27, 0.57, 0.78]])
HED_FROM_RGB = np.linalg.inv(RGB_FROM_HED)


def rgb2hed(rgb_u8):
    """skimage 0.15.0 separate_stains(rgb, hed_from_rgb): img_as_float (uint8 * (1/255), dtype.py convert()), rgb += 2,
    stains = -log(rgb) . conv_matrix."""
    rgb = np.multiply(rgb_u8, 1.0 / 255, dtype=np.float64)
    rgb += 2
    return np.reshape(np.dot(np.reshape(-np.log(rgb), (-1, 3)), HED_FROM_RGB), rgb.shape)


def hed2rgb(hed):
    """skimage 0.15.0 combine_stains(hed, rgb_from_hed): exp(-stains . conv_matrix) - 2, then
    rescale_intensity(..., in_range=(-1, 1)) whose out_range for a float image is (-1, 1): clip, normalise, scale back."""
    logrgb2 = np.dot(-np.reshape(hed, (-1, 3)), RGB_FROM_HED)
    rgb2 = np.exp(logrgb2)
    image = np.reshape(rgb2 - 2, hed.shape)
    imin, imax, omin, omax = -1.0, 1.0, -1.0, 1.0
    image = np.clip(image, imin, imax)
    image = (image - imin) / float(imax - imin)
    return image * (omax - omin) + omin


def colour_augmentation(image_hwc_u8, hmod, dmod, emod):
    """models/randaugment.py:17-48 with the three random.normalvariate draws (:30-32) passed in.  The reference's per-pixel loop
    (:35-38) adds the same three scalars to every pixel: a vector add."""
    ihc_hed = rgb2hed(image_hwc_u8)
    zdh = ihc_hed + np.array([hmod, dmod, emod], dtype=np.float64)
    zdh = hed2rgb(zdh)
    with np.errstate(invalid="ignore"):
        # (zdh * 255).astype('uint8') (:45): C cast -- truncation toward zero, then the low byte (values outside [0, 256) wrap)
        return (zdh * 255).astype(np.int64).astype(np.uint8)


def draw_colour_shifts(rng):
    """the draws of ONE Color() call (models/randaugment.py:81-84, then :30-32) from Python's `random` module state `rng`:
    three uniform(-0.035, 0.035) standard deviations (argument evaluation order h, d, e), then three normalvariate(0, std)."""
    hs, ds, es = rng.uniform(-0.035, 0.035), rng.uniform(-0.035, 0.035), rng.uniform(-0.035, 0.035)
    return rng.normalvariate(0, hs), rng.normalvariate(0, ds), rng.normalvariate(0, es)


def brightness_contrast_adjust(img_u8, alpha, beta):
    """albumentations 0.1.8 augmentations/functional.py: @clipped brightness_contrast_adjust -- dtype and np.max(img) are taken
    from the input, the result is np.clip(., 0, maxval).astype(dtype).  Written with explicit float32 roundings (what numpy 1.x's
    value-based casting makes of `float32 array * python float + float64 scalar`), so it does not depend on the numpy version."""
    maxval = np.float32(np.max(img_u8))
    x = img_u8.astype(np.float32) * np.float32(alpha) + np.float32(beta * np.mean(img_u8))
    return np.clip(x, np.float32(0), maxval).astype(np.int64).astype(np.uint8)


def draw_brightness_contrast(rng, brightness_limit=0.2, contrast_limit=0.2, p=0.5):
    """the draws of ONE Compose([RandomBrightnessContrast(...)])(image=img) call in albumentations 0.1.8: Compose's own
    random.random() < 1.0, the transform's random.random() < p, and -- only if that fires -- alpha = 1 + uniform(contrast),
    beta = 0 + uniform(brightness) (get_params order).  -> (applied, alpha, beta)"""
    rng.random()
    if not rng.random() < p:
        return False, 1.0, 0.0
    # to_tuple(limit) of 0.1.8 is (-limit, limit) UNSORTED and get_params draws random.uniform(limit[0], limit[1]) = a + (b - a) r:
    # RandAugment's val = v / 30 * 0.4 - 0.2 is negative for v < 15, and the draw is then |val| (1 - 2 r), not |val| (2 r - 1)
    alpha = 1.0 + rng.uniform(-contrast_limit, contrast_limit)
    beta = 0.0 + rng.uniform(-brightness_limit, brightness_limit)
    return True, alpha, beta
