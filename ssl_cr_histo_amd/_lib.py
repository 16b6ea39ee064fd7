"""ctypes binding of libsslcr.so -- the C-ABI declared in include/sslcr.h.

There is NO fallback: if the HIP library is missing or a call fails, this raises.  ``import torch`` must
precede loading so that the process uses a single HIP runtime (torch's libamdhip64.so.7 soname).
"""
import ctypes as C
import os

import torch  # noqa: F401  (loads libamdhip64 / librccl first)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsslcr.so")

vp, i32, f32, f64, sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t


class SslcrError(RuntimeError):
    pass


def _S(name, fields):
    return type(name, (C.Structure,), {"_fields_": fields})


ConvDesc = _S("ConvDesc", [("x", vp), ("w", vp), ("y", vp), ("in_scale", vp), ("in_shift", vp), ("bias", vp),
                           ("residual", vp), ("stats", vp)] +
              [(k, i32) for k in ("N", "H", "W", "C", "K", "R", "S", "stride", "pad", "PH", "PW", "OH", "OW", "osh",
                                  "transposed", "in_relu", "relu", "accumulate", "pix_mul", "pix_off_h", "pix_off_w")] +
              [("tap_mask", C.c_uint), ("mask_x", vp), ("mask_scale", vp), ("mask_shift", vp), ("mask_mean", vp),
               ("par4", i32), ("seg_images", i32), ("seg_stride", i32), ("out_scale", vp)])
WgradDesc = _S("WgradDesc", [("x", vp), ("dy", vp), ("dw", vp), ("in_scale", vp), ("in_shift", vp), ("in_relu", i32)] +
               [(k, i32) for k in ("N", "H", "W", "C", "K", "R", "S", "stride", "pad", "OH", "OW", "seg_images", "seg_stride")])
StemDesc = _S("StemDesc", [("x", vp), ("w", vp), ("y", vp), ("bias", vp), ("stats", vp)] +
              [(k, i32) for k in ("N", "H", "W", "OH", "OW", "in_f32", "relu")] + [("x2", vp), ("n_split", i32), ("out_scale", vp)])
StemWgradDesc = _S("StemWgradDesc", [("x", vp), ("dy", vp), ("dw", vp)] +
                   [(k, i32) for k in ("N", "H", "W", "OH", "OW", "in_f32")] + [("x2", vp), ("n_split", i32)])
BnFinalizeDesc = _S("BnFinalizeDesc", [("partials", vp), ("rows", i32), ("C", i32), ("count", f64), ("gamma", vp),
                                       ("beta", vp), ("scale", vp), ("shift", vp), ("mean", vp), ("invstd", vp),
                                       ("running_mean", vp), ("running_var", vp), ("num_batches_tracked", vp),
                                       ("momentum", f32), ("eps", f32), ("replay", i32), ("sums_out", vp),
                                       ("sums_in", vp), ("stage", vp), ("nseg", i32), ("seg_stride", i32), ("tickets", vp)])
BnActDesc = _S("BnActDesc", [("x", vp), ("scale", vp), ("shift", vp), ("res", vp), ("rscale", vp), ("rshift", vp),
                             ("y", vp), ("pixels", sz), ("C", i32), ("relu", i32), ("nseg", i32), ("seg_stride", i32), ("ybits", vp)])
PoolFwdDesc = _S("PoolFwdDesc", [("x", vp), ("scale", vp), ("shift", vp), ("y", vp), ("argmax", vp)] +
                 [(k, i32) for k in ("N", "H", "W", "C", "OH", "OW")])
PoolBwdDesc = _S("PoolBwdDesc", [("dy", vp), ("argmax", vp), ("x", vp), ("scale", vp), ("shift", vp), ("dx", vp)] +
                 [(k, i32) for k in ("N", "H", "W", "C", "OH", "OW")])
BnBwdDesc = _S("BnBwdDesc", [("dy", vp), ("x", vp), ("yact", vp), ("scale", vp), ("shift", vp), ("mean", vp),
                             ("invstd", vp), ("sums", vp), ("dx", vp), ("gout", vp), ("pixels", sz), ("C", i32),
                             ("relu_from_x", i32), ("count", f64), ("pool_dy", vp), ("pool_argmax", vp), ("pH", i32),
                             ("pW", i32), ("pOH", i32), ("pOW", i32), ("pool_y", vp), ("g_in_reduce", i32), ("dgamma", vp), ("dbeta", vp),
                             ("pg_scale", f32), ("nseg", i32), ("seg_stride", i32), ("sums_stride", i32), ("yact_bits", vp)])
LossDesc = _S("LossDesc", [("kind", i32), ("logits", vp), ("logits_t", vp), ("target_f", vp), ("target_i", vp),
                           ("dlogits", vp), ("out", vp), ("nx", i32), ("nu", i32), ("C", i32), ("lambda_u", f32),
                           ("inv_nx_global", f32), ("inv_nu_global", f32)])
TensorDesc = _S("TensorDesc", [("p", vp), ("g", vp), ("s1", vp), ("s2", vp), ("n", i32), ("K", i32), ("C", i32),
                               ("RS", i32), ("w_fwd", vp), ("w_dgrad", vp), ("pack_dtype", i32), ("dgrad_flip", i32)])
OptDesc = _S("OptDesc", [("kind", i32), ("lr", f32), ("beta1", f32), ("beta2", f32), ("eps", f32), ("wd", f32),
                         ("momentum", f32), ("bc1", f32), ("bc2", f32), ("first_step", i32), ("grad_scale", f32)])
WeakAugDesc = _S("WeakAugDesc", [("src", vp), ("dst", vp), ("params", vp)] +
                 [(k, i32) for k in ("N", "SH", "SW", "OH", "OW", "src_hwc")])
ColourAugDesc = _S("ColourAugDesc", [("src", vp), ("dst", vp), ("shift", vp), ("apply", vp), ("hed_from_rgb", f64 * 9),
                                     ("rgb_from_hed", f64 * 9), ("N", i32), ("H", i32), ("W", i32), ("hwc", i32)])
BrightnessContrastDesc = _S("BrightnessContrastDesc", [("src", vp), ("dst", vp), ("alpha_beta", vp), ("apply", vp), ("stats", vp),
                                                       ("N", i32), ("H", i32), ("W", i32)])
PackDesc = _S("PackDesc", [("w", vp), ("w_fwd", vp), ("w_dgrad", vp), ("gamma", vp), ("beta", vp), ("rmean", vp),
                           ("rvar", vp), ("eps", f32), ("bias_out", vp)] + [(k, i32) for k in ("K", "C", "R", "S", "dgrad_flip")] + [("scale_out", vp)])

Fp8Desc = _S("Fp8Desc", [("w8", vp), ("w_dequant", vp), ("x_scale", f32), ("x_scale_dev", vp), ("amax_out", vp)])
PackFp8Desc = _S("PackFp8Desc", [("w", vp), ("w8", vp), ("w_dequant", vp), ("gamma", vp), ("beta", vp), ("rmean", vp), ("rvar", vp),
                                 ("eps", f32), ("bias_out", vp), ("K", i32), ("C", i32)])

# symbol -> (restype, argtypes); every symbol include/sslcr.h declares
P = C.POINTER
SIGNATURES = {
    "sslcr_version": (i32, []),
    "sslcr_last_error": (C.c_char_p, []),
    "sslcr_conv2d": (i32, [i32, P(ConvDesc), vp]),
    "sslcr_conv2d_partial_rows": (i32, [P(ConvDesc)]),
    "sslcr_conv2d_segments_ok": (i32, [i32, P(ConvDesc)]),
    "sslcr_conv2d_kernel_name": (C.c_char_p, [i32, P(ConvDesc)]),
    "sslcr_conv2d_s2_pair": (i32, [i32, P(ConvDesc), P(ConvDesc), vp]),
    "sslcr_conv2d_s2_pair_ok": (i32, [i32, P(ConvDesc), P(ConvDesc)]),
    "sslcr_conv2d_fp8": (i32, [P(ConvDesc), P(Fp8Desc), vp]),
    "sslcr_conv2d_fp8_partial_rows": (i32, [P(ConvDesc)]),
    "sslcr_pack_conv_fp8": (i32, [P(PackFp8Desc), vp]),
    "sslcr_conv2d_wgrad": (i32, [i32, P(WgradDesc), vp]),
    "sslcr_conv2d_wgrad_kernel_name": (C.c_char_p, [i32, P(WgradDesc)]),
    "sslcr_probe_tr16": (i32, [vp, vp, vp, vp]),
    "sslcr_stem_conv": (i32, [i32, P(StemDesc), vp]),
    "sslcr_stem_partial_rows": (i32, [P(StemDesc)]),
    "sslcr_stem_conv_pool": (i32, [i32, P(StemDesc), i32, i32, vp]),
    "sslcr_stem_wgrad": (i32, [i32, P(StemWgradDesc), vp]),
    "sslcr_stem_wgrad_pool": (i32, [i32, P(StemWgradDesc), P(BnBwdDesc), vp]),
    "sslcr_bn_finalize": (i32, [P(BnFinalizeDesc), vp]),
    "sslcr_bn_act": (i32, [i32, P(BnActDesc), vp]),
    "sslcr_bn_relu_maxpool": (i32, [i32, P(PoolFwdDesc), vp]),
    "sslcr_maxpool_relu_bwd": (i32, [i32, P(PoolBwdDesc), vp]),
    "sslcr_avgpool_fwd": (i32, [i32, vp, vp, i32, i32, i32, vp]),
    "sslcr_avgpool_bwd": (i32, [i32, vp, vp, i32, i32, i32, vp]),
    "sslcr_bn_bwd_reduce": (i32, [i32, P(BnBwdDesc), vp]),
    "sslcr_bn_bwd_apply": (i32, [i32, P(BnBwdDesc), vp]),
    "sslcr_bn_param_grads": (i32, [vp, vp, vp, vp, i32, vp]),
    "sslcr_linear_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "sslcr_linear_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "sslcr_loss": (i32, [P(LossDesc), vp]),
    "sslcr_softmax_col": (i32, [vp, vp, i32, i32, i32, vp]),
    "sslcr_optimizer_step": (i32, [vp, i32, i32, P(OptDesc), vp]),
    "sslcr_axpby": (i32, [vp, vp, sz, f32, i32, vp]),
    "sslcr_fill": (i32, [vp, sz, f32, vp]),
    "sslcr_pack_conv": (i32, [i32, P(PackDesc), vp]),
    "sslcr_pack_stem": (i32, [i32, P(PackDesc), vp]),
    "sslcr_weak_augment": (i32, [P(WeakAugDesc), vp]),
    "sslcr_hed_colour_augment": (i32, [P(ColourAugDesc), vp]),
    "sslcr_brightness_contrast": (i32, [P(BrightnessContrastDesc), vp]),
}

_lib = None


def lib():
    """Load libsslcr.so (once).  Raises SslcrError if it was not built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SslcrError(f"{LIB_PATH} not built: run `python -m ssl_cr_histo_amd.build` "
                             "(or __graft_entry__.build()); there is no CPU/PyTorch fallback")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError => header/library mismatch, fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise SslcrError(lib().sslcr_last_error().decode())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())
