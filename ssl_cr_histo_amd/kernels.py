"""Thin torch-tensor front end of the per-kernel C-ABI entry points (include/sslcr.h).

Tensors are only device memory here: every function fills a descriptor with raw pointers/sizes and calls
libsslcr.so on the current HIP stream.  Activations NHWC, weights KRSC; ``dtype`` 0 = fp32, 1 = bf16.
"""
import torch

from . import _lib as L

F32, BF16 = 0, 1


def tdtype(dtype):
    return torch.bfloat16 if dtype == BF16 else torch.float32


def _dt(t):
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported activation dtype {t.dtype}")


def _chk(*ts):
    for t in ts:
        if t is not None:
            if not t.is_cuda:
                raise L.SslcrError("sslcr kernels need device tensors (no CPU fallback)")
            if not t.is_contiguous():
                raise L.SslcrError("sslcr kernels need contiguous tensors")


last_conv_kernel = ""        # kernel instance the most recent conv2d() launched (rocprofv3 spelling), for tests / profiles
last_wgrad_kernel = ""       # ... and the most recent conv2d_wgrad()


def conv2d(x, w, stride, pad, *, in_scale=None, in_shift=None, in_relu=False, bias=None, residual=None, relu=False,
           want_stats=False, out=None, transposed=False, out_hw=None, osh=1, accumulate=False, pixel_hw=None,
           pix_mul=0, pix_off=(0, 0), tap_mask=0, mask=None, par4=False, seg_images=0, out_scale=None):
    """x NHWC [N,H,W,C], w KRSC [K,R,S,C] -> y NHWC (+ partial stats [rows,2,K] fp32).

    out_scale [K] (with bias): y = epilogue(acc * out_scale + bias): eval-mode BatchNorm with its scale kept out of the filters.
    transposed=True is the dgrad gather: pixel space = the conv's input (pixel_hw), x = dY, w = [C][R][S][K].
    mask = (x_bn [like y], scale [K], shift [K], mean [K]): BatchNorm-backward front end, see sslcr_conv_desc.mask_x.
    seg_images > 0: N / seg_images segments in one launch (in_scale / in_shift [nseg, C]; stats rows split by segment)."""
    _chk(x, w, in_scale, in_shift, bias, residual, out, out_scale)
    dt = _dt(x)
    N, H, W, C = x.shape
    K, R, S, C2 = w.shape
    assert C2 == C and w.dtype == x.dtype
    if transposed:
        PH, PW = pixel_hw
    else:
        PH, PW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    OH, OW = out_hw if out_hw is not None else (PH * osh, PW * osh)
    y = out if out is not None else torch.empty((N, OH, OW, K), dtype=x.dtype, device=x.device)
    d = L.ConvDesc(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(in_scale), L.ptr(in_shift), L.ptr(bias), L.ptr(residual), None,
                   N, H, W, C, K, R, S, stride, pad, PH, PW, OH, OW, osh, int(transposed), int(in_relu), int(relu),
                   int(accumulate), int(pix_mul), int(pix_off[0]), int(pix_off[1]), int(tap_mask))
    d.par4 = int(par4)
    d.out_scale = L.ptr(out_scale)
    if seg_images:
        d.seg_images = int(seg_images)
        d.seg_stride = int(in_scale.stride(0)) if in_scale is not None and in_scale.dim() == 2 else 0
        if not L.lib().sslcr_conv2d_segments_ok(dt, d):
            raise L.SslcrError("conv2d: no segment form for this shape / dtype")
    if mask is not None:
        _chk(*mask)
        d.mask_x, d.mask_scale, d.mask_shift, d.mask_mean = (L.ptr(t) for t in mask)
    stats = None
    if want_stats:
        rows = L.lib().sslcr_conv2d_partial_rows(d)
        # NaN-poisoned: the launch must write every row it announced (a row left as found once went unnoticed here and blew up
        # the engine's BatchNorm, whose scratch is reused)
        stats = torch.full((rows, 2, K), float("nan"), dtype=torch.float32, device=x.device)
        d.stats = L.ptr(stats)
    global last_conv_kernel
    last_conv_kernel = L.lib().sslcr_conv2d_kernel_name(dt, d).decode()
    L.check(L.lib().sslcr_conv2d(dt, d, L.stream_ptr()))
    return (y, stats) if want_stats else y


def conv2d_s2_pair(x, w3, w1, *, bias3=None, bias1=None, relu3=False, want_stats=False, scale3=None, scale1=None):
    """a downsampling BasicBlock's conv1 (3x3 / 2 / pad 1, w3 [K,3,3,C]) and projection (1x1 / 2, w1 [K,1,1,C]) of ONE input in one
    launch (sslcr_conv2d_s2_pair) -> (y3, yd) or (y3, yd, stats3, statsd)"""
    _chk(x, w3, w1, bias3, bias1, scale3, scale1)
    dt = _dt(x)
    N, H, W, C = x.shape
    K = w3.shape[0]
    assert w3.shape == (K, 3, 3, C) and w1.shape == (K, 1, 1, C)
    PH, PW = H // 2, W // 2
    y3 = torch.empty((N, PH, PW, K), dtype=x.dtype, device=x.device)
    yd = torch.empty_like(y3)
    d3 = L.ConvDesc(L.ptr(x), L.ptr(w3), L.ptr(y3), None, None, L.ptr(bias3), None, None,
                    N, H, W, C, K, 3, 3, 2, 1, PH, PW, PH, PW, 1, 0, 0, int(relu3), 0, 0, 0, 0, 0)
    d1 = L.ConvDesc(L.ptr(x), L.ptr(w1), L.ptr(yd), None, None, L.ptr(bias1), None, None,
                    N, H, W, C, K, 1, 1, 2, 0, PH, PW, PH, PW, 1, 0, 0, 0, 0, 0, 0, 0, 0)
    d3.out_scale, d1.out_scale = L.ptr(scale3), L.ptr(scale1)
    s3 = sd = None
    if want_stats:
        rows = L.lib().sslcr_conv2d_partial_rows(d3)
        s3 = torch.full((rows, 2, K), float("nan"), dtype=torch.float32, device=x.device)
        sd = torch.full((rows, 2, K), float("nan"), dtype=torch.float32, device=x.device)
        d3.stats, d1.stats = L.ptr(s3), L.ptr(sd)
    if not L.lib().sslcr_conv2d_s2_pair_ok(dt, d3, d1):
        raise L.SslcrError("conv2d_s2_pair: not served")
    L.check(L.lib().sslcr_conv2d_s2_pair(dt, d3, d1, L.stream_ptr()))
    return (y3, yd, s3, sd) if want_stats else (y3, yd)


def pack_conv_fp8(w_kcrs, *, bn=None, eps=1e-5):
    """PyTorch [K,C,3,3] fp32 -> (w8 [K,3,3,C] uint8 = OCP e4m3 bits, w_dequant [K] fp32, bias [K] | None): per-output-channel
    power-of-two scale (amax -> (224, 448]); bn = (gamma, beta, running_mean, running_var) folds eval-mode BatchNorm."""
    _chk(w_kcrs)
    K, C, R, S = w_kcrs.shape
    assert R == 3 and S == 3
    dev = w_kcrs.device
    w8 = torch.empty((K, 3, 3, C), dtype=torch.uint8, device=dev)
    dq = torch.empty(K, dtype=torch.float32, device=dev)
    bias = torch.empty(K, dtype=torch.float32, device=dev) if bn is not None else None
    g, b, rm, rv = bn if bn is not None else (None, None, None, None)
    d = L.PackFp8Desc(L.ptr(w_kcrs), L.ptr(w8), L.ptr(dq), L.ptr(g), L.ptr(b), L.ptr(rm), L.ptr(rv), eps, L.ptr(bias), K, C)
    L.check(L.lib().sslcr_pack_conv_fp8(d, L.stream_ptr()))
    return w8, dq, bias


def conv2d_fp8(x, w8, w_dequant, *, x_scale=1.0, in_scale=None, in_shift=None, in_relu=False, bias=None, residual=None, relu=False,
               want_stats=False, amax_out=None):
    """fp8 (e4m3) forward conv 3x3 / stride 1 / pad 1: x bf16 NHWC [N,H,W,C], w8 [K,3,3,C] e4m3 bits -> y bf16 NHWC (+ stats)."""
    _chk(x, w8, w_dequant, in_scale, in_shift, bias, residual)
    assert x.dtype == torch.bfloat16 and w8.dtype == torch.uint8
    N, H, W, C = x.shape
    K = w8.shape[0]
    y = torch.empty((N, H, W, K), dtype=torch.bfloat16, device=x.device)
    d = L.ConvDesc(L.ptr(x), None, L.ptr(y), L.ptr(in_scale), L.ptr(in_shift), L.ptr(bias), L.ptr(residual), None,
                   N, H, W, C, K, 3, 3, 1, 1, H, W, H, W, 1, 0, int(in_relu), int(relu), 0, 0, 0, 0, 0)
    q = L.Fp8Desc(L.ptr(w8), L.ptr(w_dequant), float(x_scale), None, L.ptr(amax_out))
    stats = None
    if want_stats:
        rows = L.lib().sslcr_conv2d_fp8_partial_rows(d)
        stats = torch.full((rows, 2, K), float("nan"), dtype=torch.float32, device=x.device)
        d.stats = L.ptr(stats)
    L.check(L.lib().sslcr_conv2d_fp8(d, q, L.stream_ptr()))
    return (y, stats) if want_stats else y


def conv2d_wgrad(x, dy, dw, R, S, stride, pad, *, in_scale=None, in_shift=None, in_relu=False, seg_images=0):
    """accumulate dW [K,R,S,C] fp32 += wgrad(x NHWC, dy NHWC).  seg_images > 0: in_scale / in_shift are [N // seg_images, C],
    one producer BatchNorm per segment of seg_images images."""
    _chk(x, dy, dw, in_scale, in_shift)
    N, H, W, C = x.shape
    _, OH, OW, K = dy.shape
    assert dw.shape == (K, R, S, C) and dw.dtype == torch.float32
    d = L.WgradDesc(L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(in_scale), L.ptr(in_shift), int(in_relu),
                    N, H, W, C, K, R, S, stride, pad, OH, OW, int(seg_images), C if seg_images else 0)
    global last_wgrad_kernel
    last_wgrad_kernel = L.lib().sslcr_conv2d_wgrad_kernel_name(_dt(x), d).decode()
    L.check(L.lib().sslcr_conv2d_wgrad(_dt(x), d, L.stream_ptr()))


def pack_conv(w_kcrs, dtype, *, fwd=True, dgrad=False, bn=None, eps=1e-5, dgrad_flip=False, unfold=False):
    """PyTorch [K,C,R,S] fp32 -> (w_fwd [K,R,S,C], w_dgrad [C,R,S,K], bias[K]|None) in engine dtype.
    bn = (gamma, beta, running_mean, running_var) folds eval-mode BatchNorm into w_fwd/bias; with unfold the filter stays plain and
    the BatchNorm scale is returned as a fourth value (sslcr_pack_desc.scale_out -> sslcr_conv_desc.out_scale)."""
    _chk(w_kcrs)
    K, C, R, S = w_kcrs.shape
    dev = w_kcrs.device
    wf = torch.empty((K, R, S, C), dtype=tdtype(dtype), device=dev) if fwd else None
    wd = torch.empty((C, R, S, K), dtype=tdtype(dtype), device=dev) if dgrad else None
    bias = torch.empty(K, dtype=torch.float32, device=dev) if bn is not None else None
    g, b, rm, rv = bn if bn is not None else (None, None, None, None)
    scale = torch.empty(K, dtype=torch.float32, device=dev) if (unfold and bn is not None) else None
    d = L.PackDesc(L.ptr(w_kcrs), L.ptr(wf), L.ptr(wd), L.ptr(g), L.ptr(b), L.ptr(rm), L.ptr(rv), eps, L.ptr(bias),
                   K, C, R, S, int(dgrad_flip), L.ptr(scale))
    L.check(L.lib().sslcr_pack_conv(dtype, d, L.stream_ptr()))
    return (wf, wd, bias, scale) if unfold else (wf, wd, bias)


def pack_stem(w_kcrs, dtype, *, bn=None, eps=1e-5, unfold=False):
    _chk(w_kcrs)
    dev = w_kcrs.device
    wf = torch.empty((64, 7, 8, 4), dtype=tdtype(dtype), device=dev)
    bias = torch.empty(64, dtype=torch.float32, device=dev) if bn is not None else None
    g, b, rm, rv = bn if bn is not None else (None, None, None, None)
    scale = torch.empty(64, dtype=torch.float32, device=dev) if (unfold and bn is not None) else None
    d = L.PackDesc(L.ptr(w_kcrs), L.ptr(wf), None, L.ptr(g), L.ptr(b), L.ptr(rm), L.ptr(rv), eps, L.ptr(bias), 64, 3, 7, 7, 0, L.ptr(scale))
    L.check(L.lib().sslcr_pack_stem(dtype, d, L.stream_ptr()))
    return (wf, bias, scale) if unfold else (wf, bias)


def stem_conv(x_nchw, w_packed, *, bias=None, relu=False, want_stats=False, x2=None, out_scale=None):
    """x NCHW uint8|fp32 [N,3,H,W] (optionally followed by a second segment x2, read in place of a torch.cat)
    -> NHWC [N(+N2),OH,OW,64] in w_packed's dtype (+ partial stats)."""
    _chk(x_nchw, w_packed, bias, x2)
    N, _, H, W = x_nchw.shape
    n_split = 0
    if x2 is not None:
        assert x2.dtype == x_nchw.dtype and x2.shape[1:] == x_nchw.shape[1:]
        n_split, N = N, N + x2.shape[0]
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    assert x_nchw.dtype in (torch.uint8, torch.float32)
    y = torch.empty((N, OH, OW, 64), dtype=w_packed.dtype, device=x_nchw.device)
    _chk(out_scale)
    d = L.StemDesc(L.ptr(x_nchw), L.ptr(w_packed), L.ptr(y), L.ptr(bias), None, N, H, W, OH, OW,
                   int(x_nchw.dtype == torch.float32), int(relu), L.ptr(x2), int(n_split), L.ptr(out_scale))
    stats = None
    if want_stats:
        rows = L.lib().sslcr_stem_partial_rows(d)
        stats = torch.full((rows, 2, 64), float("nan"), dtype=torch.float32, device=x_nchw.device)
        d.stats = L.ptr(stats)
    L.check(L.lib().sslcr_stem_conv(_dt(w_packed), d, L.stream_ptr()))
    return (y, stats) if want_stats else y


def stem_conv_pool(x_nchw, w_packed, bias, *, x2=None, out_scale=None):
    """eval-mode stem in one launch: conv1 (folded BatchNorm: bias) -> ReLU -> maxpool 3x3/2 pad 1 -> NHWC [N, OH/2, OW/2, 64] (bf16)."""
    _chk(x_nchw, w_packed, bias, x2)
    N, _, H, W = x_nchw.shape
    n_split = 0
    if x2 is not None:
        n_split, N = N, N + x2.shape[0]
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    POH, POW = OH // 2, OW // 2
    y = torch.empty((N, POH, POW, 64), dtype=w_packed.dtype, device=x_nchw.device)
    _chk(out_scale)
    d = L.StemDesc(L.ptr(x_nchw), L.ptr(w_packed), L.ptr(y), L.ptr(bias), None, N, H, W, OH, OW,
                   int(x_nchw.dtype == torch.float32), 1, L.ptr(x2), int(n_split), L.ptr(out_scale))
    L.check(L.lib().sslcr_stem_conv_pool(_dt(w_packed), d, POH, POW, L.stream_ptr()))
    return y


def stem_wgrad(x_nchw, dy, dw, *, x2=None):
    _chk(x_nchw, dy, dw, x2)
    N, _, H, W = x_nchw.shape
    n_split = 0
    if x2 is not None:
        assert x2.dtype == x_nchw.dtype and x2.shape[1:] == x_nchw.shape[1:]
        n_split, N = N, N + x2.shape[0]
    _, OH, OW, _ = dy.shape
    d = L.StemWgradDesc(L.ptr(x_nchw), L.ptr(dy), L.ptr(dw), N, H, W, OH, OW, int(x_nchw.dtype == torch.float32),
                        L.ptr(x2), int(n_split))
    L.check(L.lib().sslcr_stem_wgrad(_dt(dy), d, L.stream_ptr()))


def stem_wgrad_pool(x_nchw, dw, raw, scale, shift, mean, invstd, pool, *, x2=None, count=None, dgamma=None, dbeta=None):
    """conv1 wgrad straight from the pooled gradient: pool = (pooled_dy, argmax[, pooled y]); raw = conv1's output.  Runs the
    pool-form BatchNorm-backward reduce pass, then the fused apply + wgrad kernel.  -> sums[2,64] fp64"""
    _chk(x_nchw, dw, raw, scale, shift, mean, invstd, x2, dgamma, dbeta)
    N, _, H, W = x_nchw.shape
    n_split = 0
    if x2 is not None:
        n_split, N = N, N + x2.shape[0]
    _, OH, OW, Cn = raw.shape
    pixels = raw.numel() // Cn
    sums = torch.zeros((2, Cn), dtype=torch.float64, device=raw.device)
    pdy, pam = pool[0], pool[1]
    _chk(pdy, pam)
    b = L.BnBwdDesc(None, L.ptr(raw), None, L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd), L.ptr(sums),
                    None, None, pixels, Cn, 1, float(count if count is not None else pixels),
                    L.ptr(pdy), L.ptr(pam), OH, OW, pdy.shape[1], pdy.shape[2], None, 0)
    if len(pool) > 2 and pool[2] is not None:
        _chk(pool[2])
        b.pool_y = L.ptr(pool[2])
    if dgamma is not None:
        b.dgamma, b.dbeta, b.pg_scale = L.ptr(dgamma), L.ptr(dbeta), 1.0
    # the reduce pass of the pool form never touches dx; the C-ABI check wants the pointer for the plain form only
    L.check(L.lib().sslcr_bn_bwd_reduce(_dt(raw), b, L.stream_ptr()))
    w = L.StemWgradDesc(L.ptr(x_nchw), None, L.ptr(dw), N, H, W, OH, OW, int(x_nchw.dtype == torch.float32),
                        L.ptr(x2), int(n_split))
    L.check(L.lib().sslcr_stem_wgrad_pool(_dt(raw), w, b, L.stream_ptr()))
    return sums


def bn_finalize(partials, count, gamma, beta, *, running_mean=None, running_var=None, nbt=None, momentum=0.1,
                eps=1e-5, replay=1, nseg=1, one_launch=True):
    """partial rows [rows,2,C] -> (scale, shift, mean, invstd); running stats updated in place `replay` times.
    nseg > 1: the rows are nseg equal segments, the outputs are [nseg, C] (count = elements per channel of one segment)."""
    _chk(partials, gamma, beta, running_mean, running_var, nbt)
    rows, _, Cn = partials.shape
    dev = partials.device
    shape = (nseg, Cn) if nseg > 1 else (Cn,)
    scale, shift, mean, invstd = (torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(4))
    stage = torch.empty((max(nseg, 1), 32, 2, Cn), dtype=torch.float64, device=dev)
    # one_launch: the ticketed form (row reduction + finalize in one launch); False: the two launches -- same bits
    tickets = torch.zeros((Cn + 31) // 32, dtype=torch.int32, device=dev) if one_launch else None
    d = L.BnFinalizeDesc(L.ptr(partials), rows, Cn, float(count), L.ptr(gamma), L.ptr(beta), L.ptr(scale), L.ptr(shift),
                         L.ptr(mean), L.ptr(invstd), L.ptr(running_mean), L.ptr(running_var), L.ptr(nbt), momentum, eps,
                         replay, None, None, L.ptr(stage), nseg if nseg > 1 else 0, Cn if nseg > 1 else 0, L.ptr(tickets))
    L.check(L.lib().sslcr_bn_finalize(d, L.stream_ptr()))
    if tickets is not None:
        assert int(tickets.abs().sum()) == 0, "sslcr_bn_finalize left a ticket word non-zero"
    return scale, shift, mean, invstd


def bn_act(x, scale, shift, *, res=None, rscale=None, rshift=None, relu=True, nseg=1, want_bits=False):
    """nseg > 1: x is nseg equal segments along its first dimension, scale / shift (/ rscale / rshift) are [nseg, C].
    want_bits (bf16): -> (y, bits uint8 [numel / 8]), bit (e & 7) of byte e >> 3 = (y.flatten()[e] > 0)  (sslcr_bn_act_desc.ybits)."""
    _chk(x, scale, shift, res, rscale, rshift)
    y = torch.empty_like(x)
    Cn = x.shape[-1]
    d = L.BnActDesc(L.ptr(x), L.ptr(scale), L.ptr(shift), L.ptr(res), L.ptr(rscale), L.ptr(rshift), L.ptr(y),
                    x.numel() // Cn, Cn, int(relu), nseg if nseg > 1 else 0, Cn if nseg > 1 else 0)
    bits = None
    if want_bits:
        bits = torch.full((x.numel() // 8,), 0xa5, dtype=torch.uint8, device=x.device)
        d.ybits = L.ptr(bits)
    L.check(L.lib().sslcr_bn_act(_dt(x), d, L.stream_ptr()))
    return (y, bits) if want_bits else y


def bn_relu_maxpool(x, scale, shift):
    """maxpool3x3/2 pad 1 of relu(scale * x + shift) -> (y, argmax codes); scale = shift = None: plain max-pool of x -> (y, None)."""
    _chk(x, scale, shift)
    N, H, W, Cn = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((N, OH, OW, Cn), dtype=x.dtype, device=x.device)
    am = torch.empty((N, OH, OW, Cn), dtype=torch.uint8, device=x.device) if scale is not None else None
    d = L.PoolFwdDesc(L.ptr(x), L.ptr(scale), L.ptr(shift), L.ptr(y), L.ptr(am), N, H, W, Cn, OH, OW)
    L.check(L.lib().sslcr_bn_relu_maxpool(_dt(x), d, L.stream_ptr()))
    return y, am


def maxpool_relu_bwd(dy, argmax, x, scale, shift):
    _chk(dy, argmax, x, scale, shift)
    N, H, W, Cn = x.shape
    _, OH, OW, _ = dy.shape
    dx = torch.empty_like(x)
    d = L.PoolBwdDesc(L.ptr(dy), L.ptr(argmax), L.ptr(x), L.ptr(scale), L.ptr(shift), L.ptr(dx), N, H, W, Cn, OH, OW)
    L.check(L.lib().sslcr_maxpool_relu_bwd(_dt(x), d, L.stream_ptr()))
    return dx


def avgpool_fwd(x):
    _chk(x)
    N, H, W, Cn = x.shape
    y = torch.empty((N, Cn), dtype=torch.float32, device=x.device)
    L.check(L.lib().sslcr_avgpool_fwd(_dt(x), L.ptr(x), L.ptr(y), N, H * W, Cn, L.stream_ptr()))
    return y


def avgpool_bwd(dy, shape, dtype):
    _chk(dy)
    N, H, W, Cn = shape
    dx = torch.empty(shape, dtype=tdtype(dtype), device=dy.device)
    L.check(L.lib().sslcr_avgpool_bwd(dtype, L.ptr(dy), L.ptr(dx), N, H * W, Cn, L.stream_ptr()))
    return dx


def bn_bwd(dy, x, scale, shift, mean, invstd, *, yact=None, relu_from_x=False, want_g=False, count=None, pool=None,
           g_in_reduce=False, nseg=1, dgamma=None, dbeta=None, yact_bits=None):
    """-> (dx, sums[2,C] fp64, g|None).  g_in_reduce: the reduce pass writes g and the apply pass reads it (needs yact, want_g).
    nseg > 1: x is nseg equal segments along its first dimension with constants [nseg, C]; sums come back [nseg, 2, C];
    dgamma / dbeta (fp32 [C], accumulated into) take every segment's contribution."""
    _chk(x, scale, shift, mean, invstd, yact, dgamma, dbeta)
    if dy is not None:
        _chk(dy)
    Cn = x.shape[-1]
    pixels = x.numel() // Cn
    sums = torch.zeros((nseg, 2, Cn) if nseg > 1 else (2, Cn), dtype=torch.float64, device=x.device)
    dx = torch.empty_like(x)
    g = torch.empty_like(x) if want_g else None
    d = L.BnBwdDesc(L.ptr(dy), L.ptr(x), L.ptr(yact), L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd), L.ptr(sums),
                    L.ptr(dx), L.ptr(g), pixels, Cn, int(relu_from_x), float(count if count is not None else pixels // nseg),
                    None, None, 0, 0, 0, 0, None, int(g_in_reduce))
    if dgamma is not None:
        d.dgamma, d.dbeta, d.pg_scale = L.ptr(dgamma), L.ptr(dbeta), 1.0
    if yact_bits is not None:     # the sign mask of yact as sslcr_bn_act wrote it: read instead of yact (which stays in the descriptor)
        _chk(yact_bits)
        d.yact_bits = L.ptr(yact_bits)
    if nseg > 1:
        d.nseg, d.seg_stride, d.sums_stride = nseg, Cn, 2 * Cn
    if pool is not None:          # pool = (pooled_dy [N,OH,OW,C], argmax u8[, pooled output y]): dy arrives through the stem max-pool
        pdy, pam = pool[0], pool[1]
        _chk(pdy, pam)
        d.pool_dy, d.pool_argmax = L.ptr(pdy), L.ptr(pam)
        if len(pool) > 2 and pool[2] is not None:
            _chk(pool[2])
            d.pool_y = L.ptr(pool[2])
        d.pH, d.pW, d.pOH, d.pOW = x.shape[1], x.shape[2], pdy.shape[1], pdy.shape[2]
    L.check(L.lib().sslcr_bn_bwd_reduce(_dt(x), d, L.stream_ptr()))
    L.check(L.lib().sslcr_bn_bwd_apply(_dt(x), d, L.stream_ptr()))
    return dx, sums, g


def bn_param_grads(sums, invstd, dgamma, dbeta):
    _chk(sums, invstd, dgamma, dbeta)
    L.check(L.lib().sslcr_bn_param_grads(L.ptr(sums), L.ptr(invstd), L.ptr(dgamma), L.ptr(dbeta), invstd.numel(),
                                         L.stream_ptr()))


def linear_fwd(x, w, b, relu=False):
    _chk(x, w, b)
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    L.check(L.lib().sslcr_linear_fwd(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), M, N, K, int(relu), L.stream_ptr()))
    return y


def linear_bwd(x, w, dy, *, yact=None, dw=None, db=None, want_dx=True, dx=None, dx_accumulate=False):
    _chk(x, w, dy, yact, dw, db, dx)
    M, K = x.shape
    N = w.shape[0]
    if want_dx and dx is None:
        dx = torch.empty((M, K), dtype=torch.float32, device=x.device)
    scratch = torch.empty((M, N), dtype=torch.float32, device=x.device)
    L.check(L.lib().sslcr_linear_bwd(L.ptr(x), L.ptr(w), L.ptr(dy), L.ptr(yact), L.ptr(dx) if want_dx else None,
                                     L.ptr(dw), L.ptr(db), M, N, K, int(dx_accumulate), L.ptr(scratch), L.stream_ptr()))
    return dx


def loss(kind, logits, *, logits_t=None, target_f=None, target_i=None, nx, lambda_u=1.0, want_grad=True,
         nx_global=None, nu_global=None):
    _chk(logits, logits_t, target_f, target_i)
    Ns, Cn = logits.shape
    nu = Ns - nx
    dl = torch.zeros_like(logits) if want_grad else None
    out = torch.zeros(4, dtype=torch.float32, device=logits.device)
    d = L.LossDesc(kind, L.ptr(logits), L.ptr(logits_t), L.ptr(target_f), L.ptr(target_i), L.ptr(dl), L.ptr(out), nx, nu,
                   Cn, lambda_u, 1.0 / (nx_global or nx), 1.0 / max(1, (nu_global or nu)))
    L.check(L.lib().sslcr_loss(d, L.stream_ptr()))
    return out, dl


def softmax_col(logits, col=-1):
    """softmax(logits, dim=1)[:, col] for fp32 logits [N, C] (test_Camelyon16.py:58-60)."""
    _chk(logits)
    n, c = logits.shape
    out = torch.empty(n, dtype=torch.float32, device=logits.device)
    L.check(L.lib().sslcr_softmax_col(L.ptr(logits), L.ptr(out), n, c, col % c, L.stream_ptr()))
    return out
