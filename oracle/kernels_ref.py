"""Per-kernel CPU references (test oracle): each mirrors ONE HIP kernel's contract in the
engine's own data layout (NHWC activations, KRSC weights) using fp32 torch CPU ops, so the
``-m gpu`` tests can compare kernel by kernel.  The math itself is the reference's
(torch ops called by torchvision resnet18 -- see oracle/model.py)."""
import torch
import torch.nn.functional as F


def nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def nchw(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2).contiguous()


def krsc(w_kcrs):
    return w_kcrs.permute(0, 2, 3, 1).contiguous()


def conv_fwd(x_nhwc, w_krsc, stride, pad, in_scale=None, in_shift=None, in_relu=False,
             bias=None, residual=None, relu=False, out_scale=None):
    """y = epilogue(conv(prologue(x))).  prologue: per-channel x*scale+shift (+relu) applied to
    in-bounds pixels only (zero padding stays zero) == BN-apply fused into the consumer's load."""
    x = x_nhwc.float()
    if in_scale is not None:
        x = x * in_scale + in_shift
        if in_relu:
            x = F.relu(x)
    y = F.conv2d(nchw(x), w_krsc.float().permute(0, 3, 1, 2), None, stride, pad)
    y = nhwc(y)
    if out_scale is not None:            # eval-mode BatchNorm scale kept out of the filters (sslcr_conv_desc.out_scale)
        y = y * out_scale
    if bias is not None:
        y = y + bias
    if residual is not None:
        y = y + residual.float()
    if relu:
        y = F.relu(y)
    return y


def channel_stats(y_nhwc):
    """per-channel (sum, sum of squares) over N*H*W -- what the conv epilogue emits."""
    y = y_nhwc.double().reshape(-1, y_nhwc.shape[-1])
    return y.sum(0), (y * y).sum(0)


def bn_scale_shift(s, ss, count, gamma, beta, eps=1e-5):
    mean = s / count
    var = ss / count - mean * mean            # biased
    inv = 1.0 / torch.sqrt(var + eps)
    scale = gamma.double() * inv
    return scale, beta.double() - mean * scale, mean, var


def conv_dgrad(dy_nhwc, w_krsc, stride, pad, in_hw):
    """dx of conv2d wrt its input (NHWC)."""
    K, R, S, C = w_krsc.shape
    N = dy_nhwc.shape[0]
    dx = torch.nn.grad.conv2d_input((N, C, in_hw[0], in_hw[1]), w_krsc.float().permute(0, 3, 1, 2),
                                    nchw(dy_nhwc.float()), stride, pad)
    return nhwc(dx)


def conv_wgrad(x_nhwc, dy_nhwc, w_shape_krsc, stride, pad):
    K, R, S, C = w_shape_krsc
    dw = torch.nn.grad.conv2d_weight(nchw(x_nhwc.float()), (K, C, R, S), nchw(dy_nhwc.float()), stride, pad)
    return dw.permute(0, 2, 3, 1).contiguous()


def fp8_e4m3(t):
    """round-to-nearest-even to OCP e4m3 after clamping to +-448 (what the fp8 conv kernel does on its load path)."""
    return t.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


def fp8_weight_pack(w_kcrs, bn=None, eps=1e-5):
    """restatement of sslcr_pack_conv_fp8: per-output-channel power-of-two scale with amax * scale in (224, 448], e4m3 rounding.
    -> (quantised weights * scale as fp32 [K,C,3,3], dequant [K], bias [K] | None)"""
    w = w_kcrs.float()
    bias = None
    if bn is not None:
        g, b, rm, rv = bn
        f = g / torch.sqrt(rv + eps)
        w = w * f.view(-1, 1, 1, 1)
        bias = b - rm * f
    amax = w.abs().flatten(1).max(1).values
    e = torch.floor(torch.log2(448.0 / amax.clamp_min(1e-30)))
    scale = torch.where(amax > 0, torch.pow(2.0, e), torch.ones_like(amax))
    return fp8_e4m3(w * scale.view(-1, 1, 1, 1)), 1.0 / scale, bias


def conv3x3_fp8(x_nhwc, wq_kcrs, dequant, x_scale=1.0, in_scale=None, in_shift=None, in_relu=False, bias=None, residual=None, relu=False):
    """the fp8 forward conv on CPU: quantise the (transformed) activations to e4m3, convolve in fp32, dequantise, epilogue."""
    x = x_nhwc.float()
    if in_scale is not None:
        x = x * (in_scale * x_scale) + in_shift * x_scale
    else:
        x = x * x_scale
    if in_relu:
        x = x.clamp_min(0.0)
    xq = fp8_e4m3(x)
    y = torch.nn.functional.conv2d(nchw(xq), wq_kcrs, None, 1, 1) * (dequant / x_scale).view(1, -1, 1, 1)
    raw = y.clone()
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    y = nhwc(y)
    if residual is not None:
        y = y + residual.float()
    if relu:
        y = y.clamp_min(0.0)
    return y, nhwc(raw)
