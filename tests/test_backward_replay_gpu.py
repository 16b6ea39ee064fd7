"""Layer-wise backward replay at FULL size: every weight-gradient and data-gradient kernel launch of one SSL_CR step at the
benchmark's shape (student 192 + 448 = 640 images of 256x256) is checked on the tensors the engine itself fed it.

The engine-level gradient tests compare final gradients with the reference's, where bf16 storage of the *forward* alone moves the
early layers by tens of percent -- a bound that cannot see a 10 % error in one backward kernel.  Here the debug tap
(include/sslcr.h: sslcr_net_debug_tap / sslcr_net_debug_tensor) hands back, per BasicBlock, the saved activations X and the
transient gradients dY exactly as the kernels read them; the oracle (oracle/kernels_ref.py: torch-CPU fp32 autograd formulas of
nn.Conv2d, the ops torchvision resnet18 runs under models/net.py:32,77) recomputes dW and dX from those same tensors, so what is
left is the kernel's own arithmetic: 1.2e-2 of the tensor's max in bf16 (output rounding), 2e-4 in fp32 -- the per-kernel bounds of
tests/test_kernels_gpu.py, now at 640 x 64^2 ... 640 x 8^2 instead of <= 72 images of <= 32 x 32."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cases as C  # noqa: E402
from oracle import kernels_ref as R  # noqa: E402

from test_engine_gpu import _engine, build, freeze  # noqa: E402
from test_kernels_gpu import close  # noqa: E402

TOL = {"fp32": 2e-4, "bf16": 1.2e-2}
BLOCKS = [f"layer{l}.{b}" for l in (1, 2, 3, 4) for b in (0, 1)]


def _q(t, dtype):
    return t.to(torch.bfloat16).float() if dtype == "bf16" else t.float()


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_every_backward_conv_kernel_on_the_engines_own_tensors(dtype):
    eng = _engine(dtype)
    name = "bpq_cr_full"
    c = C.CASES[name]
    mt, ct = build("finetune", "finetune", 1, True)
    ms, cs = build("finetune", "finetune", 1, True)
    freeze(mt, 64)
    freeze(ms, 0)
    (xl, yl), = C.labeled_batches(name)
    (uw, us), = C.unlabeled_batches(name)
    te, st = eng.bind(mt, ct), eng.bind(ms, cs)
    mt.eval()
    ms.train()
    hw = c["hw"]
    st.debug_tap(True)
    try:
        eng.step_ssl_cr(te, st, "mse", xl.reshape(-1, 3, hw, hw), yl.reshape(-1), uw, us, c["lambda_u"])
        torch.cuda.synchronize()
        pnames = [k for k, _ in ms.named_parameters()]
        params = dict(ms.named_parameters())
        threads = torch.get_num_threads()
        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))     # torch-CPU convolutions get slower beyond ~32 threads
        report = []
        try:
            for blk in range(7, -1, -1):
                pre = f"model.{BLOCKS[blk]}"
                has_ds = f"{pre}.downsample.0.weight" in params
                stride = 2 if has_ds else 1

                def T(kind):
                    t, fl = st.debug_tensor(blk, kind)
                    return t.float().cpu(), fl

                def grad_krsc(pname):
                    return st.grad(pnames.index(pname)).cpu().permute(0, 2, 3, 1).contiguous()

                def w_krsc(pname):
                    return _q(params[pname].detach().cpu(), dtype).permute(0, 2, 3, 1).contiguous()
                x_in, _ = T(10)
                raw1, _ = T(6)
                sc1, sh1 = T(11)[0], T(12)[0]
                G, _ = T(0)
                dRaw2, _ = T(1)
                dAct1, flags = T(2)
                dRaw1, _ = T(3)
                N, oh, ow, K = dRaw2.shape
                xh, xw = x_in.shape[1], x_in.shape[2]
                # ---- conv2: wgrad on relu(bn1(raw1)) computed on the fly, dgrad (+ bn1's ReLU mask where the kernel fuses it)
                pre1 = raw1.double() * sc1.double() + sh1.double()
                x2 = _q(torch.clamp_min(raw1 * sc1 + sh1, 0.0), dtype)
                close(grad_krsc(f"{pre}.conv2.weight"), R.conv_wgrad(x2, dRaw2, (K, 3, 3, K), 1, 1), TOL[dtype], f"{pre}.conv2 wgrad")
                want = R.conv_dgrad(dRaw2, w_krsc(f"{pre}.conv2.weight"), 1, 1, (oh, ow))
                if flags & 1:
                    # the mask is a sign decision on an fp32 fma: leave out the elements whose pre-activation is zero to rounding
                    tie = pre1.abs() <= 1e-6 * (raw1.double() * sc1.double()).abs().clamp_min(1e-30)
                    want = torch.where(pre1 > 0, want, torch.zeros_like(want))
                    want = torch.where(tie, dAct1, want)
                close(dAct1, want, TOL[dtype], f"{pre}.conv2 dgrad (mask fused: {flags & 1})")
                del pre1, x2, want, dAct1, raw1
                # ---- conv1 (3x3, stride 1 or 2) and the 1x1/2 projection: wgrads, and the block-input gradient they sum to
                C1 = x_in.shape[3]
                close(grad_krsc(f"{pre}.conv1.weight"), R.conv_wgrad(x_in, dRaw1, (K, 3, 3, C1), stride, 1), TOL[dtype], f"{pre}.conv1 wgrad")
                want = R.conv_dgrad(dRaw1, w_krsc(f"{pre}.conv1.weight"), stride, 1, (xh, xw))
                if has_ds:
                    dRawD, _ = T(4)
                    close(grad_krsc(f"{pre}.downsample.0.weight"), R.conv_wgrad(x_in, dRawD, (K, 1, 1, C1), 2, 0), TOL[dtype],
                          f"{pre}.downsample.0 wgrad")
                    want = want + R.conv_dgrad(dRawD, w_krsc(f"{pre}.downsample.0.weight"), 2, 0, (xh, xw))
                else:
                    want = want + G
                dXin, _ = T(5)
                close(dXin, want, TOL[dtype], f"{pre} block-input gradient (conv1 dgrad + shortcut)")
                report.append(f"{pre}: N={N} {xh}x{xw}x{C1} -> {oh}x{ow}x{K} ok")
        finally:
            torch.set_num_threads(threads)
        print(f"[{dtype}] backward replay:\n   " + "\n   ".join(report))
    finally:
        st.debug_tap(False)
