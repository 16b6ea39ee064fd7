#!/usr/bin/env python
"""Micro-benchmark of the stride-2 family at the layer{2,3,4}.0 shapes of the bench workload (bf16, 256x256 input): the 3x3 / 2 conv,
the 1x1 / 2 projection, the pair in one launch, their dgrads and weight gradients.  HIP events over `reps` back-to-back launches.
python tools/s2_bench.py [reps] [N]      (SSLCR_S2=0 / SSLCR_S2D=0 / SSLCR_S2W=0 keep the gather kernels, for A/B columns)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_cr_histo_amd import kernels as K  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 640
what = sys.argv[3] if len(sys.argv) > 3 else "fwd,dgrad,wgrad"
dev = "cuda:0"
bf = torch.bfloat16
SHAPES = [("layer2.0", 64, 64, 128), ("layer3.0", 32, 128, 256), ("layer4.0", 16, 256, 512)]


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for name, H, C, Ko in SHAPES:
    OH = H // 2
    x = torch.randn((N, H, H, C), device=dev).relu().to(bf)
    w3 = (torch.randn((Ko, 3, 3, C), device=dev) * 0.05).to(bf)
    w1 = (torch.randn((Ko, 1, 1, C), device=dev) * 0.1).to(bf)
    f3 = 2.0 * N * OH * OH * Ko * C * 9
    f1 = f3 / 9
    line = f"{name} N={N} {H}x{H} C{C}->K{Ko}:"
    if "fwd" in what:
        t3 = timeit(lambda: K.conv2d(x, w3, 2, 1, want_stats=True))
        k3 = K.last_conv_kernel.split("::")[-1][:22]
        t1 = timeit(lambda: K.conv2d(x, w1, 2, 0, want_stats=True))
        line += f" 3x3/2 [{k3}] {t3 * 1e6:7.1f} us {f3 / t3 / 1e12:6.1f} TF/s | 1x1/2 {t1 * 1e6:6.1f} us"
        try:
            tp = timeit(lambda: K.conv2d_s2_pair(x, w3, w1, want_stats=True))
            line += f" | pair {tp * 1e6:7.1f} us {(f3 + f1) / tp / 1e12:6.1f} TF/s (two launches {(t3 + t1) * 1e6:.1f})"
            b = torch.randn(Ko, device=dev)
            te = timeit(lambda: K.conv2d_s2_pair(x, w3, w1, bias3=b, bias1=b, relu3=True))
            line += f" | eval pair {te * 1e6:7.1f} us"
        except Exception as e:      # noqa: BLE001
            line += f" | pair: {str(e)[:40]}"
    dy = torch.randn((N, OH, OH, Ko), device=dev).to(bf)
    if "dgrad" in what:
        wd = (torch.randn((C, 3, 3, Ko), device=dev) * 0.05).to(bf)
        td = timeit(lambda: K.conv2d(dy, wd, 2, 1, transposed=True, pixel_hw=(H // 2, H // 2), pix_mul=2, par4=True, out_hw=(H, H)))
        line += f" | dgrad [{K.last_conv_kernel.split('::')[-1][:22]}] {td * 1e6:7.1f} us {f3 / td / 1e12:6.1f} TF/s"
    if "wgrad" in what:
        dw = torch.zeros((Ko, 3, 3, C), device=dev)
        tw = timeit(lambda: K.conv2d_wgrad(x, dy, dw, 3, 3, 2, 1))
        dw1 = torch.zeros((Ko, 1, 1, C), device=dev)
        tw1 = timeit(lambda: K.conv2d_wgrad(x, dy, dw1, 1, 1, 2, 0))
        line += f" | wgrad 3x3 {tw * 1e6:7.1f} us {f3 / tw / 1e12:6.1f} TF/s, 1x1 {tw1 * 1e6:6.1f} us"
    print(line, flush=True)
