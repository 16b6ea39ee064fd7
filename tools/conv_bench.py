#!/usr/bin/env python
"""Micro-benchmark of the conv / wgrad kernels at the ResNet18 layer shapes of the bench workload (N=640, 256x256 input).
Prints TFLOP/s per shape from HIP events; run under rocprofv3 --pmc for counters.   python tools/conv_bench.py [bf16|fp32] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_cr_histo_amd import kernels as K  # noqa: E402

dt = 1 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
only = sys.argv[3] if len(sys.argv) > 3 else ""
N = int(os.environ.get("CB_N", 640))
dev = "cuda:0"
td = K.tdtype(dt)
SHAPES = [("layer1 3x3/1", 64, 64, 64, 3, 1), ("layer2 3x3/1", 32, 128, 128, 3, 1), ("layer3 3x3/1", 16, 256, 256, 3, 1),
          ("layer4 3x3/1", 8, 512, 512, 3, 1), ("layer2.0 3x3/2", 64, 64, 128, 3, 2), ("layer3.0 3x3/2", 32, 128, 256, 3, 2),
          ("layer4.0 3x3/2", 16, 256, 512, 3, 2), ("layer2.0 1x1/2", 64, 64, 128, 1, 2)]


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


for name, H, C, Ko, R, s in SHAPES:
    if only and only not in name:
        continue
    pad = R // 2
    OH = (H + 2 * pad - R) // s + 1
    x = torch.randn((N, H, H, C), device=dev).to(td)
    w = (torch.randn((Ko, R, R, C), device=dev) * 0.05).to(td)
    sc = torch.rand(C, device=dev) + 0.5
    sh = torch.randn(C, device=dev)
    flops = 2.0 * N * OH * OH * Ko * C * R * R
    t = timeit(lambda: K.conv2d(x, w, s, pad, in_scale=sc, in_shift=sh, in_relu=True, want_stats=True))
    line = f"{name:16s} N={N} {H}x{H} C{C}->K{Ko}: fwd(train) {t * 1e6:8.1f} us {flops / t / 1e12:7.1f} TF/s"
    if os.environ.get("CB_VARIANTS"):
        t1 = timeit(lambda: K.conv2d(x, w, s, pad))
        t2 = timeit(lambda: K.conv2d(x, w, s, pad, in_scale=sc, in_shift=sh, in_relu=True))
        t3 = timeit(lambda: K.conv2d(x, w, s, pad, want_stats=True))
        bias = torch.randn(Ko, device=dev)
        res = torch.randn((N, OH, OH, Ko), device=dev).to(td)
        t4 = timeit(lambda: K.conv2d(x, w, s, pad, bias=bias, residual=res, relu=True))
        line += f" [plain {t1 * 1e6:.0f} | +prologue {t2 * 1e6:.0f} | +stats {t3 * 1e6:.0f} | eval-fused {t4 * 1e6:.0f} us]"
    dy = torch.randn((N, OH, OH, Ko), device=dev).to(td)
    dw = torch.zeros((Ko, R, R, C), device=dev)
    t = timeit(lambda: K.conv2d_wgrad(x, dy, dw, R, R, s, pad, in_scale=sc, in_shift=sh, in_relu=True))
    line += f" | wgrad {t * 1e6:8.1f} us {flops / t / 1e12:7.1f} TF/s"
    if R == 3:
        wd = (torch.randn((C, R, R, Ko), device=dev) * 0.05).to(td)
        if s == 1:
            t = timeit(lambda: K.conv2d(dy, wd, 1, 1))
        else:
            t = timeit(lambda: K.conv2d(dy, wd, s, pad, transposed=True, pixel_hw=(H, H)))
        line += f" | dgrad {t * 1e6:8.1f} us {flops / t / 1e12:7.1f} TF/s"
    print(line, flush=True)
