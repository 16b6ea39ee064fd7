#!/usr/bin/env python
"""time one conv2d shape:  python tools/microbench/conv_shape.py N H C K [reps]   (3x3 stride 1, bf16, plain + stats)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssl_cr_histo_amd import kernels as K
N, H, C, Ko = (int(v) for v in sys.argv[1:5]); reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
td = K.tdtype(1); dev = "cuda:0"
x = torch.randn((N, H, H, C), device=dev).to(td); w = (torch.randn((Ko, 3, 3, C), device=dev) * 0.05).to(td)
def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
fl = 2.0 * N * H * H * Ko * C * 9
for name, fn in (("plain", lambda: K.conv2d(x, w, 1, 1)), ("+stats", lambda: K.conv2d(x, w, 1, 1, want_stats=True))):
    s = t(fn); print(f"N={N} {H}x{H} C{C}->K{Ko} {name:7s} {s*1e6:8.1f} us {fl/s/1e12:7.1f} TF/s  [{K.last_conv_kernel}]")
