#!/usr/bin/env python
"""stem forward / wgrad micro-benchmark (N=640, 256x256 uint8)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_cr_histo_amd import kernels as K
N = int(os.environ.get("CB_N", 640)); reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda:0"
x = torch.randint(0, 256, (N, 3, 256, 256), dtype=torch.uint8, device=dev)
w = torch.randn(64, 3, 7, 7, device=dev) * 0.03
wp, _ = K.pack_stem(w, 1)
def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
y, st = K.stem_conv(x, wp, want_stats=True)
print(f"stem fwd  N={N}: {t(lambda: K.stem_conv(x, wp, want_stats=True)):.1f} us")
dy = torch.randn_like(y); dw = torch.zeros(64, 3, 7, 7, device=dev)
print(f"stem wgrad N={N}: {t(lambda: K.stem_wgrad(x, dy, dw)):.1f} us")
