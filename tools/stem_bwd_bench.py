#!/usr/bin/env python
"""stem backward micro-benchmark at the headline shape (N=1088, 256x256 uint8): bn_bwd pool-form apply + stem_wgrad vs the fused
sslcr_stem_wgrad_pool.  Both include the pool-form reduce pass."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_cr_histo_amd import kernels as K
N = int(os.environ.get("CB_N", 1088)); reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
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
raw, st = K.stem_conv(x, wp, want_stats=True)
g, b = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)
sc, sh, mean, invstd = K.bn_finalize(st, raw.numel() // 64, g, b)
pooled, am = K.bn_relu_maxpool(raw, sc, sh)
dyp = torch.randn_like(pooled)
dw = torch.zeros(64, 3, 7, 7, device=dev)
def unfused():
    dx, _, _ = K.bn_bwd(None, raw, sc, sh, mean, invstd, relu_from_x=True, pool=(dyp, am, pooled))
    K.stem_wgrad(x, dx, dw)
print(f"reduce + apply + wgrad N={N}: {t(unfused):.1f} us")
print(f"reduce + fused wgrad   N={N}: {t(lambda: K.stem_wgrad_pool(x, dw, raw, sc, sh, mean, invstd, (dyp, am, pooled))):.1f} us")
