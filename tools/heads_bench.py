"""Times the head GEMMs of the default step (fc.0 640x512x1024, fc.2 640x256x512, classifier 640x1x768; forward and backward) through
the C-ABI, in a HIP graph so that the host's launch cost does not show: python tools/heads_bench.py  (SSLCR_GEMM_R=1: the one-tile form)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_cr_histo_amd import kernels as K
from ssl_cr_histo_amd import _lib as L

dev = torch.device("cuda", 0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 640
g = torch.Generator().manual_seed(1)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
shapes = [("fc.0", 1024, 512), ("fc.2", 512, 256), ("cls", 768, 1)]
st = torch.cuda.Stream()
for name, Kd, N in shapes:
    x, w, b, dy = r(M, Kd), r(N, Kd), r(N), r(M, N)
    y = torch.empty(M, N, device=dev); dx = torch.empty(M, Kd, device=dev); dw = torch.zeros(N, Kd, device=dev); db = torch.zeros(N, device=dev)
    scratch = torch.empty(M, N, device=dev)
    def fwd():
        L.check(L.lib().sslcr_linear_fwd(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), M, N, Kd, 1, L.stream_ptr()))
    def bwd():
        L.check(L.lib().sslcr_linear_bwd(L.ptr(x), L.ptr(w), L.ptr(dy), L.ptr(y), L.ptr(dx), L.ptr(dw), L.ptr(db), M, N, Kd, 0, L.ptr(scratch), L.stream_ptr()))
    for what, fn in (("fwd", fwd), ("bwd", bwd)):
        with torch.cuda.stream(st):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 200
            e0.record(st)
            for _ in range(n):
                fn()
            e1.record(st)
            torch.cuda.synchronize()
        print(f"{name:5s} {what} M={M} K={Kd} N={N}: {e0.elapsed_time(e1) / n * 1e3:7.2f} us per call")
    ref = torch.relu(x @ w.t() + b)
    print("      fwd max err", float((y - ref).abs().max() / ref.abs().max()))
