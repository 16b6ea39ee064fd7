"""duration of one 3x3/1 conv launch against the number of images: the intercept is what a launch costs whatever its size.
   python tools/conv_intercept.py [C] [HW]      (bf16, K = C, train-mode form with statistics)"""
import sys
import torch
sys.path.insert(0, ".")
from ssl_cr_histo_amd import kernels as K

C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 32
w = (torch.randn(C, 3, 3, C, device="cuda") * 0.05).bfloat16()
pts = []
for n in (16, 32, 64, 128, 192, 256, 384, 640, 1280):
    x = torch.randn(n, HW, HW, C, device="cuda").bfloat16()
    for _ in range(3):
        K.conv2d(x, w, 1, 1, want_stats=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        K.conv2d(x, w, 1, 1, want_stats=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    pts.append((n, us))
    print(f"N={n:5d}  {us:8.1f} us  {K.last_conv_kernel}")
(n0, t0), (n1, t1) = pts[-3], pts[-1]
b = (t1 - t0) / (n1 - n0)
print(f"slope {b * 1e3:.1f} ns per image, intercept {t1 - b * n1:.1f} us (from N={n0} and N={n1})")
