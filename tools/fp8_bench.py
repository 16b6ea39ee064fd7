"""micro-benchmark of the fp8 forward conv kernel: python tools/fp8_bench.py [N H C K reps]  (layer2 shape by default)"""
import sys
import time

import torch

sys.path.insert(0, ".")
from ssl_cr_histo_amd import kernels as K

N, H, C, Ko, reps = (int(v) for v in (sys.argv[1:6] + ["448", "32", "128", "128", "20"][len(sys.argv) - 1:]))
dev = "cuda:0"
x = (torch.randn(N, H, H, C, device=dev) * 1.5).to(torch.bfloat16)
w = torch.randn(Ko, C, 3, 3, device=dev) * 0.04
w8, dq, _ = K.pack_conv_fp8(w)
sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
wb = K.pack_conv(w, 1)[0]
for name, fn in (("fp8 eval-form", lambda: K.conv2d_fp8(x, w8, dq, relu=True)),
                 ("fp8 train-form", lambda: K.conv2d_fp8(x, w8, dq, in_scale=sc, in_shift=sh, in_relu=True, want_stats=True)),
                 ("bf16 train-form", lambda: K.conv2d(x, wb, 1, 1, in_scale=sc, in_shift=sh, in_relu=True, want_stats=True))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:16s} N={N} {H}x{H} C={C} K={Ko}: {dt * 1e6:8.1f} us  {2 * N * H * H * C * Ko * 9 / dt / 1e12:7.1f} TF/s")
