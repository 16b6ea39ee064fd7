#!/usr/bin/env python
"""The benchmark step with the inputs handed over as HOST tensors (pinned uint8, the DataLoader's output) instead of resident in HBM:
images/s including the PCIe copy of 1088 x 3 x 256 x 256 bytes per step (DESIGN section 5 quotes this next to bench.py's value,
which by contract starts with the inputs in HBM).   python tools/pcie_inclusive.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ssl_cr_histo_amd import engine as E  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
eng = E.set_engine(E.Engine(dev, "bf16"))
mt, ct = bench.build_nets(dev)
ms, cs = bench.build_nets(dev)
for m in (mt, ct):
    m.eval()
for p in list(mt.parameters()) + list(ct.parameters()):
    p.requires_grad = False
te, st = eng.bind(mt, ct), eng.bind(ms, cs)
opt = torch.optim.Adam(list(ms.parameters()) + list(cs.parameters()), lr=1e-4, weight_decay=1e-4)
b, mu, hw = 64, 7, 256
nx, nu = 3 * b, mu * b
host = [torch.randint(0, 256, (n, 3, hw, hw), dtype=torch.uint8).pin_memory() for n in (nx, nu, nu)]
y = torch.rand(nx).to(dev)
resident = [h.to(dev) for h in host]


def run(src, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        x, u_w, u_s = (t.to(dev, non_blocking=True) for t in src)
        eng.step_ssl_cr(te, st, "mse", x, y, u_w, u_s, 1.0)
        st.optimizer_step(opt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def run_epoch(prefetch, n):
    """the same through the drop-in epoch function (eval_BreastPathQ_SSL_CR.train) fed by host loaders, with / without steps._ahead"""
    import types
    from ssl_cr_histo_amd import steps as S
    lab = [(host[0].reshape(b, 3, 3, hw, hw), torch.rand(b, 3))] * n
    unl = [(host[1], host[2])] * n
    a = types.SimpleNamespace(lambda_u=1.0, print_freq=0, device_prefetch=prefetch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    S.bpq_cr_train(a, mt, ms, ct, cs, lab, unl, opt, 1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


run(resident, 5)
t_res = run(resident, steps)
t_host = run(host, steps)
mb = sum(h.numel() for h in host) / 1e6
print(f"inputs resident in HBM: {t_res * 1e3:.3f} ms/step, {(nx + 2 * nu) / t_res:.0f} images/s")
print(f"inputs from pinned host memory ({mb:.0f} MB/step over PCIe, copies on the compute stream): {t_host * 1e3:.3f} ms/step, "
      f"{(nx + 2 * nu) / t_host:.0f} images/s; copy share {(t_host - t_res) * 1e3:.3f} ms = {mb / 1e3 / max(t_host - t_res, 1e-9):.1f} GB/s")
run_epoch(True, 3)
t_e0, t_e1 = run_epoch(False, steps), run_epoch(True, steps)
print(f"epoch function on host loaders: {t_e0 * 1e3:.3f} ms/step = {(nx + 2 * nu) / t_e0:.0f} images/s ; with args.device_prefetch (next batch copied on a side "
      f"stream under the current step): {t_e1 * 1e3:.3f} ms/step = {(nx + 2 * nu) / t_e1:.0f} images/s")
