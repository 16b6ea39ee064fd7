#!/usr/bin/env python
"""Pretty-print a bench.py JSON line:  python tools/show_bench.py gpurun_out/bench.json"""
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{d['value']:.1f} {d['unit']}  {d['ms_per_step']:.3f} ms/step  n_gpus={d['n_gpus']}")
for key in ("roofline", "roofline_hbm"):
    r = d.get(key)
    if r: print(f"  {key:13s} {r['kernel'][7:60]:54s} {r['achieved']:8.1f} {r['unit']} frac={r['frac']:.3f} avg={r['avg_launch_us']:.1f} us share={r['time_share_of_step']:.3f}")
for k in d.get("conv_kernels", []):
    print(f"  {k['kernel'][7:70]:64s} n={k['launches_per_step']:3d} avg={k['avg_launch_us']:8.1f} us tf={k['tflops']:7.1f} ms={k['ms_per_step']:.3f}")
print("  conv_all", d.get("conv_all"))
print("  cpu_baseline", d.get("cpu_baseline"))
