#!/bin/bash
# A/B the whole-step bench between prebuilt libraries exp/lib_*.so on ONE box (box-to-box variance is ~2%)
cp ssl_cr_histo_amd/libsslcr.so /tmp/cur.so
for r in 1 2; do
for f in exp/lib_*.so; do
  cp $f ssl_cr_histo_amd/libsslcr.so
  python bench.py 2>/dev/null | tail -1 > /tmp/b.json
  python - "$f" <<PY
import json, sys
d = json.load(open("/tmp/b.json"))
print(f"{sys.argv[1]:28s} {d['value']:9.1f} img/s {d['ms_per_step']:7.3f} ms  hbm {d['roofline_hbm']['achieved']:7.1f} GB/s ({d['roofline_hbm']['avg_launch_us']:.1f} us)  mfma {d['roofline']['achieved']:7.1f} TF/s")
PY
done; done
cp /tmp/cur.so ssl_cr_histo_amd/libsslcr.so
