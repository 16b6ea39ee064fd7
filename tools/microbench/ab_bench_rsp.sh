#!/bin/bash
# A/B the RSP-pretraining bench (3 x N=128 passes per step) between prebuilt libraries exp/lib_*.so on ONE box
cp ssl_cr_histo_amd/libsslcr.so /tmp/cur.so
for f in exp/lib_*.so; do
  cp $f ssl_cr_histo_amd/libsslcr.so
  python bench.py --workload rsp --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python - "$f" <<PY
import json, sys
d = json.load(open("/tmp/b.json"))
print(f"{sys.argv[1]:28s} {d['value']:9.1f} img/s {d['ms_per_step']:7.3f} ms  mfma {d['roofline']['kernel'][7:40]} {d['roofline']['achieved']:7.1f} TF/s")
PY
done
cp /tmp/cur.so ssl_cr_histo_amd/libsslcr.so
