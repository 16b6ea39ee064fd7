#!/bin/bash
# A/B of the elementwise kernels' grid cap (SSLCR_EW_CAP): tools/ew_cap_sweep.sh
for c in ${CAPS:-4096 2048 1024 512 4096 2048}; do
  SSLCR_EW_CAP=$c timeout 200 python bench.py --no-cpu-baseline --no-also 2>/dev/null > /tmp/b.json
  python - "$c" <<'PY'
import json, sys
d = json.load(open("/tmp/b.json")); h = d["roofline_hbm"]
print("cap", sys.argv[1], d["value"], d["ms_per_step"], h["kernel"][:34], h["achieved"], h["avg_launch_us"])
PY
done
