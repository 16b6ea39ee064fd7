#!/bin/bash
# time conv_bench with the current library and each experimental library under exp/ (ablation builds: results are garbage,
# timings are the point)
cp ssl_cr_histo_amd/libsslcr.so /tmp/base.so
for f in /tmp/base.so exp/lib_*.so; do
  cp $f ssl_cr_histo_amd/libsslcr.so
  echo "== $f"
  CB_VARIANTS=1 timeout 60 python tools/conv_bench.py bf16 20 "${1:-3x3/1}" 2>/dev/null | sed -e "s/ C[0-9]*->K[0-9]*//;s/| wgrad.*| dgrad/| dgrad/" | cut -c1-24,50-200
done
cp /tmp/base.so ssl_cr_histo_amd/libsslcr.so
