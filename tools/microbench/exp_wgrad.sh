#!/bin/bash
# wgrad TF/s of tools/conv_bench.py for each exp/lib_*.so on ONE box: exp_wgrad.sh ["shape filter"]
cp ssl_cr_histo_amd/libsslcr.so /tmp/cur.so
for r in 1 2; do
for f in exp/lib_*.so; do
  cp $f ssl_cr_histo_amd/libsslcr.so
  echo "== $f"
  timeout 120 python tools/conv_bench.py bf16 20 "${1:-3x3/1}" 2>/dev/null | sed -e "s/ N=.*| wgrad/ wgrad/;s/| dgrad.*//"
done; done
cp /tmp/cur.so ssl_cr_histo_amd/libsslcr.so
