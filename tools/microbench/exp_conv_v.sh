#!/bin/bash
# conv_bench.py with the epilogue variants (plain | +prologue | +stats | eval-fused) for each exp/lib_*.so on ONE box
cp ssl_cr_histo_amd/libsslcr.so /tmp/cur.so
for r in 1 2; do
for f in exp/lib_*.so; do
  cp $f ssl_cr_histo_amd/libsslcr.so
  echo "== $f"
  CB_VARIANTS=1 python tools/conv_bench.py bf16 20 "${1:-3x3/1}" 2>/dev/null | sed -e "s/ C[0-9]*->K[0-9]*//;s/| wgrad.*| dgrad/| dgrad/" | cut -c1-24,50-200
done; done
cp /tmp/cur.so ssl_cr_histo_amd/libsslcr.so
