#!/bin/bash
# like exp_conv.sh but with the CB_VARIANTS columns (plain | +prologue | +stats | eval-fused)
cp ssl_cr_histo_amd/libsslcr.so /tmp/cur.so
for f in exp/lib_*.so; do
  cp $f ssl_cr_histo_amd/libsslcr.so
  echo "== $f"
  CB_VARIANTS=1 python tools/conv_bench.py bf16 20 "${1:-3x3/1}" 2>/dev/null | sed -e "s/ C[0-9]*->K[0-9]*//" | cut -c1-30,85-160
done
cp /tmp/cur.so ssl_cr_histo_amd/libsslcr.so
