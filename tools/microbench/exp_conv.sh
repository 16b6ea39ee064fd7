#!/bin/bash
# time tools/conv_bench.py with each library exp/lib_*.so on ONE box
cp ssl_cr_histo_amd/libsslcr.so /tmp/cur.so
for f in exp/lib_*.so; do
  cp $f ssl_cr_histo_amd/libsslcr.so
  echo "== $f"
  python tools/conv_bench.py bf16 20 "${1:-3x3/1}" 2>/dev/null | sed -e "s/ C[0-9]*->K[0-9]*//" | cut -c1-30,68-140
done
cp /tmp/cur.so ssl_cr_histo_amd/libsslcr.so
