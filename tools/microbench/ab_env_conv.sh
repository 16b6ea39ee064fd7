#!/bin/bash
# conv_bench.py epilogue variants with and without an env-var switch on ONE box: ab_env_conv.sh VAR ["shape filter"]
v=$1
for r in 1 2; do
for s in "" 1; do
  if [ -n "$s" ]; then export $v=1; else unset $v; fi
  echo "== $v=$s"
  CB_VARIANTS=1 timeout 120 python tools/conv_bench.py bf16 20 "${2:-3x3/1}" 2>/dev/null | sed -e "s/ C[0-9]*->K[0-9]*//;s/| wgrad.*| dgrad/| dgrad/" | cut -c1-24,50-200
done; done
