#!/bin/bash
# wgrad TF/s of tools/conv_bench.py with and without an env-var switch on ONE box: ab_env_wgrad.sh VAR ["shape filter"]
v=$1
for r in 1 2; do
for s in "" 1; do
  if [ -n "$s" ]; then export $v=1; else unset $v; fi
  echo "== $v=$s"
  timeout 120 python tools/conv_bench.py bf16 20 "${2:-3x3/1}" 2>/dev/null | sed -e "s/.*| wgrad/wgrad/;s/| dgrad.*//" 
done; done
