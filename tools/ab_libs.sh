#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_libs.sh <old.so> <new.so> [rounds] -- alternates them under the default bench step
R=$GRAFT_REPO_ROOT; L=$R/ssl_cr_histo_amd
for i in $(seq ${3:-2}); do
  for v in $1 $2; do
    cp $L/$v $L/libsslcr.so
    timeout 200 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-roofline $AB_ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
  done
done
cp $L/$2 $L/libsslcr.so
