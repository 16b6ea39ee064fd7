#!/bin/bash
# same-box A/B of an environment switch under the default step:  tools/ab_env.sh VAR [rounds] [extra bench args...]
# alternates VAR=0 / VAR=1 runs of `bench.py --no-also --no-cpu-baseline --no-pmc --no-roofline` and prints ms/step of each
V=$1; R=${2:-3}; shift; shift
for i in $(seq 1 $R); do
  for x in 0 1; do
    env $V=$x python bench.py --steps 20 --warmup 5 --no-also --no-cpu-baseline --no-pmc --no-roofline "$@" 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$x', d['ms_per_step'], d['value'])"
  done
done
