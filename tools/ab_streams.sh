#!/bin/bash
# same-box A/B of the engine's side streams under the default step: teacher forward on the aux stream, weight gradients on theirs
for i in 1 2 3; do
for cfg in "0 0" "1 0" "0 1" "1 1"; do set -- $cfg
 python bench.py --steps 20 --warmup 5 --no-also --no-cpu-baseline --no-pmc --no-roofline --aux-stream $1 --wgrad-stream $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('aux=$1 wgrad=$2', d['ms_per_step'], d['ms_per_step_median_events'])"
done; done
