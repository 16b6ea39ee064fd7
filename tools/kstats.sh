#!/bin/bash
# per-kernel rocprofv3 stats of one bench command (run on the GPU box): tools/kstats.sh <steps> [bench args...]  -> top kernels, launches per step, busy time per step
R=$GRAFT_REPO_ROOT; S=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- python $R/bench.py --steps $S --warmup 3 --no-cpu-baseline --no-also --no-roofline --no-pmc "$@" > /tmp/kst.log 2>&1
grep "^{" /tmp/kst.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"bench under rocprofv3:\", d[\"value\"], \"img/s\", d[\"ms_per_step\"], \"ms/step\")"
python - "$(find /tmp/kst -name '*kernel_stats.csv' | head -1)" $S <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) + 3
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print(f"kernel time per step {tot / steps / 1e6:.3f} ms, launches per step {calls / steps:.0f}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{float(r["TotalDurationNs"]) / tot * 100:5.1f}% {float(r["AverageNs"]) / 1e3:8.1f} us x {int(r["Calls"]) / steps:6.1f}/step  {r["Name"][:100]}')
PY
