#!/bin/bash
# same-box per-kernel A/B of two library builds: tools/ab_kernel.sh <old.so> <new.so> <kernel-name regex> -- rocprofv3 kernel stats of the default bench step under each
R=$GRAFT_REPO_ROOT; L=$R/ssl_cr_histo_amd
cd /tmp && export TMPDIR=/tmp
for v in $1 $2; do
  cp $L/$v $L/libsslcr.so
  rm -rf /tmp/abk_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk_$v -o run -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-roofline --no-pmc > /tmp/abk_$v.log 2>&1
  echo "== $v"
  f=$(find /tmp/abk_$v -name '*kernel_stats.csv' | head -1)
  python - "$f" "$3" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
for r in rows:
    if pat.search(r["Name"]):
        print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.2f}')
PY
done
cp $L/$2 $L/libsslcr.so
