#!/bin/bash
# the judged evidence of a round in one call: tools/refresh_profiles.sh <tag>  ->  gpurun_out/<tag>/
#   bench.json                 default bench.py line
#   bench_under_rocprofv3.json the same command under rocprofv3 --kernel-trace --stats
#   kernel_stats.csv           its per-kernel summary (average duration of the roofline kernel must agree with bench.json)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-refresh}
mkdir -p $out
python $R/bench.py 2>/dev/null | grep '^{' | tail -1 > $out/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o t -- python $R/bench.py > /tmp/prof_bench.log 2>&1
grep '^{' /tmp/prof_bench.log | tail -1 > $out/bench_under_rocprofv3.json
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
ls -la $out; head -4 $out/kernel_stats.csv
