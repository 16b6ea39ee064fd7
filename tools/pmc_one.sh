#!/bin/bash
# counters of one command, per kernel (mean over dispatches): tools/pmc_one.sh "<counters>" <kernel regex> -- <command...>
R=$GRAFT_REPO_ROOT; C="$1"; K="$2"; shift 3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_one
(cd $R && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_one -o p -- "$@" > /tmp/pmc_one.log 2>&1) || tail -5 /tmp/pmc_one.log
python - "$(find /tmp/pmc_one -name '*counter_collection.csv' | head -1)" "$K" <<'PY'
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Kernel_Name"]):
        agg[r["Kernel_Name"].split("(")[0][-70:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k, {n: round(sum(v) / len(v), 1) for n, v in c.items()}, "dispatches", max(len(v) for v in c.values()))
PY
