#!/bin/bash
# HBM traffic of the conv micro-benchmark kernels by PMC (run on the GPU box): tools/pmc_traffic.sh <outdir> ["shape filter"]
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes, counters only (MI355X_MICROARCH.md, HBM section).
cd /tmp && export TMPDIR=/tmp
SHAPE="${2:-layer2 3x3/1}"
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$1/$c -o pmc -- python $GRAFT_REPO_ROOT/tools/conv_bench.py bf16 3 "$SHAPE" > $GRAFT_REPO_ROOT/gpurun_out/$1/$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, collections, glob, json
agg=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for f in glob.glob("gpurun_out/$1/*/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sslcr" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out={}
for k,v in agg.items():
    out[k]={c:{"mean":sum(x)/len(x),"n":len(x),"values":x} for c,x in v.items()}
    print(k, {c:(round(sum(x)/len(x),1),len(x)) for c,x in v.items()})
json.dump(out, open("gpurun_out/$1/traffic.json","w"), indent=1)
PY
