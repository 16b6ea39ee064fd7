cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$1 -o pmc -- python $GRAFT_REPO_ROOT/tools/conv_bench.py bf16 2 "layer2 3x3/1" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, collections
rows=list(csv.DictReader(open("gpurun_out/$1/pmc_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "sslcr" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][12:52]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k, {c.replace("SQ_",""): f"{sum(x)/len(x):.3e}" for c,x in v.items()})
PY
