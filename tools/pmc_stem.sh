#!/bin/bash
# PMC counter passes over the stem micro-benchmark (run on the GPU box):  tools/pmc_stem.sh <outdir-name>
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$1
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC"
      "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM")
i=0
for s in "${SETS[@]}"; do
  rocprofv3 --pmc $s --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$1/set$i -o pmc -- python $GRAFT_REPO_ROOT/tools/stem_bench.py 2 > $GRAFT_REPO_ROOT/gpurun_out/$1/set$i.log 2>&1
  i=$((i+1))
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, collections, glob
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/$1/set*/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "stem" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][12:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,x in sorted(v.items()):
        print(f"   {c:28s} {sum(x)/len(x):.4e}  (n={len(x)})")
PY
