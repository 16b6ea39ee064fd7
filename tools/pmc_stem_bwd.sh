#!/bin/bash
# issue / stall counters of the stem-backward micro-benchmark (GPU box): tools/pmc_stem_bwd.sh <tag>
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-pmc_stem_bwd}; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for s in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  (cd $R && CB_N=640 timeout 200 rocprofv3 --pmc $s --kernel-trace --output-format csv -d $out/set$i -o pmc -- python tools/stem_bwd_bench.py 2 > $out/set$i.log 2>&1)
  i=$((i+1))
done
cd $R
python - "$out" <<'PY'
import csv, collections, glob, sys
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/set*/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "stem_wgrad" in r["Kernel_Name"] or "apply_pool" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,x in sorted(v.items()):
        print(f"   {c:28s} {sum(x)/len(x):.4e}  (n={len(x)})")
PY
