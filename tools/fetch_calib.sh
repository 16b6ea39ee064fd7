#!/bin/bash
# FETCH_SIZE of tools/microbench/fetch_calib's known-byte-count kernels (run on the GPU box): tools/fetch_calib.sh <tag> -> gpurun_out/<tag>/fetch_calib.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-fetch_calib}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/tools/microbench/fetch_calib > $O/fetch_calib.txt 2>&1
rm -rf /tmp/fc_pmc
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fc_pmc -o fc -- $R/tools/microbench/fetch_calib > /tmp/fc_pmc.log 2>&1
python - "$(find /tmp/fc_pmc -name '*counter_collection.csv' | head -1)" >> $O/fetch_calib.txt <<'PY'
import csv, sys
print("rocprofv3 --pmc FETCH_SIZE (KiB as reported; x2 = the guide's wide-read correction):")
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE":
        v = float(r["Counter_Value"])
        print(f'  {r["Kernel_Name"].split("(")[0]:16s} FETCH_SIZE {v / 1024:9.1f} MiB   x2 = {2 * v / 1024:9.1f} MiB')
PY
cat $O/fetch_calib.txt
