#!/bin/bash
# per-launch kernel timeline of ONE training step (the last one of a short bench run): tools/step_trace.sh <outdir> [bench args]
# writes gpurun_out/<outdir>/step.txt: launch order, kernel, grid, duration -- to see which LAYER a slow launch belongs to
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 "$@" > $out/bench.log 2>&1
cd $GRAFT_REPO_ROOT
python - "$out" <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
f = glob.glob(out + "/trace/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last optimizer launch ends a step; take the launches between the last two
idx = [i for i, r in enumerate(rows) if "optimizer_" in r["Kernel_Name"]]
lo, hi = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
with open(out + "/step.txt", "w") as w:
    for i, r in enumerate(rows[lo:hi]):
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("sslcr::", "").replace("unsigned short", "bf16")
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        w.write(f"{i:4d} {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:8.1f} us  grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d}x{r['Grid_Size_Y']:>4s} wg {r['Workgroup_Size_X']:>4s}  {n[:80]}\n")
print(open(out + "/step.txt").read()[-3000:])
PY
