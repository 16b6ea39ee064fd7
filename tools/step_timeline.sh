#!/bin/bash
# one step of the default bench as a kernel timeline (run on the GPU box): tools/step_timeline.sh <out.txt> [bench args...]
# rocprofv3 --kernel-trace over a short serialised run; the launches between the last two optimizer_chunks_kernel are one step
R=$GRAFT_REPO_ROOT; OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/stl
rocprofv3 --kernel-trace --output-format csv -d /tmp/stl -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also --no-roofline --no-pmc --aux-stream 0 --wgrad-stream 0 "$@" > /tmp/stl.log 2>&1
python - "$(find /tmp/stl -name '*kernel_trace.csv' | head -1)" > $R/$OUT <<'PY'
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
opt = [i for i, r in enumerate(rows) if "optimizer_chunks_kernel" in r["Kernel_Name"]]
a, b = opt[-2] + 1, opt[-1] + 1
while a < b and "pack_stem" not in rows[a]["Kernel_Name"] and "stem" not in rows[a]["Kernel_Name"]: a += 1      # (the step begins at its first stem launch)
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
busy = 0.0
for i, r in enumerate(step):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("sslcr::", "").replace("unsigned short", "bf16")
    busy += (e - s) / 1e3
    print(f"{i:4d} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} us  grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d}x{int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y'])):3d} wg {int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']):4d}  {name}")
print(f"# {len(step)} launches, {busy / 1e3:.3f} ms of kernel time, {(int(step[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms first start to last end (under the tracer)")
PY
tail -1 $R/$OUT
