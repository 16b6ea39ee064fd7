"""reduce the rocprofv3 --pmc passes of tools/pmc_step.sh to one JSON: per kernel template instance, averaged over its launches.

  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)   matrix-pipe busy cycles (summed over SIMDs) over
                the cycles the dispatch occupies the GPU -- the MfmaUtil definition.  rocprofv3 reports GRBM_GUI_ACTIVE summed
                over the 8 XCDs (checked against the same dispatch's timestamps: 2 431 424 cycles in 139.6 us = 17.4 GHz =
                8 x 2.18 GHz), hence the / 8; SQ_VALU_MFMA_BUSY_CYCLES = 16 cycles per v_mfma_f32_16x16x32_bf16, i.e. the time the
                instruction needs at the 2.5 PF peak, so mfma_busy is achieved / peak at the clock the kernel actually ran at
  mfma_busy_wave_life = the round-1 definition (busy cycles per SIMD over 4 x SQ_WAVE_CYCLES / SQ_WAVES), kept for comparison
  traffic_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024   FETCH_SIZE doubled: the gfx950 wide-read correction of
                /opt/skills/guides/MI355X_MICROARCH.md (HBM section); WRITE_SIZE as reported (KiB)
"""
import collections
import csv
import glob
import json
import re
import sys


def mean(x):
    return sum(x) / len(x) if x else None


def reduce_dir(out, command=None):
    """-> the JSON document for the rocprofv3 --pmc passes found under `out` (pass*/**/pmc_counter_collection.csv)."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/pass*/**/pmc_counter_collection.csv", recursive=True):
        per_dispatch = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if "sslcr" not in r["Kernel_Name"]:
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            per_dispatch[(r["Dispatch_Id"], name)][r["Counter_Name"]] = float(r["Counter_Value"])
        for (_, name), c in per_dispatch.items():
            for k, v in c.items():
                agg[name][k].append(v)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE", 0) > 0:
                agg[name]["_busy"].append(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0))
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("SQ_WAVES", 0) > 0 and c.get("SQ_WAVE_CYCLES", 0) > 0:
                agg[name]["_busy_wl"].append((c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (4.0 * c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"]))
    kernels = []
    for name, c in agg.items():
        n_disp = max(len(v) for v in c.values())
        e = {"kernel": name, "dispatches_measured": n_disp}
        if c.get("_busy"):
            e["mfma_busy"] = round(mean(c["_busy"]), 4)
        if c.get("_busy_wl"):
            e["mfma_busy_wave_life"] = round(mean(c["_busy_wl"]), 4)
        if c.get("SQ_INSTS_MFMA") and mean(c["SQ_INSTS_MFMA"]) > 0:
            e["valu_per_mfma"] = round(mean(c["SQ_INSTS_VALU"]) / mean(c["SQ_INSTS_MFMA"]), 3)
            e["mfma_insts_per_launch"] = round(mean(c["SQ_INSTS_MFMA"]))
        if c.get("GRBM_GUI_ACTIVE"):
            e["gui_active_cycles"] = round(mean(c["GRBM_GUI_ACTIVE"]) / 8.0)
        if c.get("FETCH_SIZE") and c.get("WRITE_SIZE"):
            e["fetch_kib"] = round(mean(c["FETCH_SIZE"]), 1)
            e["write_kib"] = round(mean(c["WRITE_SIZE"]), 1)
            e["traffic_bytes_per_launch"] = round((2.0 * mean(c["FETCH_SIZE"]) + mean(c["WRITE_SIZE"])) * 1024.0)
        kernels.append(e)
    kernels.sort(key=lambda e: -(e.get("gui_active_cycles", 0) * e["dispatches_measured"]))
    cmd = command or "python bench.py --steps 3 --warmup 2 --no-roofline --no-cpu-baseline --no-also"
    return {"note": f"rocprofv3 --pmc passes (counters only, --kernel-trace) over `{cmd}`: per kernel template instance, mean over ALL "
                    "its launches of the run (the launch mix of the benchmark step itself).  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                    "(1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); traffic_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 "
                    "wide-read correction on FETCH_SIZE).",
            "kernels": kernels}


if __name__ == "__main__":
    out = sys.argv[1]
    doc = reduce_dir(out)
    json.dump(doc, open(out + "/pmc_step.json", "w"), indent=1)
    for e in doc["kernels"][:40]:
        print(f"{e['kernel'][:78]:78s} n={e['dispatches_measured']:4d} busy {e.get('mfma_busy')} wl {e.get('mfma_busy_wave_life')} "
              f"valu/mfma {e.get('valu_per_mfma')} traffic {e.get('traffic_bytes_per_launch')}")
