#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc CSVs of one tag directory (tools/gpu_check.sh, tools/pmc_sched.sh)
into one JSON: every counter summed over the render-kernel dispatches of its pass, how many
dispatches that was, and the dispatch record (kernel name, grid, registers, scratch).
Usage: tools/pmc_summary.py gpurun_out/<tag> [run.json]   (run.json: the JSON line of the profiled
command - workload, spp, kernel, ms per launch - is stored beside the counters)"""
import collections
import csv
import glob
import json
import re
import sys

out_dir = sys.argv[1]
agg = collections.OrderedDict()
rows = collections.Counter()
meta = {}
KERNELS = ("render_kernel", "render_pool_kernel", "render_pool4_kernel", "render_stage_kernel", "render_cu_kernel")
def wanted(name):
    # the CU scheduler has a build for statistics launches (<..., true>) beside the one frames are timed on
    # (<..., false>): a profile is of the timed one
    if "render_cu_kernel" in name:
        return re.search(r"16, 4, false, \d>", name) is not None      # <TEX, DEEP, 16, 4, DIAG = false, EARLY>
    return any(n in name for n in KERNELS)


for f in sorted(glob.glob(f"{out_dir}/pmc*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if wanted(r["Kernel_Name"]):
            agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            rows[(r["Counter_Name"], r["Dispatch_Id"])] += 1
            meta = {k: r[k] for k in ["Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
                                      "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count"]}
launches = {}
for (name, disp) in rows:
    launches[name] = launches.get(name, 0) + 1
d = {}
g = agg.get
if g("SQ_ACTIVE_INST_VALU"):
    d["valu_lane_utilization"] = g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64)
    d["valu_active_share_of_wave_cycles"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")
    d["wait_any_share"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
if g("SQ_LDS_IDX_ACTIVE"):
    d["lds_bank_conflict_share"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
if g("SQC_ICACHE_REQ"):
    d["icache_miss_share"] = g("SQC_ICACHE_MISSES") / g("SQC_ICACHE_REQ")
n = max(launches.values()) if launches else 0
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None and n:
    d["hbm_side_bytes_per_launch"] = (g("FETCH_SIZE") * 2 + g("WRITE_SIZE")) * 1024 / n
out = {"dispatch": meta, "launches_per_pass": n, "counters": agg, "derived": d,
       "note": "each counter is summed over the dispatches of the timed render kernel in its pass (launches_per_pass of them; the "
               "statistics launch that follows them runs another build and is not counted); FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM)"}
if len(sys.argv) > 2:
    try:
        out["run"] = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    except (OSError, ValueError, IndexError):
        pass
print(json.dumps(out, indent=1))
