#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs of tools/gpu_check.sh into one JSON (sums over the
render_kernel dispatches of each pass)."""
import collections
import csv
import glob
import json
import sys

out_dir = sys.argv[1]
agg = collections.OrderedDict()
meta = {}
for f in sorted(glob.glob(f"{out_dir}/pmc*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if any(n in r["Kernel_Name"] for n in ("render_kernel", "render_pool_kernel", "render_pool4_kernel", "render_stage_kernel")):
            agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            meta = {k: r[k] for k in ["Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
                                      "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count"]}
d = {}
g = agg.get
if g("SQ_ACTIVE_INST_VALU"):
    d["valu_lane_utilization"] = g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64)
    d["valu_active_share_of_wave_cycles"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")
    d["wait_any_share"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
print(json.dumps({"dispatch": meta, "counters": agg, "derived": d,
                  "note": "each counter is summed over the render_kernel dispatches of its pass "
                          "(bench.py --steps 1 --warmup 0 --spp 512: one stats launch + one timed launch)"},
                 indent=1))
