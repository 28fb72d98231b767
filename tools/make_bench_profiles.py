#!/usr/bin/env python3
"""From the PMC passes of tools/gpu_check.sh (bench.py --steps 1 --warmup 0 --spp 512: one stats
launch + one timed launch, so every counter is summed over TWO launches of the frame) to the two
small files bench.py reads: profiles/valu.json (VALU wave instructions and lane utilisation per
launch) and profiles/traffic.json (HBM-side bytes per launch: FETCH_SIZE x 2 - the gfx950
correction of MI355X_MICROARCH.md "HBM" - + WRITE_SIZE, both in KiB units of rocprofv3).
Usage: tools/make_bench_profiles.py gpurun_out/<tag> profiles/<name>"""
import json, os, sys
src, dst = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = json.load(open(os.path.join(src, "pmc_summary.json")))
k = s["counters"]
launches = 2
workload = "disney_spheres.json, mis integrator, 512 spp, 1800x800"
valu = {"workload": workload, "kernel": s["dispatch"],
        "valu_wave_insts_per_launch": k["SQ_INSTS_VALU"] / launches,
        "valu_lane_utilization": round(k["SQ_THREAD_CYCLES_VALU"] / (k["SQ_ACTIVE_INST_VALU"] * 64), 4),
        "salu_insts_per_launch": k.get("SQ_INSTS_SALU", 0) / launches,
        "lds_bank_conflict_share": round(k["SQ_LDS_BANK_CONFLICT"] / k["SQ_LDS_IDX_ACTIVE"], 4) if k.get("SQ_LDS_IDX_ACTIVE") else None,
        "wait_any_share_of_wave_cycles": round(k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"], 4),
        "source": os.path.join(dst, "pmc_summary.json") + " (rocprofv3 --pmc, separate passes)"}
traffic = {"workload": workload,
           "hbm_bytes_per_launch": int((k["FETCH_SIZE"] * 2 + k["WRITE_SIZE"]) * 1024 / launches),
           "fetch_bytes_per_launch": int(k["FETCH_SIZE"] * 2 * 1024 / launches),
           "write_bytes_per_launch": int(k["WRITE_SIZE"] * 1024 / launches),
           "note": "FETCH_SIZE doubled (gfx950 reports half of wide reads), WRITE_SIZE as is; L2 <-> fabric, Infinity-Cache hits included",
           "source": os.path.join(dst, "pmc_summary.json")}
os.makedirs(os.path.join(ROOT, dst), exist_ok=True)
json.dump(s, open(os.path.join(ROOT, dst, "pmc_summary.json"), "w"), indent=1)
json.dump(valu, open(os.path.join(ROOT, "profiles", "valu.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(valu)); print(json.dumps(traffic))
