#!/usr/bin/env python3
"""From the PMC passes of tools/gpu_check.sh (bench.py --steps 1 --warmup 0 --spp 512: one stats
launch, which runs the scheduler's DIAG build and is left out, + one timed launch) to the two
small files bench.py reads: profiles/valu.json (VALU wave instructions and lane utilisation per
launch) and profiles/traffic.json (HBM-side bytes per launch: FETCH_SIZE x 2 - the gfx950
correction of MI355X_MICROARCH.md "HBM" - + WRITE_SIZE, both in KiB units of rocprofv3).
Both carry what ties them to a build: the kernel's name as vimg_hip_launch_kernel reports it, the
sha256 of the library they were taken on, the workload, its rays per launch and the kernel time
of the trace pass; bench.py uses them only when all of these match the run (else "pmc": "stale").
Usage (here, after the gpurun call): tools/make_bench_profiles.py gpurun_out/<tag> profiles/<name> <bench line json>"""
import csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
src, dst, line_file = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = json.load(open(os.path.join(src, "pmc_summary.json")))
line = json.loads([l for l in open(line_file) if l.startswith("{")][-1])
k = s["counters"]
launches = s["launches_per_pass"]
workload = line["config"]["workload"]
kernel_name = line["roofline"]["kernel"]
assert kernel_name.split("<")[0] in s["dispatch"]["Kernel_Name"], (kernel_name, s["dispatch"]["Kernel_Name"])
rays = round(line["value"] * 1e6 * line["ms_per_step"] * 1e-3)
ms = None
for f in glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if kernel_name.split("<")[0] in r["Name"] and ("render_cu_kernel" not in r["Name"] or re.search(r"16, 4, false, \d>", r["Name"])):
            ms = float(r["AverageNs"]) * 1e-6
tie = {"workload": workload, "kernel_name": kernel_name, "library_sha256": bench.library_fingerprint(),
       "spp": line["config"]["spp"], "rays_per_launch": rays, "ms_per_launch_under_rocprof": ms,
       "dispatch": s["dispatch"]}
valu = dict(tie, **{
        "valu_wave_insts_per_launch": k["SQ_INSTS_VALU"] / launches,
        "valu_lane_utilization": round(k["SQ_THREAD_CYCLES_VALU"] / (k["SQ_ACTIVE_INST_VALU"] * 64), 4),
        "salu_insts_per_launch": k.get("SQ_INSTS_SALU", 0) / launches,
        "lds_bank_conflict_share": round(k["SQ_LDS_BANK_CONFLICT"] / k["SQ_LDS_IDX_ACTIVE"], 4) if k.get("SQ_LDS_IDX_ACTIVE") else None,
        "wait_any_share_of_wave_cycles": round(k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"], 4),
        "source": os.path.join(dst, "pmc_summary.json") + " (rocprofv3 --pmc, separate passes)"})
traffic = dict(tie, **{
           "hbm_bytes_per_launch": int((k["FETCH_SIZE"] * 2 + k["WRITE_SIZE"]) * 1024 / launches),
           "fetch_bytes_per_launch": int(k["FETCH_SIZE"] * 2 * 1024 / launches),
           "write_bytes_per_launch": int(k["WRITE_SIZE"] * 1024 / launches),
           "note": "FETCH_SIZE doubled (gfx950 reports half of wide reads), WRITE_SIZE as is; L2 <-> fabric, Infinity-Cache hits included",
           "source": os.path.join(dst, "pmc_summary.json")})
os.makedirs(os.path.join(ROOT, dst), exist_ok=True)
s["tie"] = tie
json.dump(s, open(os.path.join(ROOT, dst, "pmc_summary.json"), "w"), indent=1)
json.dump(valu, open(os.path.join(ROOT, "profiles", "valu.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(valu)); print(json.dumps(traffic))
