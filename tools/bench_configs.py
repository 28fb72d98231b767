#!/usr/bin/env python3
"""Throughput of the synthetic stand-ins of BASELINE configs 3-5 on one GPU (reduced spp; these
are measurements for DESIGN.md, not the headline bench)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scenes
import bench
from vimg_amd import hip

hip.init(0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cases = {
    "config3 (Disney array, env map, 1366x1024)": lambda: scenes.config3_scene(),
    "config4 (2 displaced meshes 627K tris, normal map, HDRI, thin lens, 1366x768)": lambda: scenes.config4_scene(),
    "config5 (1.0M tris, mip-mapped textures, normal maps, RG map, 1366x768)": lambda: scenes.config5_scene(n=700),
}
only = sys.argv[2] if len(sys.argv) > 2 else ""
for name, mk in cases.items():
    if only and only not in name:
        continue
    t0 = time.perf_counter(); s = mk(); t_build = time.perf_counter() - t0
    t0 = time.perf_counter(); d = hip.DeviceScene(s); t_up = time.perf_counter() - t0
    v = s.view.contents
    w, h = s.resolution
    p = s.default_params(samples=spp)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    d.time_renders(p, out, 1)                         # warm-up
    ms = d.time_renders(p, out, 2)
    _, st = d.render(p, out=out)                      # event counts (after the timed launches, as bench.py)
    sec = float(ms.mean()) * 1e-3
    ab = bench.algorithmic_bytes(st, w * h)
    print(json.dumps({"scene": name, "tris": v.num_tris, "bvh_nodes": v.bvh.num_nodes,
                      "bvh_depth": v.bvh.max_depth, "hbm_scene_bytes": d.bytes, "spp": spp,
                      "host_build_s": round(t_build, 2), "upload_s": round(t_up, 3),
                      "ms": round(sec * 1e3, 2), "mrays_per_s": round(st.rays / sec / 1e6, 1),
                      "rays_per_path": round(st.rays / st.paths, 3),
                      "internal_visits_per_ray": round(st.internal_visits / st.rays, 2),
                      "prim_tests_per_ray": round(st.prim_tests / st.rays, 2),
                      "algorithmic_bytes_per_ray": round(ab / st.rays, 1),
                      "algorithmic_GBps": round(ab / sec / 1e9, 1), "nan_samples": st.nan_samples}),
          flush=True)
    d.close()
