#!/usr/bin/env python3
"""GPU builders (LBVH, PLOC) against the host sweep-SAH builder on the config-5 stand-in (519 K
triangles at the default size): build time (primitive bounds, host buffers in and out) and what the
tree costs the renderer (Mrays/s, node visits per ray).  VIMG_HIP_DIAG=1 prints the builder's own
split of its time (kernels / layout + download).
Usage (GPU box): python tools/bench_lbvh.py [spp] [sweep | LBVH | PLOC]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from vimg_amd import hip, abi
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
s = scenes.config5_scene()
p = s.default_params(samples=spp, depth=2 ** 32 - 1)
out = {}
only = sys.argv[2] if len(sys.argv) > 2 else ""      # e.g. "PLOC": that builder alone
for name in ("sweep SAH (host)", "LBVH (GPU)", "PLOC (GPU)"):
    if only and not name.startswith(only):
        continue
    if name.startswith("LBVH"):          # (a first build pays for the kernels' loading: not timed)
        s.build_bvh_with(hip.lbvh_builder())
    elif name.startswith("PLOC"):
        s.build_bvh_with(hip.ploc_builder())
    t0 = time.perf_counter()
    if name.startswith("LBVH"):
        s.build_bvh_with(hip.lbvh_builder())
    elif name.startswith("PLOC"):
        s.build_bvh_with(hip.ploc_builder())
    else:
        s.build_bvh(abi.BVH_SWEEP)
    t_build = time.perf_counter() - t0
    d = hip.DeviceScene(s)
    img, st = d.render(p)
    ms = float(d.time_renders(p, img, 2).min())
    b = s.view.contents.bvh
    out[name] = {"build_s": round(t_build, 4), "nodes": int(b.num_nodes), "depth": int(b.max_depth),
                 "render_ms": round(ms, 2), "mrays_per_s": round(st.rays / ms / 1e3, 1),
                 "internal_visits_per_ray": round(st.internal_visits / st.rays, 2),
                 "prim_tests_per_ray": round(st.prim_tests / st.rays, 2), "kernel": d.kernel}
    print(json.dumps({name: out[name]}), flush=True)
