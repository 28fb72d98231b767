#!/usr/bin/env python3
"""The CU scheduler against the lane-bound kernel on small frames: same bits, same event counts.
Usage (GPU box): python tools/cu_check.py [key=value ...]   (VimgHipOptions fields for the CU launch)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes
from vimg_amd import hip
kv = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:] if "=" in a)}
hip.init(0)
ok = True
cases = [("disney_spheres.json", (136, 72), 12, {}), ("glass_in_box.json", (96, 72), 8, {}),
         ("cornell_box_spheres.json", (80, 80), 8, {}), ("disney_spheres.json", (400, 200), 16, dict(pool_segments=4)),
         ("disney_spheres.json", (640, 320), 8, dict(pool_slots=64, pool_segments=3))]
for name, res, spp, extra in cases:
    s = scenes.json_scene(name, res=res)
    p = s.default_params(samples=spp)
    ref, rst = hip.DeviceScene(s, scheduler="lane").render_to_host(p)
    d = hip.DeviceScene(s, scheduler="cu", **{**kv, **extra})
    for rep in range(2):
        img, st = d.render_to_host(p)
        same = bool((img.view(np.uint32) == ref.view(np.uint32)).all())
        print(f"{name} {res} {spp} spp {extra} rep {rep}: {d.kernel_for(p)} same bits {same}, rays {st.rays} vs {rst.rays}", flush=True)
        ok &= same and st.rays == rst.rays
    plain = d.render_to_host(p, stats=False)      # the build without diagnostics
    same_plain = bool((plain.view(np.uint32) == ref.view(np.uint32)).all())
    print("  launch without statistics: same bits", same_plain, flush=True)
    ok &= same_plain
    px = d.trace_pixel(p, res[0] // 2, res[1] // 2)
    tp_same = bool((np.asarray(px, dtype=np.float32).view(np.uint32) == ref[res[1] - 1 - res[1] // 2, res[0] // 2].view(np.uint32)).all())
    print("  trace_pixel same bits", tp_same, flush=True)
    ok &= tp_same
s = scenes.feature_scene(res=(72, 48), envmap=True, lens=True)
p = s.default_params(samples=6, depth=7)
ref, rst = hip.DeviceScene(s, scheduler="lane").render_to_host(p)
img, st = hip.DeviceScene(s, scheduler="cu", **kv).render_to_host(p)
same = bool((img.view(np.uint32) == ref.view(np.uint32)).all())
print(f"feature scene: same bits {same}, rays {st.rays} vs {rst.rays}", flush=True)
ok &= same and st.rays == rst.rays
print("CU_CHECK", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
