#!/usr/bin/env python3
"""Time the rows of SURVEY 8f that sit beside the path: the post chain (tonemap + sRGB + 8-bit) and
the heatmap integrator at BASELINE config 2's size, device buffers in and out."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes
from vimg_amd import hip, abi
import ctypes as C
s = scenes.json_scene("disney_spheres.json")
d = hip.DeviceScene(s)
p = s.default_params(samples=4)
img, _ = d.render(p)
lib = abi.hip_lib()
rgb8 = torch.empty((800, 1800, 3), dtype=torch.uint8, device="cuda")
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for tm, name in enumerate(["clamp", "agx", "reinhard", "aces"]):
    ms = timed(lambda: lib.vimg_hip_post_rgb8(C.c_void_p(img.data_ptr()), 1800, 800, tm, C.c_void_p(rgb8.data_ptr()), None))
    print(json.dumps({"what": f"post chain {name} 1800x800", "ms": round(ms, 4),
                      "GBps_algorithmic": round(1800 * 800 * 15 / ms / 1e6, 1)}), flush=True)
ms = timed(lambda: d.render_heatmap(p, 20.0, out=img), n=10)
print(json.dumps({"what": "heatmap 1800x800, 4 spp", "ms": round(ms, 4),
                  "Mrays_per_s": round(1800 * 800 * 4 / ms / 1e3, 1)}), flush=True)
