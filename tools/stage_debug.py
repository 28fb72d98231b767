#!/usr/bin/env python3
"""Debug driver of the staged scheduler: one small frame through scheduler=stage against the
lane-bound kernel.  Usage (GPU box): python tools/stage_debug.py [w h spp] [key=value options]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes
from vimg_amd import hip
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
opts = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
w, h, spp = (nums + [136, 72, 12])[:3] if len(nums) >= 3 else (136, 72, 12)
hip.init(0)
s = scenes.json_scene("disney_spheres.json", res=(w, h))
p = s.default_params(samples=spp)
print("lane...", flush=True)
lane, st_lane = hip.DeviceScene(s, scheduler="lane").render_to_host(p)
print("stage upload...", flush=True)
d = hip.DeviceScene(s, scheduler="stage", **opts)
print("kernel", d.kernel, flush=True)
img, st = d.render_to_host(p)
print("stage done", st.as_dict(), flush=True)
same = (img.view(np.uint32) == lane.view(np.uint32)).all(axis=-1)
print("identical pixels", same.mean(), "stats equal", st.as_dict() == st_lane.as_dict())
