#!/usr/bin/env python3
"""Mrays/s of disney_spheres against image size at fixed spp: how much of a frame is the tail
(the last pixels of each wave's pool / each lane) and how much the steady state."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes
from vimg_amd import hip
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for res in [(904, 400), (1800, 800), (2544, 1128), (3600, 1600), (5088, 2264)]:
    s = scenes.json_scene("disney_spheres.json", res=res)
    d = hip.DeviceScene(s)
    p = s.default_params(samples=spp)
    out, st = d.render(p)
    ms = d.time_renders(p, out, 2)
    print(f"{res[0]}x{res[1]}  {d.kernel}  {ms.min():9.2f} ms  {st.rays / ms.min() / 1e3:8.1f} Mrays/s", flush=True)
