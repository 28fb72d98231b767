#!/usr/bin/env python3
"""Soak: render BASELINE config 2 at full size and sample count several times with the pooled
scheduler (segments travelling through the per-pixel records) and compare every frame bit for
bit with the lane-bound kernel's.  Usage: tools/soak.py [frames] [spp] [scene: disney | config3 | config4 | config5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scenes
from vimg_amd import hip
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 512
which = sys.argv[3] if len(sys.argv) > 3 else "disney"
s = scenes.json_scene("disney_spheres.json") if which == "disney" else \
    {"config3": scenes.config3_scene, "config4": scenes.config4_scene, "config5": lambda: scenes.config5_scene(n=700)}[which]()
p = s.default_params(samples=spp)
os.environ["VIMG_HIP_SCHED"] = "lane"
ref, st0 = hip.DeviceScene(s).render(p)
del os.environ["VIMG_HIP_SCHED"]
d = hip.DeviceScene(s)
print("reference:", st0.rays, "rays; pooled kernel:", d.kernel, flush=True)
bad = 0
for i in range(frames):
    img, st = d.render(p)
    same = bool(torch.equal(img, ref)) and st.rays == st0.rays
    bad += (not same)
    print(f"frame {i}: {'identical' if same else 'DIFFERENT'}", flush=True)
sys.exit(1 if bad else 0)
