#!/usr/bin/env python3
"""Stage profile of render_pool_kernel: `make prof` builds the library with s_memtime timers around
every stage (lane 0 of each wave, summed over waves); this renders BASELINE config 2 at a few
samples per pixel with that build and prints the share of wave time per stage.
Usage (GPU box): [VIMG_PROFILE_SCENE=config5] python tools/stage_profile.py [spp [width height]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["VIMG_HIP_DIAG"] = "1"
os.environ["VIMG_HIP_LIB"] = os.path.join(ROOT, "v-img_amd", "lib", "prof", "libvimg_hip.so")
os.environ.setdefault("VIMG_HIP_SCHED", "pool")   # the stage timers live in render_pool_kernel
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from vimg_amd import hip
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
res = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else None
which = os.environ.get("VIMG_PROFILE_SCENE", "disney_spheres")
if which == "disney_spheres":
    s = scenes.json_scene("disney_spheres.json", res=res)
else:   # config3 / config4 / config5 stand-ins at their BASELINE sizes
    s = {"config3": scenes.config3_scene, "config4": scenes.config4_scene, "config5": scenes.config5_scene}[which]()
d = hip.DeviceScene(s)
p = s.default_params(samples=spp)
img, st = d.render_to_host(p)
print(st.as_dict())
