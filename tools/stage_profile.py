#!/usr/bin/env python3
"""Stage profile of render_pool_kernel: `make prof` builds the library with s_memtime timers around
every stage (lane 0 of each wave, summed over waves); this renders BASELINE config 2 at a few
samples per pixel with that build and prints the share of wave time per stage.
Usage (GPU box): python tools/stage_profile.py [spp [width height]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["VIMG_HIP_DIAG"] = "1"
os.environ["VIMG_HIP_LIB"] = os.path.join(ROOT, "v-img_amd", "lib", "prof", "libvimg_hip.so")
os.environ.setdefault("VIMG_HIP_POOL", "1")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from vimg_amd import hip
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
res = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else None
s = scenes.json_scene("disney_spheres.json", res=res)
d = hip.DeviceScene(s)
p = s.default_params(samples=spp)
img, st = d.render_to_host(p)
print(st.as_dict())
