#!/usr/bin/env python3
"""Whole-call timing through the host-buffer entry points: scene upload (H2D) and
vimg_hip_render_to_host (kernel + D2H of the W*H*3 float framebuffer)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from vimg_amd import hip
hip.init(0)
s = scenes.json_scene("disney_spheres.json")
t0 = time.perf_counter(); d = hip.DeviceScene(s); t1 = time.perf_counter()
p = s.default_params()
d.render_to_host(s.default_params(samples=1), stats=False)          # warm-up (module load)
t2 = time.perf_counter(); img, st = d.render_to_host(p, stats=False), None; t3 = time.perf_counter()
print(f"upload {1e3*(t1-t0):.2f} ms ({d.bytes} bytes in HBM); render_to_host 512 spp {1e3*(t3-t2):.2f} ms")
