#!/usr/bin/env python3
"""Usage (GPU box): [STRESS_SCHED=pool4|pool4g|pool] python tools/stress_pool.py
Stress of the pooled scheduler's corner configurations: tiny and odd pool sizes, batch sizes,
refill / starvation thresholds, class counts and segment counts, on three small scenes; every image
must equal the lane-bound kernel's bit for bit (and no configuration may trip the watchdog)."""
import itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes
from vimg_amd import hip
cases = [("disney", scenes.json_scene("disney_spheres.json", res=(104, 56)), dict(samples=9)),
         ("feature", scenes.feature_scene(res=(56, 40)), dict(samples=5, depth=6)),
         # a tree beyond LDS (the DEEP builds), env map, thin lens, normal map
         ("deep", scenes.config4_scene(res=(64, 36), n_lat=20, env=(64, 32)), dict(samples=4, depth=8))]
knobs = {"VIMG_HIP_POOL_SLOTS": ["8", "17", "64"], "VIMG_HIP_POOL_SEGMENTS": ["1", "2", "7"],
         "VIMG_HIP_POOL_VBATCH": ["1", "13", "64"], "VIMG_HIP_POOL_REFILL": ["1", "64"],
         "VIMG_HIP_POOL_STARVE": ["1", "64"], "VIMG_HIP_POOL_CLASSES": ["1", "2", "3"]}
if os.environ.get("STRESS_SCHED") == "pool4g":   # the group build: who takes a full batch
    knobs["VIMG_HIP_POOL_GBREAK"] = ["0", "64"]
bad = n = 0
for name, s, kw in cases:
    p = s.default_params(**kw)
    os.environ["VIMG_HIP_SCHED"] = "lane"
    ref, st0 = hip.DeviceScene(s).render_to_host(p)
    os.environ["VIMG_HIP_SCHED"] = os.environ.get("STRESS_SCHED", "pool4")
    for combo in itertools.product(*knobs.values()):
        for k, v in zip(knobs, combo):
            os.environ[k] = v
        img, st = hip.DeviceScene(s).render_to_host(p)
        n += 1
        if not (np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and st.as_dict() == st0.as_dict()):
            bad += 1
            print("MISMATCH", name, dict(zip(knobs, combo)), flush=True)
    print(f"{name}: {n} configurations so far, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
