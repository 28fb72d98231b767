#!/usr/bin/env python3
"""Random corner configurations of the CU scheduler against the lane-bound kernel: same bits, same
event counts, no watchdog.  Usage (GPU box): python tools/stress_cu.py [iterations] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes
from vimg_amd import hip
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
hip.init(0)
cases = []
for name, res, spp, kw in (("disney_spheres.json", (200, 96), 10, {}), ("glass_in_box.json", (112, 80), 6, {}),
                           ("cornell_box_spheres.json", (96, 96), 6, {})):
    s = scenes.json_scene(name, res=res)
    cases.append((name, s, s.default_params(samples=spp, **kw)))
s = scenes.config4_scene(res=(96, 54), n_lat=48, env=(128, 64))
cases.append(("config4 stand-in", s, s.default_params(samples=6, depth=10)))
s = scenes.feature_scene(res=(72, 48), envmap=True, lens=True)
cases.append(("feature", s, s.default_params(samples=6, depth=7)))
refs = [hip.DeviceScene(s, scheduler="lane").render_to_host(p) for _, s, p in cases]
bad = 0
for it in range(n_iter):
    k = rnd.randrange(len(cases))
    name, s, p = cases[k]
    walkers = rnd.choice([1, 2, 5, 9, 10, 13, 15, 16])
    opts = dict(scheduler="cu", pool_slots=rnd.choice([8, 16, 24, 40, 64, 96, 200, 512, 2048]), cu_walkers=walkers,
                pool_refill=rnd.choice([1, 2, 7, 16, 33, 64]), pool_starve=rnd.choice([1, 4, 16, 40, 64]),
                cu_patience=rnd.choice([0, 1, 4, 30]), pool_vbatch=rnd.choice([1, 8, 32, 64]), pool_classes=rnd.choice([1, 2, 3]),
                pool_segments=rnd.choice([1, 2, 3, 5]), cu_flex=rnd.choice([1, 17, 33, 49, 0 if walkers < 16 else 1, 32]),
                lds_stack=rnd.choice([1, 2, 4, 32]), cu_join=rnd.choice([1, 8, 64]), cu_lowwater=rnd.choice([1, 16, 64, 4096]),
                lds_leaf=rnd.choice([0, -1]), cu_sleep=rnd.choice([1, 4, 32]))
    d = hip.DeviceScene(s, **opts)
    ok = True
    for stats in (True, False, True):
        r = d.render_to_host(p, stats=stats)
        img = r[0] if stats else r
        ok &= bool((img.view(np.uint32) == refs[k][0].view(np.uint32)).all())
        if stats:
            ok &= r[1].as_dict() == refs[k][1].as_dict()
    d.close()
    if not ok:
        bad += 1
        print("MISMATCH", name, opts, flush=True)
    if it % 25 == 24:
        print(f"{it + 1} configurations, {bad} bad", flush=True)
print("STRESS_CU", "OK" if bad == 0 else f"FAILED ({bad})")
sys.exit(0 if bad == 0 else 1)
