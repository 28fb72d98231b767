#!/usr/bin/env python3
"""Cross-check of the retired schedulers (development build of the library, VIMG_HIP_LIB) against
the lane-bound kernel: run once, in a process of its own, by
tests/test_gpu_parity.py::test_dev_build_schedulers_give_the_same_bits."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
assert os.environ.get("VIMG_HIP_LIB", "").endswith(os.path.join("dev", "libvimg_hip.so")), "VIMG_HIP_LIB must name the development build"
import test_gpu_parity as T  # noqa: E402
from vimg_amd import hip  # noqa: E402

hip.init(0)
both = {**T.DEV_SCHEDULES, "cu": T.SCHEDULES["cu"], "cu/5": T.SCHEDULES["cu/5"]}
for scene_name in ("disney_spheres.json", "glass_in_box.json", "feature"):
    s, p = T.scheduler_scene(scene_name)
    T.check_schedules_against_lane(s, p, both, scene_name)
    print("ok", scene_name, flush=True)
feature_pick = ("pool", "pool/5", "pool4", "pool4/5", "pool4/4", "pool4/stack1", "pool4/stack3", "pool4g", "pool4g/5",
                "pool4g/few", "stage", "stage/few")
for case, mk in T.FEATURE_CASES.items():
    s, kw = mk()
    T.check_schedules_against_lane(s, s.default_params(**kw), {k: T.DEV_SCHEDULES[k] for k in feature_pick}, case, twice=False)
    print("ok", case, flush=True)
print("DEV_SCHEDULERS OK")
