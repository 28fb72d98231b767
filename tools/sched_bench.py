#!/usr/bin/env python3
"""Frame time of one scheduler configuration on BASELINE config 2 (or a stand-in scene).
Usage (GPU box): python tools/sched_bench.py <scheduler> [spp] [scene] [key=value ...]
  scheduler: lane | pool | stage | auto;  scene: disney (default) | config3 | config4 | config5
  key=value: VimgHipOptions fields (stage_slots=..., stage_seg_len=...), tile_world=N, tile_rank=R, res=WxH"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scenes
from vimg_amd import hip
sched = sys.argv[1]
pos = [a for a in sys.argv[2:] if "=" not in a]
kv = dict(a.split("=") for a in sys.argv[2:] if "=" in a)
spp = int(pos[0]) if pos else 64
which = pos[1] if len(pos) > 1 else "disney"
tw, tr = int(kv.pop("tile_world", 1)), int(kv.pop("tile_rank", 0))
res = tuple(int(v) for v in kv.pop("res").split("x")) if "res" in kv else None
steps = int(kv.pop("steps", 2))
scene_kw = {}          # stand-in parameters: n_lat= / env=WxH (config4), n= / tex= (config5)
for k in ("n_lat", "n", "tex"):
    if k in kv:
        scene_kw[k] = int(kv.pop(k))
if "env" in kv:
    scene_kw["env"] = tuple(int(v) for v in kv.pop("env").split("x"))
opts = {k: int(v) for k, v in kv.items()}
if sched != "auto":
    opts["scheduler"] = sched
hip.init(0)
if which == "disney":
    s = scenes.json_scene("disney_spheres.json", res=res)
else:
    s = {"config3": scenes.config3_scene, "config4": scenes.config4_scene,
         "config5": lambda **kw: scenes.config5_scene(**{"n": 700, **kw})}[which](**scene_kw)
d = hip.DeviceScene(s, **opts)
p = s.default_params(samples=spp, tile_rank=tr, tile_world=tw)
w, h = s.resolution
n_out = w * h if tw == 1 else d.shard_pixels(p)
out = torch.empty((n_out, 3), dtype=torch.float32, device="cuda")
diag = bool(os.environ.get("VIMG_HIP_DIAG"))
if diag:
    _, st = d.render(p, out=out)  # (a diagnostics run: the statistics launch prints them; the times below then carry the +1 %)
d.time_renders(p, out, 1)        # warm-up (first launch: pool allocation, cold instruction cache)
ms = d.time_renders(p, out, steps)
if not diag:
    _, st = d.render(p, out=out)  # event counts AFTER the timed launches: the statistics build uses scratch, and a queue
                                  # that has once dispatched a kernel with scratch sets it up for every later dispatch (+1 %)
sec = float(ms.min()) * 1e-3
print(json.dumps({"scheduler": sched, "kernel": d.kernel_for(p), "scene": which, "res": [w, h], "spp": spp, "tile_world": tw,
                  "opts": {**opts, **scene_kw}, "ms": [round(float(m), 2) for m in ms], "mrays_per_s": round(st.rays / sec / 1e6, 1),
                  "rays": st.rays}), flush=True)
