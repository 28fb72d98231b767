#!/usr/bin/env python3
"""Time the pre-step kernels (mip chain, env-map CDFs) against the host library's OpenMP loops on
the sizes of BASELINE configs 4/5 (4096x2048 HDRI, 2048x2048 textures).  Host buffers in and out on
both sides, so the GPU figure includes PCIe both ways; kernel-only time comes from rocprofv3."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vimg_amd import abi, hip, host

def host_scene_with(img, env):
    s = host.HostScene()
    t = s.add_texture_image(img)
    if env:
        s.set_background_envmap(t)
    return s

rng = np.random.default_rng(1)
out = []
for name, (w, h), env in [("texture 2048x2048 mip chain", (2048, 2048), False),
                          ("HDRI 4096x2048 mip chain + CDFs", (4096, 2048), True)]:
    img = rng.random((h, w, 3), dtype=np.float32)
    hip.build_mip_chain(img[:64, :64])          # warm-up (context, code objects)
    t0 = time.perf_counter(); host_scene_with(img, env); t_host = time.perf_counter() - t0
    t0 = time.perf_counter()
    hip.build_mip_chain(img)
    if env:
        hip.build_env_cdfs(img)
    t_gpu = time.perf_counter() - t0
    mip_bytes = w * h * 12 * (1 + 1 / 4) * 4 / 3          # every level read once and written once
    out.append({"what": name, "host_s": round(t_host, 4), "gpu_incl_pcie_s": round(t_gpu, 4),
                "algorithmic_MB": round((mip_bytes + (w * h * 12 + 2 * h * (w + 1) * 4 if env else 0)) / 1e6, 1)})
    print(json.dumps(out[-1]), flush=True)
