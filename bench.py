#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the MIS path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full render of the workload (one pass of the hot path over one frame of
synthetic-free input: the reference's own disney_spheres scene).  N == 1 runs BASELINE.json
configs[1] exactly: scenes/disney_spheres.json, mis integrator, 512 spp, 1800x800.  N > 1 is
launched under torch.distributed.run, one rank per GPU: the scene is replicated, 8x8 image tiles are
dealt to the ranks (tile t -> rank t % N), and the per-rank framebuffer slabs are gathered once
per step with RCCL (all_gather) and de-interleaved — no collective on the data path.  Scaling is
WEAK: the image grows with sqrt(N) per axis so that every GPU keeps 1800x800 pixels of work
(pixels are the unit of parallelism: the reference draws one sequential PCG stream per pixel).
`--strong` keeps the image at 1800x800 instead (reported in DESIGN.md, not the default).

Prints ONE JSON line on rank 0 (see the contract in the task statement).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCENE_JSON = os.path.join(ROOT, "tests", "golden", "scenes", "disney_spheres.json")
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# Algorithmic bytes per event on the reference's own storage layout
B_ROOT = 8 + 24            # root node + its box, per BVH query
B_INTERNAL = 8 + 48        # node header + sibling AABB pair, per internal-node visit
B_LEAF = 8                 # node header, per leaf visit
B_TRI = 4 + 48             # obj index + indices and vertices, per triangle test
B_SPHERE = 4 + 16          # obj index + centre/radius, per sphere test
B_MATERIAL = 72            # material record, per path vertex (one per next-event estimation)
B_PIXEL = 12               # framebuffer, once per pixel


def algorithmic_bytes(stats, pixels):
    tri = stats.prim_tests - stats.sphere_tests
    return (stats.rays * B_ROOT + stats.internal_visits * B_INTERNAL + stats.leaf_visits * B_LEAF
            + tri * B_TRI + stats.sphere_tests * B_SPHERE + stats.shadow_rays * B_MATERIAL
            + pixels * B_PIXEL)


def load_scene(res):
    import vimg_amd
    with open(SCENE_JSON) as f:
        d = json.load(f)
    d["camera"]["resolution"] = [int(res[0]), int(res[1])]
    return vimg_amd.HostScene.from_json_text(json.dumps(d))


def cpu_share():
    """Host cores this job may use: min(affinity mask, cgroup CPU quota).  The GPU box exposes
    all hardware threads of the node but gives a one-GPU job a 16-CPU share."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    if n > 64:            # no quota visible: stay within the documented per-GPU share
        n = 16
    return n


def measured_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 PMC pass of this same workload."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if t.get("workload") == workload:
            return int(t["hbm_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_baseline(spp):
    """The CPU port (oracle/) timed on this box's host cores on a bounded sample of the same
    workload: the same 1800x800 pixels, the first `spp` of the 512 samples."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    scene = load_scene((1800, 800))
    params = scene.default_params(samples=spp)
    t0 = time.perf_counter()
    share = cpu_share()
    _, st, threads = O.render(scene, params, threads=share)
    dt = time.perf_counter() - t0
    return {"value": round(st.rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": int(threads),
            "kind": "port", "seconds": round(dt, 2),
            "sample": f"disney_spheres.json 1800x800, first {spp} of 512 spp (same pixels, same "
                      f"seeds), oracle/liboracle.so on {threads} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=512)
    ap.add_argument("--strong", action="store_true", help="keep 1800x800 total (strong scaling)")
    ap.add_argument("--cpu-spp", type=int, default=64, help="samples of the CPU baseline leg (about 15 s on 16 threads)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 path (host-staged gather), not a measurement")
    ap.add_argument("--verify", action="store_true",
                    help="N>1: rank 0 also renders the whole frame alone and compares bit for bit")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal: every rank uses GPU 0 (needs --backend gloo)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from vimg_amd import hip, dist as vdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run "
                             "(one rank per GPU)")
        args.gpus = world
    dev_index = 0 if args.one_device else local_rank
    torch.cuda.set_device(dev_index)
    hip.init(dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")

    n = world
    if args.strong or n == 1:
        res = (1800, 800)
    else:
        res = (8 * round(1800 * math.sqrt(n) / 8), 8 * round(800 * math.sqrt(n) / 8))
    scene = load_scene(res)
    dev = hip.DeviceScene(scene)
    kernel_name = dev.kernel
    params = scene.default_params(samples=args.spp, tile_rank=rank, tile_world=n)
    W, H = res
    # a dedicated (non-default) stream: the kernels are launched on it and the HIP events that
    # time them are recorded on it
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)

    if n == 1:
        frame = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        slab = frame
    else:
        stride = vdist.shard_stride_pixels(W, H, n)
        slab = torch.zeros((stride, 3), dtype=torch.float32, device="cuda")
        gathered = torch.empty((n * stride, 3), dtype=torch.float32, device="cuda")
        frame = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")

    def step(ev=None):
        if ev is not None:
            ev[0].record(stream)
        dev.render_async(params, slab, stream=stream)
        if ev is not None:
            ev[1].record(stream)
        if n > 1:
            if args.backend == "nccl":
                dist.all_gather_into_tensor(gathered, slab)     # RCCL over xGMI, once per frame
            else:
                host = torch.empty(gathered.shape, dtype=torch.float32)
                dist.all_gather_into_tensor(host, slab.cpu())
                gathered.copy_(host)
            dev.assemble_shards(gathered, n, stride, out=frame, stream=stream)

    # event counts of one step (deterministic: same seeds every step); not timed
    _, st = dev.render(params, out=slab, stats=True, stream=stream)
    local_pixels = st.paths // args.spp
    red_dev = "cuda" if args.backend == "nccl" else "cpu"
    counts = torch.tensor([st.rays, st.paths, algorithmic_bytes(st, local_pixels)],
                          dtype=torch.float64, device=red_dev)
    if n > 1:
        dist.all_reduce(counts)
    total_rays, total_paths = float(counts[0]), float(counts[1])
    local_bytes = algorithmic_bytes(st, local_pixels)

    for _ in range(args.warmup):
        step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]
    if n > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    torch.cuda.synchronize()
    if n > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if n > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0])
    kernel_ms = sum(a.elapsed_time(b) for a, b in events) / args.steps

    if args.verify and n > 1 and rank == 0:
        alone, _ = dev.render(scene.default_params(samples=args.spp), stream=stream)
        torch.cuda.synchronize()
        assert torch.equal(alone, frame), "assembled shards differ from the single-GPU frame"
        print("verify: assembled frame is bit-identical to the single-GPU frame", file=sys.stderr)
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rays * args.steps / elapsed / 1e6
        achieved = local_bytes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "Mrays/sec (primary+secondary) at 512 spp",
            "value": round(value, 2),
            "unit": "Mrays/s",
            "n_gpus": n,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong" if (args.strong and n > 1) else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "reference scene file (scenes/disney_spheres.json), no external assets",
            "config": {
                "workload": f"disney_spheres.json, mis integrator, {args.spp} spp, {W}x{H}"
                            + ("" if n == 1 else f" tile-sharded over {n} GPUs "
                               f"({'fixed image' if args.strong else '1800x800 pixels per GPU'})"),
                "integrator": "mis", "spp": args.spp, "resolution": [W, H], "depth": "unbounded",
                "bvh": "sweep SAH (host)", "sharding": f"tiles%{n}",
            },
            "mpaths_per_s": round(total_paths * args.steps / elapsed / 1e6, 3),
            "rays_per_path": round(total_rays / total_paths, 4),
            "kernel_ms": round(kernel_ms, 3),
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": measured_traffic(f"disney_spheres.json, mis integrator, {args.spp} spp, {W}x{H}")
                if n == 1 else None,
                "kernel": kernel_name,
                "bytes_per_launch": int(local_bytes),
                "note": "algorithmic bytes on the reference layout (SURVEY.md 8d); the 2 KB scene "
                        "is LDS/L1 resident, so the kernel is VALU/latency bound, not HBM bound",
            },
        }
        if n == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.cpu_spp)
        print(json.dumps(out), flush=True)
    if n > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
