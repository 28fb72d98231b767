#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the MIS path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full render of the workload: BASELINE.json configs[1] exactly
(scenes/disney_spheres.json, mis integrator, 512 spp, 1800x800), for every N.

N == 1 renders the frame on one GPU.  N > 1: `python bench.py --gpus N` starts N fresh child
processes itself (python -m torch.distributed.run, one rank per GPU, before anything in this
process has touched the GPU) and relays rank 0's JSON line; when it is already running under
torch.distributed.run (RANK / WORLD_SIZE in the environment) it is a rank.  The scene is
replicated, 8x8 image tiles are dealt to the ranks (tile t -> rank t % N, the reference's static
interleave, include/integrators.h:57-65,101), and the per-rank framebuffer slabs are gathered once
per step with RCCL (one all_gather of equal padded slabs) and de-interleaved — no collective on
the data path.  Scaling is STRONG: the frame stays 1800x800 at 512 spp, `value` is the rays of
that frame over the slowest rank's time; rank 0 also times the whole frame alone once (untimed
extra) so that the line carries t1_ms and efficiency = T1 / (N * T_N), and the weak-scaling figure
rides along under "weak" (the frame grown with sqrt(N) per axis: every GPU keeps 1800x800 pixels).
`--weak` makes that frame the headline instead.

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
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2     # wave64 VALU instructions/s (MI355X_MICROARCH.md:54,473)
VALU_LANE_PEAK = VALU_ISSUE_PEAK * 64     # lane-ops/s

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


def library_fingerprint(path=None):
    """sha256 of the HIP library in use: what a committed PMC pass is tied to (a kernel or policy
    change makes a new library, and the counts of the old one must not be divided by the new time)."""
    import hashlib
    if path is None:
        path = os.environ.get("VIMG_HIP_LIB") or os.path.join(ROOT, "v-img_amd", "lib", "libvimg_hip.so")
    try:
        h = hashlib.sha256()
        with open(path, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 20), b""):
                h.update(chunk)
        return h.hexdigest()
    except OSError:
        return None


def load_pmc_profile(name, workload, kernel, lib_sha256, profiles_dir=None):
    """One of the two committed PMC summaries (profiles/valu.json, profiles/traffic.json) - but only
    when it was taken on THIS workload, THIS kernel and THIS build of the library.  Returns
    (profile or None, "fresh" | "stale: <why>" | "absent")."""
    path = os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), name)
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        return None, "absent"
    for key, want in (("workload", workload), ("kernel_name", kernel), ("library_sha256", lib_sha256)):
        if t.get(key) != want:
            return None, f"stale: {key} of profiles/{name} is {str(t.get(key))[:48]!r}, this run has {str(want)[:48]!r}"
    return t, "fresh"


def build_roofline(kernel_ms, local_bytes, local_rays, kernel_name, workload, lib_sha256, profiles_dir=None):
    """The `roofline` object of the JSON line for one rank's launch.  The scene of the headline
    workload (5.8 KB) is served from LDS, so the bound is vector issue, not HBM
    (MI355X_MICROARCH.md:54,473: a wave64 VALU instruction takes 2 cycles on a SIMD-32, 1024 SIMDs
    x 2.4 GHz / 2 = 1.2288e12 wave instructions/s = 7.86e13 lane-ops/s).  The instruction counts are
    a property of the deterministic workload and come from the committed PMC pass - divided by the
    time measured live, and only when that pass belongs to this kernel and this library ("pmc":
    "fresh"); otherwise the line falls back to the labelled algorithmic-bytes figure and says why.
    A shard (N > 1) executes the frame's instructions per ray: the counts scale with its rays."""
    achieved = local_bytes / (kernel_ms * 1e-3) / 1e9
    vp, v_state = load_pmc_profile("valu.json", workload, kernel_name, lib_sha256, profiles_dir)
    tp, t_state = load_pmc_profile("traffic.json", workload, kernel_name, lib_sha256, profiles_dir)
    roof = {
        "bound": "valu",
        "kernel": kernel_name,
        "pmc": "fresh" if (v_state == t_state == "fresh") else ("absent" if (v_state == t_state == "absent") else "stale"),
        "hbm": {
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "bytes_per_launch": int(local_bytes),
            "note": "ALGORITHMIC bytes on the reference layout (SURVEY.md 8d); served by LDS/L1 "
                    "here, not by HBM - see traffic for the bytes that do cross the L2",
        },
        "traffic": None,
    }
    if roof["pmc"] != "fresh":
        roof["pmc_detail"] = {"valu": v_state, "traffic": t_state}
        vp = tp = None      # both or neither: the two passes describe one build
    share = 1.0
    if vp and vp.get("rays_per_launch"):
        share = local_rays / float(vp["rays_per_launch"])      # a shard's part of the frame's instructions
    if tp:
        roof["traffic"] = int(tp["hbm_bytes_per_launch"] * (local_rays / float(tp["rays_per_launch"]) if tp.get("rays_per_launch") else 1.0))
    if vp:
        wave_insts = vp["valu_wave_insts_per_launch"] * share
        wave_rate = wave_insts / (kernel_ms * 1e-3)
        lane_rate = wave_rate * 64.0 * vp["valu_lane_utilization"]
        roof.update({
            "achieved": round(lane_rate / 1e12, 3), "peak": round(VALU_LANE_PEAK / 1e12, 2),
            "unit": "Tlane-op/s", "frac": round(lane_rate / VALU_LANE_PEAK, 4),
            "valu": {
                "wave_insts_per_launch": wave_insts,
                "wave_insts_per_s": round(wave_rate, 0),
                "issue_peak_per_s": VALU_ISSUE_PEAK,
                "issue_frac": round(wave_rate / VALU_ISSUE_PEAK, 4),
                "lane_utilization": vp["valu_lane_utilization"],
                "share_of_profiled_launch": round(share, 5),
                "source": vp.get("source", "profiles/valu.json"),
            },
        })
        if roof["traffic"]:
            roof["measured_hbm_frac"] = round(roof["traffic"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    else:
        roof.update({"bound": "valu (unquantified: no fresh PMC pass)",
                     "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5),
                     "note": "no PMC pass of this workload, kernel and library committed: algorithmic-bytes "
                             "equivalent against the HBM peak, not measured HBM and not the binding roof"})
    return roof


def cpu_baseline(spp):
    """The CPU side timed on this box's host cores on a bounded sample of the same workload: the
    same 1800x800 pixels, the first `spp` of the 512 samples, all hardware threads of the job's
    CPU share.  Two builds of the CPU restatement (oracle/) are timed:
      * the baseline SURVEY.md 8d defines - `value`: -O3 with the reference's AVX2 two-sibling slab
        test on an approximate reciprocal (include/simd_hit.h:121-156, include/bvh.h:109-116), i.e.
        what the reference's release binary runs (oracle/liboracle_avx2.so; its image differs from
        the scalar path's, quirk Q9, so it is timed, never compared);
      * the scalar parity partner the GPU image is bit-compared with (oracle/liboracle.so:
        -ffp-contract=off, exact 1/x) - `scalar_parity_build`.
    The reference itself cannot be built here, hence kind = "port"."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    scene = load_scene((1800, 800))
    params = scene.default_params(samples=spp)
    share = cpu_share()
    legs = {}
    for key, name in (("avx2", "liboracle_avx2.so"), ("scalar", "liboracle.so")):
        lib = O.load(name)
        t0 = time.perf_counter()
        _, st, threads = O.render(scene, params, threads=share, lib=lib)
        dt = time.perf_counter() - t0
        legs[key] = {"value": round(st.rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": int(threads),
                     "seconds": round(dt, 2), "library": "oracle/" + name}
    out = dict(legs["avx2"])
    out.update({
        "kind": "port",
        "build": "g++ -O3 -march=x86-64-v3, AVX2 two-sibling slab test on _mm256_rcp_ps (the reference's "
                 "release path); README-derived figure for the reference itself: ~74 Mrays/s on a Ryzen 7 7700",
        "sample": f"disney_spheres.json 1800x800, first {spp} of 512 spp (same pixels, same seeds), "
                  f"{legs['avx2']['cores']} threads",
        "scalar_parity_build": dict(legs["scalar"], build="g++ -O3 -march=x86-64-v3 -ffp-contract=off, scalar slab "
                                                          "test with exact 1/x: the build the GPU image is bit-compared with"),
    })
    return out


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment: start N fresh
    ranks as CHILD processes (never an exec: under rocprofv3 the GPU is initialised before this
    program starts) and relay rank 0's line.  This process makes no GPU call."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    if proc.returncode != 0 or line is None:
        raise SystemExit(proc.returncode or 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=512)
    ap.add_argument("--weak", action="store_true",
                    help="N>1: grow the image with sqrt(N) per axis (1800x800 pixels per GPU) instead of "
                         "sharding the fixed 1800x800 frame")
    ap.add_argument("--strong", action="store_true", help="(default) keep 1800x800 total")
    ap.add_argument("--cpu-spp", type=int, default=64, help="samples of the CPU baseline legs (about 15 s on 16 threads)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 path (host-staged gather), not a measurement")
    ap.add_argument("--verify", action="store_true",
                    help="N>1: rank 0 also compares the assembled frame with the single-GPU frame bit for bit")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--res", type=int, nargs=2, default=None, help="rehearsal: another frame size")
    ap.add_argument("--no-weak-leg", action="store_true", help="N>1: skip the weak-scaling figure that rides along")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)

    import torch
    import torch.distributed as dist
    from vimg_amd import hip, dist as vdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    base = tuple(args.res) if args.res else (1800, 800)
    grown = (8 * round(base[0] * math.sqrt(n) / 8), 8 * round(base[1] * math.sqrt(n) / 8))
    res = base if (n == 1 or not args.weak) else grown
    # a dedicated (non-default) stream: the kernels are launched on it and the HIP events that
    # time them are recorded on it
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    red_dev = "cuda" if args.backend == "nccl" else "cpu"

    def measure(res, steps, warmup):
        """W warm-up steps, then exactly `steps` steps of one frame of `res` between barriers;
        returns the slowest rank's wall time and what was rendered."""
        scene = load_scene(res)
        dev = hip.DeviceScene(scene)
        params = scene.default_params(samples=args.spp, tile_rank=rank, tile_world=n)
        W, H = res
        if n == 1:
            frame = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
            slab, gathered, stride = frame, None, 0
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

        for _ in range(warmup):
            step()
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                  for _ in range(steps)]
        if n > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(events[k])
        torch.cuda.synchronize()
        if n > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        dev.check()     # a frame the kernel's watchdog gave up is an error, not a measurement
        kernel_ms = sum(a.elapsed_time(b) for a, b in events) / steps
        # event counts of one step (deterministic: same seeds every step); not timed, and AFTER the timed
        # launches: the statistics build uses scratch, and rocprofv3's dispatch records show a queue's
        # scratch on every later dispatch (the timed kernel's own record: Scratch_Size 0)
        _, st = dev.render(params, out=slab, stats=True, stream=stream)
        local_pixels = st.paths // args.spp
        local_bytes = algorithmic_bytes(st, local_pixels)
        counts = torch.tensor([st.rays, st.paths, local_bytes], dtype=torch.float64, device=red_dev)
        if n > 1:
            dist.all_reduce(counts)
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=red_dev)
        if n > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return {"scene": scene, "dev": dev, "params": params, "frame": frame, "res": res,
                "elapsed": float(t[0]), "kernel_ms": kernel_ms, "kernel_ms_max": float(t[1]),
                "rays": float(counts[0]), "paths": float(counts[1]), "local_bytes": local_bytes, "local_rays": int(st.rays),
                "kernel": dev.kernel_for(params)}

    m = measure(res, args.steps, args.warmup)
    scene, dev, frame = m["scene"], m["dev"], m["frame"]
    W, H = res
    elapsed, kernel_ms, kernel_ms_max = m["elapsed"], m["kernel_ms"], m["kernel_ms_max"]
    total_rays, total_paths, local_bytes, kernel_name = m["rays"], m["paths"], m["local_bytes"], m["kernel"]
    local_rays = m["local_rays"]

    # untimed extra on rank 0 (strong scaling): the whole frame alone, for T1 and --verify
    t1_ms = None
    if n > 1 and rank == 0 and not args.weak:
        whole = scene.default_params(samples=args.spp)
        alone = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        dev.render_async(whole, alone, stream=stream)           # warm-up of the whole-frame launch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        dev.render_async(whole, alone, stream=stream)
        e1.record(stream)
        torch.cuda.synchronize()
        t1_ms = e0.elapsed_time(e1)
        if args.verify:
            assert torch.equal(alone, frame), "assembled shards differ from the single-GPU frame"
            print("verify: assembled frame is bit-identical to the single-GPU frame", file=sys.stderr)
    if n > 1:
        dist.barrier()
    # riding along for N > 1 (strong scaling is the headline): the same measurement on the frame grown
    # with sqrt(N) per axis, where every GPU keeps 1800x800 pixels of work
    weak = None
    if n > 1 and not args.weak and not args.no_weak_leg:
        try:
            mw = measure(grown, args.steps, 1)
            weak = {"resolution": list(grown), "ms_per_step": round(mw["elapsed"] / args.steps * 1e3, 3),
                    "value": round(mw["rays"] * args.steps / mw["elapsed"] / 1e6, 2), "unit": "Mrays/s",
                    "kernel": mw["kernel"], "note": "weak scaling: 1800x800 pixels per GPU"}
        except Exception as e:   # the headline must not die with the extra
            weak = {"error": repr(e)[:200]}
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rays * args.steps / elapsed / 1e6
        achieved = local_bytes / (kernel_ms * 1e-3) / 1e9
        workload = f"disney_spheres.json, mis integrator, {args.spp} spp, {W}x{H}"
        strong = n > 1 and not args.weak
        out = {
            "metric": "Mrays/sec (primary+secondary) at 512 spp",
            "value": round(value, 2),
            "unit": "Mrays/s",
            "n_gpus": n,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak" if (n > 1 and args.weak) else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "reference scene file (scenes/disney_spheres.json), no external assets",
            "config": {
                "workload": workload + ("" if n == 1 else f" tile-sharded over {n} GPUs "
                                        f"({'1800x800 pixels per GPU' if args.weak else 'fixed frame'})"),
                "integrator": "mis", "spp": args.spp, "resolution": [W, H], "depth": "unbounded",
                "bvh": "sweep SAH (host)", "sharding": f"tiles%{n}",
            },
            "mpaths_per_s": round(total_paths * args.steps / elapsed / 1e6, 3),
            "rays_per_path": round(total_rays / total_paths, 4),
            "kernel_ms": round(kernel_ms, 3),
            "kernel_ms_slowest_rank": round(kernel_ms_max, 3),
        }
        if weak is not None:
            out["weak"] = weak
        if strong and t1_ms is not None:
            out["t1_ms"] = round(t1_ms, 3)
            out["efficiency"] = round(t1_ms / (n * ms_per_step), 4)
            out["efficiency_kernels_only"] = round(t1_ms / (n * kernel_ms_max), 4)
        roof = build_roofline(kernel_ms, local_bytes, local_rays, kernel_name, workload, library_fingerprint())
        out["roofline"] = roof
        if n == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.cpu_spp)
        print(json.dumps(out), flush=True)
    if n > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
