"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Numerics bar: the kernels evaluate exactly the oracle's float/double expression tree
(-ffp-contract=off both sides, IEEE divide/sqrt, same min/max selects); the only operations
that are not bit-defined across the two are the double-precision libm/OCML transcendentals
(cos, sin, acos, atan2, log, log2, pow), which both round to < 1 ulp (double) before the
result is narrowed to float.  The tests therefore demand:
  * bit-exact integers / indices (hit flags, primitive ids, ray counts),
  * images: >= 99.9 % of pixels bit-identical and max |diff| <= 1e-5 relative on the rest at
    test sizes (a last-bit difference in a transcendental can reroute a path; it is rare),
  * unit probes: <= 2 ulp.
"""
import numpy as np
import pytest

import oracle_lib as O
import scenes

pytestmark = pytest.mark.gpu


def _dev(scene):
    from vimg_amd import hip
    return hip.DeviceScene(scene)


def _ulp_diff(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    d[np.isnan(a) & np.isnan(b)] = 0
    return d


def _compare_images(gpu, cpu, what, min_exact=0.999, tol=1e-5, k_sigma=6.0):
    """The bar of the module docstring, per pixel: at least `min_exact` of the pixels bit-identical;
    every other pixel either within tol * (largest radiance of the image) of the oracle's value
    (last-bit difference carried through), or - a path re-routed by a last-bit difference in a
    transcendental - still a draw of the same estimator: within k_sigma standard deviations of the
    oracle's per-pixel estimate, the deviation taken from the oracle image itself (pixels have
    independent RNG streams, so the spread of the 5x5 neighbourhood around a pixel estimates the spread
    of its own mean; a twentieth of the image's range is added for neighbourhoods that happen to be
    flat).  The re-routed ones are counted inside the (1 - min_exact) allowance and the image mean
    is held to tol as well."""
    from scipy.ndimage import uniform_filter
    assert gpu.shape == cpu.shape
    exact = (gpu.view(np.uint32) == cpu.view(np.uint32)).all(axis=-1)
    frac = exact.mean()
    scale = max(1e-3, float(np.abs(cpu[np.isfinite(cpu)]).max())) if np.isfinite(cpu).any() else 1.0
    diff = np.abs(gpu.astype(np.float64) - cpu.astype(np.float64))
    diff[np.isnan(gpu) & np.isnan(cpu)] = 0.0
    px_diff = diff.max(axis=-1)
    rerouted = px_diff > tol * scale
    c = np.nan_to_num(cpu.astype(np.float64), nan=0.0, posinf=scale, neginf=0.0)
    mean = uniform_filter(c, size=(5, 5, 1), mode="nearest")
    var = np.maximum(uniform_filter(c * c, size=(5, 5, 1), mode="nearest") - mean * mean, 0.0)
    sigma = np.sqrt(var).max(axis=-1) * (25.0 / 24.0) ** 0.5
    bound = k_sigma * sigma + 0.05 * scale
    print(f"{what}: bit-identical pixels {frac * 100:.4f} %, max |diff| {px_diff.max():.3e}, "
          f"re-routed pixels {int(rerouted.sum())}"
          + (f", the largest at {float((px_diff[rerouted] / np.maximum(sigma[rerouted], 1e-30)).max()):.2f} sigma" if rerouted.any() else ""))
    assert frac >= min_exact, f"{what}: only {frac * 100:.3f} % pixels bit-identical"
    assert rerouted.mean() <= 1.0 - min_exact, f"{what}: {int(rerouted.sum())} pixels beyond {tol} of the range"
    assert np.all(px_diff[rerouted] <= bound[rerouted]), f"{what}: a differing pixel is beyond {k_sigma} sigma of the oracle's estimate"
    assert np.isfinite(gpu).all() == np.isfinite(cpu).all()
    assert np.abs(gpu.mean() - cpu.mean()) <= tol * scale + 1e-3 * abs(cpu.mean())


SCENE_CASES = [
    ("disney_spheres.json", (120, 56), 8, None),
    ("glass_in_box.json", (96, 72), 8, None),
    ("cornell_box_spheres.json", (80, 80), 8, None),
    ("empty_box.json", (64, 64), 4, None),
    ("MIS_light_tests/sphere_light_small_mis.json", (64, 64), 16, None),
    ("MIS_light_tests/sphere_light_medium_mis.json", (64, 64), 16, None),
]


@pytest.mark.parametrize("name,res,spp,depth", SCENE_CASES)
def test_image_matches_oracle_json_scenes(name, res, spp, depth):
    s = scenes.json_scene(name, res=res)
    p = s.default_params(samples=spp) if depth is None else s.default_params(samples=spp, depth=depth)
    cpu, cst, _ = O.render(s, p)
    d = _dev(s)
    gpu, gst = d.render_to_host(p)
    _compare_images(gpu, cpu, name)
    # event counts are integers: every query of every path agrees except on re-routed paths
    assert gst.paths == cst.paths
    assert abs(gst.rays - cst.rays) <= max(4, 1e-4 * cst.rays)
    assert gst.nan_samples == cst.nan_samples


def test_odyssey_quad_light_floor_on_the_gpu():
    """The reference's quad-light known-answer scene without its (absent) monolith: GPU = oracle bit
    for bit at the file's own settings, and the GPU image against the reference's picture on the
    pixels the monolith cannot touch (tests/test_oracle_pins.py:check_against_odyssey_reference)."""
    from test_oracle_pins import check_against_odyssey_reference
    s = scenes.odyssey_without_monolith()
    p = s.default_params()
    cpu, cst, _ = O.render(s, p)
    gpu, gst = _dev(s).render_to_host(p)
    _compare_images(gpu, cpu, "odyssey without monolith")
    assert gst.paths == cst.paths and abs(gst.rays - cst.rays) <= max(4, 1e-4 * cst.rays)
    check_against_odyssey_reference(gpu, s, "GPU")


def _dev_opts(scene, **opts):
    from vimg_amd import hip
    return hip.DeviceScene(scene, **opts)


# scheduler configurations (VimgHipOptions) of the PRODUCT library: every one must give the lane-bound
# kernel's bits.  (The schedulers of rounds 1 and 2 - pool, pool4, pool4g, stage - live in the
# development build and are cross-checked there: test_dev_build_schedulers_give_the_same_bits.)
SCHEDULES = {
    "lane": dict(scheduler="lane"),
    # the CU-wide scheduler as the policy configures it
    "cu": dict(scheduler="cu"),
    # every pixel's samples cut into 5 segments that travel through per-pixel records in global memory;
    # at test sizes far more slots are in flight than there are pixels, so slots constantly draw
    # segments whose predecessor is still running (the waiting path)
    "cu/5": dict(scheduler="cu", pool_segments=5),
    # few slots per compute unit: several generations of pixels per slot, rings that wrap often
    "cu/few": dict(scheduler="cu", pool_slots=64, pool_segments=2),
    "cu/tiny": dict(scheduler="cu", pool_slots=8, pool_segments=3),
    # every wave walks and shades / one wave walks / fifteen walk
    "cu/w16": dict(scheduler="cu", cu_walkers=16),
    "cu/w1": dict(scheduler="cu", cu_walkers=1, pool_segments=2),
    "cu/w15": dict(scheduler="cu", cu_walkers=15, cu_flex=0),
    # batches and refills as eager as they get; as patient as they get
    "cu/eager": dict(scheduler="cu", pool_refill=1, pool_starve=1, cu_patience=0, cu_join=1, pool_vbatch=8),
    "cu/patient": dict(scheduler="cu", pool_refill=48, pool_starve=64, cu_patience=64, cu_join=64, cu_sleep=32),
    # rays queued as soon as they are known (three-way join of shadow ray, path ray and vertex stage) / at the end
    # of the vertex stage; batches never split over the halves of the wave
    "cu/early": dict(scheduler="cu", cu_flex=33, pool_segments=2),
    "cu/late": dict(scheduler="cu", cu_flex=1),
    "cu/nosplit": dict(scheduler="cu", cu_flex=49),
    # one queue of shading vertices instead of one per material class; two
    "cu/1class": dict(scheduler="cu", pool_classes=1),
    "cu/2class": dict(scheduler="cu", pool_classes=2, pool_segments=2),
    # only the first entries of a lane's traversal stack in LDS, the rest in global memory: one entry /
    # three, so that nearly every push and pop takes that path (the build for trees beyond LDS)
    "cu/stack1": dict(scheduler="cu", lds_stack=1),
    "cu/stack3": dict(scheduler="cu", lds_stack=3, pool_segments=2),
    # the top of the tree outside LDS too (one kilobyte of node cache), no leaf copy in LDS
    "cu/nolds": dict(scheduler="cu", lds_budget_kb=1, lds_leaf=0),
}
# the same for the development build (tests/dev_schedulers.py)
DEV_SCHEDULES = {
    "pool": dict(scheduler="pool"),
    "pool/5": dict(scheduler="pool", pool_segments=5),
    "pool/64": dict(scheduler="pool", pool_segments=64),
    "pool4": dict(scheduler="pool4"),
    "pool4/5": dict(scheduler="pool4", pool_segments=5),
    "pool4/4": dict(scheduler="pool4", waves_per_simd=4, pool_segments=3),
    "pool4g": dict(scheduler="pool4g"),
    "pool4g/5": dict(scheduler="pool4g", pool_segments=5),
    "pool4g/few": dict(scheduler="pool4g", pool_slots=24, pool_segments=2),
    "pool4/stack1": dict(scheduler="pool4", lds_stack=1),
    "pool4/stack3": dict(scheduler="pool4", lds_stack=3, pool_segments=2),
    "stage": dict(scheduler="stage"),
    "stage/few": dict(scheduler="stage", stage_slots=300, stage_seg_len=1),
    "stage/whole": dict(scheduler="stage", stage_slots=1000, stage_seg_len=1 << 20, stage_walk_quota=128),
}
KERNEL_OF = {"lane": "render_kernel", "cu": "render_cu_kernel", "pool": "render_pool_kernel", "pool4": "render_pool4_kernel",
             "pool4g": "render_pool4_kernel", "stage": "render_stage_kernel"}


def scheduler_scene(scene_name):
    if scene_name == "feature":
        s = scenes.feature_scene(res=(72, 48), envmap=True, lens=True)
        return s, s.default_params(samples=6, depth=7)
    s = scenes.json_scene(scene_name, res=(136, 72))
    return s, s.default_params(samples=12)


def check_schedules_against_lane(s, p, schedules, what, twice=True):
    """Every configuration of `schedules` renders the lane-bound kernel's bits and event counts
    (a second launch on the same records and trace_pixel included when `twice`)."""
    ref_dev = _dev_opts(s, scheduler="lane")
    ref, rst = ref_dev.render_to_host(p)
    ref_px = ref_dev.trace_pixel(p, 17, 23) if twice else None
    for name, opts in schedules.items():
        d = _dev_opts(s, **opts)
        assert d.kernel.startswith(KERNEL_OF[opts["scheduler"]]), (name, d.kernel)
        assert ("group" in d.kernel) == (opts["scheduler"] == "pool4g"), (name, d.kernel)
        img, st = d.render_to_host(p)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (what, name, d.kernel)
        assert st.as_dict() == rst.as_dict(), (what, name)
        # a launch without statistics runs the build without the diagnostics (render_cu_kernel<..., DIAG = false>,
        # the one frames are timed on): same bits
        plain = d.render_to_host(p, stats=False)
        assert np.array_equal(plain.view(np.uint32), ref.view(np.uint32)), (what, name, "launch without statistics")
        if twice:
            again, _ = d.render_to_host(p)            # a second launch on the same records
            assert np.array_equal(img.view(np.uint32), again.view(np.uint32)), (what, name)
            px = d.trace_pixel(p, 17, 23)
            assert np.array_equal(np.asarray(px).view(np.uint32), np.asarray(ref_px).view(np.uint32)), (what, name)
        d.close()
    return ref, rst


FEATURE_CASES = {
    "config3": lambda: (scenes.config3_scene(res=(96, 72), env=(128, 64)), dict(samples=8, depth=12)),
    "config4": lambda: (scenes.config4_scene(res=(96, 54), n_lat=48, env=(128, 64)), dict(samples=8, depth=12)),
    "config5": lambda: (scenes.config5_scene(res=(96, 54), n=64, tex=64), dict(samples=8, depth=12)),
    "big_mesh": lambda: (scenes.big_mesh_scene(res=(96, 64)), dict(samples=6)),
    "cornell material": lambda: (scenes.json_scene("cornell_box_spheres.json", res=(80, 80)),
                                 dict(samples=8, integrator="material", depth=16)),
    "feature material": lambda: (scenes.feature_scene(res=(72, 48)), dict(samples=6, integrator="material", depth=8)),
    "glass s_normal": lambda: (scenes.json_scene("glass_in_box.json", res=(96, 72)), dict(samples=4, integrator="s_normal")),
    "disney g_normal": lambda: (scenes.json_scene("disney_spheres.json", res=(120, 56)), dict(samples=4, integrator="g_normal")),
    "sphere lights": lambda: (scenes.json_scene("MIS_light_tests/sphere_light_medium_mis.json", res=(64, 64)), dict(samples=16)),
    "const background light": lambda: (scenes.feature_scene(res=(72, 48), envmap=False, lens=False), dict(samples=6, depth=7)),
}


@pytest.mark.parametrize("scene_name", ["disney_spheres.json", "glass_in_box.json", "feature"])
def test_all_schedulers_give_the_same_bits(scene_name):
    """render_kernel (one path per lane) and render_cu_kernel (paths pooled per compute unit, walking
    and shading waves coupled by rings in LDS) in every configuration of SCHEDULES are schedules of
    the same per-path arithmetic: identical images, identical event counts; trace_pixel and repeated
    launches on the same scratch included; and the image is the oracle's."""
    s, p = scheduler_scene(scene_name)
    ref, rst = check_schedules_against_lane(s, p, SCHEDULES, scene_name)
    cpu, cst, _ = O.render(s, p)
    _compare_images(ref, cpu, scene_name)
    assert rst.paths == cst.paths


@pytest.mark.parametrize("case", list(FEATURE_CASES))
def test_cu_scheduler_on_every_feature(case):
    """The CU scheduler in its corner configurations on every feature the path has - image textures
    with mips, normal and RG maps, env-map and constant-background lights, sphere lights, thin lens,
    deep trees (the DEEP builds), the material and normal integrators: the lane-bound kernel's bits
    and event counts."""
    s, kw = FEATURE_CASES[case]()
    p = s.default_params(**kw)
    pick = ("cu", "cu/5", "cu/few", "cu/tiny", "cu/w16", "cu/w1", "cu/eager", "cu/early", "cu/late", "cu/1class", "cu/stack1", "cu/stack3",
            "cu/nolds")
    check_schedules_against_lane(s, p, {k: SCHEDULES[k] for k in pick}, case, twice=False)


def test_dev_build_schedulers_give_the_same_bits():
    """The schedulers of rounds 1 and 2 (render_pool_kernel, render_pool4_kernel per wave and per
    workgroup, render_stage_kernel) are kept as reference implementations in the development build
    of the library (make dev): ONE child process loads that build (VIMG_HIP_LIB) and checks every
    configuration of DEV_SCHEDULES against the lane-bound kernel and the CU scheduler on the three
    scheduler scenes and the ten feature cases (tests/dev_schedulers.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev = os.path.join(root, "v-img_amd", "lib", "dev", "libvimg_hip.so")
    assert os.path.exists(dev), "make dev"
    env = dict(os.environ, VIMG_HIP_LIB=dev)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "dev_schedulers.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "DEV_SCHEDULERS OK" in r.stdout


def test_product_library_refuses_the_retired_schedulers():
    from vimg_amd import hip
    s = scenes.json_scene("disney_spheres.json", res=(32, 16))
    for name in ("pool", "pool4", "pool4g", "stage"):
        with pytest.raises(hip.HipError, match="development build"):
            _dev_opts(s, scheduler=name)


@pytest.mark.parametrize("integrator", ["s_normal", "g_normal"])
def test_normal_integrators_bit_exact(integrator):
    # BASELINE config 1: glass_in_box with the 'normal' integrator (no transcendental on the path)
    s = scenes.json_scene("glass_in_box.json", res=(160, 120))
    p = s.default_params(integrator=integrator, samples=8)
    cpu, cst, _ = O.render(s, p)
    gpu, gst = _dev(s).render_to_host(p)
    assert np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32))
    assert gst.closest_rays == cst.closest_rays == gst.paths


@pytest.mark.parametrize("envmap,lens", [(True, True), (False, False)])
def test_feature_scene_matches_oracle(envmap, lens):
    s = scenes.feature_scene(res=(96, 64), envmap=envmap, lens=lens)
    p = s.default_params(samples=8, depth=10)
    cpu, cst, _ = O.render(s, p)
    gpu, gst = _dev(s).render_to_host(p)
    _compare_images(gpu, cpu, f"feature(env={envmap},lens={lens})", min_exact=0.995)
    assert abs(gst.rays - cst.rays) <= max(16, 1e-3 * cst.rays)


def test_non_square_rg_map_reproduces_the_references_indexing():
    """Quirk Q6: TextureRG::get_at_uv indexes the +x neighbours with "* height"
    (include/texture/texture_RG.h:47,52).  On a map wider than tall that is a wrong but in-bounds
    texel: the GPU reproduces it (= the oracle, which restates the line), and the picture differs
    from the one a "* width" reading gives only through that texel - checked by rendering the same
    scene with the map's rows duplicated to a square, where the two indexings coincide on x0 / y0 but
    not on x1 / y1.  A map taller than wide makes the reference read beyond its vector: refused."""
    from vimg_amd import hip
    s = scenes.feature_scene(res=(96, 64), rg_shape=(32, 8))
    p = s.default_params(samples=8, depth=8)
    cpu, cst, _ = O.render(s, p)
    for sched in ("lane", "cu"):
        gpu, gst = _dev_opts(s, scheduler=sched).render_to_host(p)
        _compare_images(gpu, cpu, f"wide RG map ({sched})", min_exact=0.995)
        assert gst.paths == cst.paths
    square, _, _ = O.render(scenes.feature_scene(res=(96, 64), rg_shape=(8, 8)), p)
    assert not np.array_equal(square, cpu)          # the map matters to the picture
    with pytest.raises(hip.HipError, match="taller than wide"):
        _dev(scenes.feature_scene(res=(32, 24), rg_shape=(8, 32)))


def test_big_mesh_scene_full_stats():
    s = scenes.big_mesh_scene(res=(128, 96))
    p = s.default_params(samples=4, depth=8)
    cpu, cst, _ = O.render(s, p)
    gpu, gst = _dev(s).render_to_host(p)
    _compare_images(gpu, cpu, "big mesh")
    for k in ("closest_rays", "shadow_rays", "internal_visits", "leaf_visits", "prim_tests"):
        a, b = getattr(gst, k), getattr(cst, k)
        assert abs(a - b) <= max(64, 1e-3 * b), (k, a, b)


def test_binned_bvh_and_ragged_resolution():
    from vimg_amd import abi
    s = scenes.json_scene("disney_spheres.json", res=(61, 37), bvh=abi.BVH_BINNED)   # not /8
    p = s.default_params(samples=4)
    cpu, _, _ = O.render(s, p)
    gpu, _ = _dev(s).render_to_host(p)
    _compare_images(gpu, cpu, "ragged 61x37 binned")


def test_trace_pixel_matches_render_and_oracle():
    s = scenes.json_scene("disney_spheres.json", res=(120, 56))
    p = s.default_params(samples=8)
    d = _dev(s)
    img, _ = d.render_to_host(p)
    for (x, y) in [(0, 0), (60, 28), (119, 55), (33, 10)]:
        px = d.trace_pixel(p, x, y)
        assert np.array_equal(px.view(np.uint32), img[56 - 1 - y, x].view(np.uint32))
        ref = O.trace_pixel(s, p, x, y)
        assert np.allclose(px, ref, rtol=1e-5, atol=1e-6)


def test_probes_camera_hits_occlusion():
    s = scenes.feature_scene(res=(96, 64))
    d = _dev(s)
    rng = np.random.default_rng(1)
    n = 4096
    cam_in = np.stack([rng.uniform(0, 96, n), rng.uniform(0, 64, n), rng.random(n), rng.random(n)],
                      1).astype(np.float32)
    cg, cc = d.probe(O.PROBE_CAMERA_RAY, cam_in), O.probe(s, O.PROBE_CAMERA_RAY, cam_in)
    assert _ulp_diff(cg, cc).max() <= 2
    rays = cc[:, :6]
    hg, hc = d.probe(O.PROBE_CLOSEST_HIT, rays), O.probe(s, O.PROBE_CLOSEST_HIT, rays)
    assert np.array_equal(hg[:, 0], hc[:, 0])            # hit / miss
    assert np.array_equal(hg[:, 2:4], hc[:, 2:4])        # primitive, material ids
    both = hc[:, 0] == 1
    assert both.sum() > n // 2
    assert _ulp_diff(hg[both][:, 1], hc[both][:, 1]).max() == 0          # t: no transcendental
    assert _ulp_diff(hg[both][:, 4:13], hc[both][:, 4:13]).max() == 0    # p, n_s, n_g
    assert _ulp_diff(hg[both][:, 13:26], hc[both][:, 13:26]).max() <= 4  # uv (acos/atan2), frame
    occ_in = np.concatenate([hc[both][:, 4:7], -rays[both][:, 3:6],
                             rng.uniform(0.1, 6, (both.sum(), 1)).astype(np.float32)], 1)
    og, oc = d.probe(O.PROBE_OCCLUDED, occ_in), O.probe(s, O.PROBE_OCCLUDED, occ_in)
    assert np.array_equal(og, oc)


def test_probes_bsdf_and_lights():
    s = scenes.feature_scene(res=(96, 64))
    d = _dev(s)
    rng = np.random.default_rng(2)
    n = 4096
    cam_in = np.stack([rng.uniform(0, 96, n), rng.uniform(0, 64, n), rng.random(n), rng.random(n)],
                      1).astype(np.float32)
    rays = O.probe(s, O.PROBE_CAMERA_RAY, cam_in)[:, :6]
    wo = rng.normal(size=(n, 3)).astype(np.float32)
    wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    ev_in = np.concatenate([rays, wo, rng.uniform(0, 0.02, (n, 2)).astype(np.float32),
                            (rng.random((n, 1)) < 0.5).astype(np.float32)], 1)
    eg, ec = d.probe(O.PROBE_BSDF_EVAL, ev_in), O.probe(s, O.PROBE_BSDF_EVAL, ev_in)
    assert np.array_equal(eg[:, 0], ec[:, 0])
    assert np.allclose(eg, ec, rtol=2e-5, atol=1e-7, equal_nan=True)
    sm_in = np.concatenate([rays, rng.integers(0, 1 << 20, (n, 1)).astype(np.float32),
                            (rng.random((n, 1)) < 0.5).astype(np.float32)], 1)
    sg, sc = d.probe(O.PROBE_BSDF_SAMPLE, sm_in), O.probe(s, O.PROBE_BSDF_SAMPLE, sm_in)
    assert np.array_equal(sg[:, :2], sc[:, :2])
    assert np.allclose(sg, sc, rtol=2e-5, atol=1e-6)
    li_in = np.concatenate([rng.uniform(-2, 2, (n, 3)).astype(np.float32),
                            rng.integers(0, 1 << 20, (n, 1)).astype(np.float32)], 1)
    lg, lc = d.probe(O.PROBE_LIGHT_SAMPLE, li_in), O.probe(s, O.PROBE_LIGHT_SAMPLE, li_in)
    assert np.allclose(lg, lc, rtol=2e-5, atol=1e-7)
    bg_in = np.concatenate([wo, rng.uniform(0, 0.05, (n, 2)).astype(np.float32)], 1)
    bg, bc = d.probe(O.PROBE_BACKGROUND, bg_in), O.probe(s, O.PROBE_BACKGROUND, bg_in)
    assert np.allclose(bg, bc, rtol=2e-5, atol=1e-7)


def test_sincos_is_the_two_calls_it_replaces():
    """The warps take sin and cos of an angle from ONE sincos (device_math.h: D_sincos; the compiler does not
    merge the two calls, and they are a tenth of a Lambertian vertex).  The pictures only stay the
    reference's if that is the same pair of doubles the separate calls return: 2^24 arguments - the
    angles 2 pi u the warps draw, the whole float range around them, negative and large ones, the
    special values - give the same bits, doubles included; and the floats are glibc's."""
    s = scenes.json_scene("disney_spheres.json", res=(16, 8))
    d = _dev(s)
    rng = np.random.default_rng(7)
    n = 1 << 22
    x = np.concatenate([
        (np.float32(2.0) * np.float32(np.pi) * rng.random(n, dtype=np.float32)).astype(np.float32),
        rng.integers(0, np.float32(6.2831855).view(np.uint32), n, dtype=np.uint32).view(np.float32),
        rng.uniform(-1e6, 1e6, n).astype(np.float32),
        (rng.normal(size=n) * 10.0 ** rng.uniform(-30, 8, n)).astype(np.float32),
    ])
    x[:8] = np.array([0.0, -0.0, np.pi, -np.pi, 2 * np.pi, 1e-40, 3.4e38, -3.4e38], dtype=np.float32)
    out = d.probe(8, x.reshape(-1, 1))
    assert np.array_equal(out[:, 0].view(np.uint32), out[:, 2].view(np.uint32))
    assert np.array_equal(out[:, 1].view(np.uint32), out[:, 3].view(np.uint32))
    assert np.all(out[:, 4] == 1.0)
    # ... and they are the host's: float(cos(double(x))) differs from glibc's on no more than a few arguments
    # in a million (double results that differ in their last bit AND straddle a float rounding boundary)
    ref_c, ref_s = np.cos(x.astype(np.float64)).astype(np.float32), np.sin(x.astype(np.float64)).astype(np.float32)
    small = np.abs(x) < 1e5
    assert np.mean(out[small, 2] != ref_c[small]) < 1e-5 and np.mean(out[small, 3] != ref_s[small]) < 1e-5


def test_tile_shards_reassemble_to_the_single_gpu_image():
    # image is independent of the number of shards by construction (per-pixel seeds)
    import torch
    from vimg_amd import dist as vdist
    s = scenes.json_scene("disney_spheres.json", res=(123, 61))
    d = _dev(s)
    p1 = s.default_params(samples=4)
    full, _ = d.render(p1)
    for world in (2, 3, 8):
        stride = vdist.shard_stride_pixels(123, 61, world)
        gathered = torch.zeros((world, stride, 3), dtype=torch.float32, device="cuda")
        paths = 0
        for r in range(world):
            pr = s.default_params(samples=4, tile_rank=r, tile_world=world)
            _, st = d.render(pr, out=gathered[r])
            paths += st.paths
        assert paths == 123 * 61 * 4
        img = d.assemble_shards(gathered, world, stride)
        assert torch.equal(img, full)
        host = vdist.assemble_numpy(gathered.cpu().numpy(), 123, 61, world)
        assert np.array_equal(host, full.cpu().numpy())


@pytest.mark.parametrize("name", ["config4", "config5"])
def test_config4_and_5_standins_as_2_4_8_shards(name):
    """BASELINE configs[3] and [4] in the form they are stated - tile-sharded over 2 / 4 / 8 GPUs -
    on their stand-ins (deep trees, textures, normal / RG maps, HDRI + thin lens): every shard of
    every split rendered on this GPU (tile t -> rank t % N, include/integrators.h:57-65,101) with
    the scheduler the policy picks for it, and with a small pool asked for by name; the shards
    reassemble, on the device, to the bits of the whole frame.  Stats add up exactly."""
    import torch
    from vimg_amd import dist as vdist
    s = {"config4": lambda: scenes.config4_scene(res=(200, 112), n_lat=64, env=(128, 64)),
         "config5": lambda: scenes.config5_scene(res=(200, 112), n=80, tex=64)}[name]()
    w, h = s.resolution
    kw = dict(samples=6, depth=10)
    d = _dev(s)
    full, st_full = d.render(s.default_params(**kw))
    pooled = _dev_opts(s, scheduler="cu", pool_slots=40, pool_segments=2)
    for world in (2, 4, 8):
        for dev in (d, pooled):
            stride = vdist.shard_stride_pixels(w, h, world)
            gathered = torch.zeros((world, stride, 3), dtype=torch.float32, device="cuda")
            tot = {}
            for r in range(world):
                pr = s.default_params(tile_rank=r, tile_world=world, **kw)
                assert dev.shard_pixels(pr) == 64 * len(vdist.shard_tiles(w, h, r, world))
                _, st = dev.render(pr, out=gathered[r])
                for k, v in st.as_dict().items():
                    tot[k] = tot.get(k, 0) + v
            img = dev.assemble_shards(gathered, world, stride)
            assert torch.equal(img, full), (name, world, dev.kernel)
            assert tot == st_full.as_dict(), (name, world)


@pytest.mark.parametrize("name", ["config3", "config4", "config5"])
def test_config_standins_at_full_size_properties(name):
    """The stand-ins of BASELINE configs 3-5 at the sizes SURVEY.md 8d gives them (1366x1024 Disney
    array under a 1024x512 env map; 1366x768 with 635 K / 1.0 M triangles), where the oracle
    cannot render the frame in test time: size-independent properties - finite, deterministic,
    the policy's scheduler equals the lane-bound kernel bit for bit, plausible rays per path,
    an eighth of the frame as a shard equals the same pixels of the frame - and three
    single-pixel traces against the oracle at the full sample count of the test."""
    import torch
    from vimg_amd import dist as vdist
    s = {"config3": scenes.config3_scene, "config4": scenes.config4_scene,
         "config5": lambda: scenes.config5_scene(n=700)}[name]()
    w, h = s.resolution
    assert (w, h) == ((1366, 1024) if name == "config3" else (1366, 768))
    p = s.default_params(samples=4, depth=16)
    d = _dev(s)
    assert d.kernel.startswith("render_cu_kernel")
    img, st = d.render_to_host(p)
    assert st.paths == w * h * 4 and st.nan_samples == 0
    assert np.isfinite(img).all() and img.min() >= 0
    again, st2 = d.render_to_host(p)
    assert np.array_equal(img.view(np.uint32), again.view(np.uint32)) and st.as_dict() == st2.as_dict()
    lane, st_lane = _dev_opts(s, scheduler="lane").render_to_host(p)
    assert np.array_equal(img.view(np.uint32), lane.view(np.uint32)) and st.as_dict() == st_lane.as_dict()
    assert 1.5 < st.rays / st.paths < 12.0
    p8 = s.default_params(samples=4, depth=16, tile_rank=5, tile_world=8)
    slab, _ = d.render(p8)
    stride = vdist.shard_stride_pixels(w, h, 8)
    gathered = np.zeros((8, stride, 3), dtype=np.float32)
    gathered[5, :slab.shape[0]] = slab.cpu().numpy()
    mine = vdist.assemble_numpy(gathered, w, h, 8)
    mask = np.zeros((8, stride, 3), dtype=np.float32)
    mask[5, :slab.shape[0]] = 1
    own = vdist.assemble_numpy(mask, w, h, 8)[..., 0] > 0
    assert np.array_equal(mine[own].view(np.uint32), img[own].view(np.uint32))
    for (x, y) in ((w // 2, h // 2), (w // 3, (2 * h) // 3), (w - 7, 11)):
        ref = O.trace_pixel(s, p, x, y)
        got = d.trace_pixel(p, x, y)
        assert np.allclose(got, ref, rtol=2e-5, atol=1e-6), (name, x, y, got, ref)
        assert np.allclose(img[h - 1 - y, x], ref, rtol=2e-5, atol=1e-6)


def test_default_stream_launches_are_ordered_with_torch():
    """`stream=None` means torch's current stream, not the library's private non-blocking one: a
    shard rendered into a slab the wrapper allocates (torch.zeros on the current stream, then the
    kernel) must come out complete every time - with an unordered launch the zero fill can land
    after the kernel's writes - and a torch consumer enqueued right behind the render sees the
    finished pixels."""
    import torch
    from vimg_amd import dist as vdist
    s = scenes.json_scene("disney_spheres.json", res=(264, 120))
    d = _dev(s)
    w, h = s.resolution
    full, _ = d.render(s.default_params(samples=4))
    p = s.default_params(samples=4, tile_rank=1, tile_world=3)
    stride = vdist.shard_stride_pixels(w, h, 3)
    mask = np.zeros((3, stride, 3), dtype=np.float32)
    mask[1, :d.shard_pixels(p)] = 1
    own = vdist.assemble_numpy(mask, w, h, 3)[..., 0] > 0
    want = full.cpu().numpy()[own]
    for _ in range(20):
        junk = torch.full((stride, 3), 7.0, device="cuda")      # keeps the current stream busy before the launch
        slab, _ = d.render(p)                                   # out=None: allocated and zero-filled by the wrapper
        total = slab.sum()                                      # consumer on the current stream, no host sync in between
        gathered = np.zeros((3, stride, 3), dtype=np.float32)
        gathered[1, :slab.shape[0]] = slab.cpu().numpy()
        got = vdist.assemble_numpy(gathered, w, h, 3)[own]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert abs(float(total) - float(want.sum())) <= 1e-3 * abs(float(want.sum()))
        del junk


def test_full_size_properties_disney_spheres():
    """BASELINE config 2 at full resolution, few samples: size-independent properties."""
    s = scenes.json_scene("disney_spheres.json")
    p = s.default_params(samples=4)
    d = _dev(s)
    img, st = d.render_to_host(p)
    assert img.shape == (800, 1800, 3)
    assert st.paths == 1800 * 800 * 4 and st.nan_samples == 0
    assert np.isfinite(img).all() and img.min() >= 0
    again, st2 = d.render_to_host(p)
    assert np.array_equal(img, again) and st.rays == st2.rays        # deterministic
    assert 5.5 < st.rays / st.paths < 6.0                              # SURVEY: 5.71 rays/path
    # first hit on the light is exact: pixels looking at the emitter see exactly its radiance
    assert np.any(np.all(img == 2.0, axis=-1))
    # a single-pixel trace agrees with the oracle at full size
    ref = O.trace_pixel(s, p, 900, 400)
    assert np.allclose(d.trace_pixel(p, 900, 400), ref, rtol=1e-5, atol=1e-6)


def test_full_size_schedulers_and_segments_agree():
    """BASELINE config 2 at full resolution: the policy launches the CU scheduler with the samples of
    a pixel cut into segments; the same with seven segments, with whole pixels, with a small pool,
    the lane-bound kernel and a thin shard all give the same bits.  Size-independent property: the
    image does not depend on scheduler, segment count, pool size or shard count."""
    s = scenes.json_scene("disney_spheres.json")
    p = s.default_params(samples=16)
    d = _dev(s)
    assert d.kernel.startswith("render_cu_kernel")
    auto, st_auto = d.render_to_host(p)
    seg, st_seg = _dev_opts(s, scheduler="cu", pool_segments=7).render_to_host(p)   # 16 samples in segments of 3 (+1)
    whole, st_whole = _dev_opts(s, scheduler="cu", pool_segments=1).render_to_host(p)
    small, st_small = _dev_opts(s, scheduler="cu", pool_slots=256, cu_walkers=12).render_to_host(p)
    lane_dev = _dev_opts(s, scheduler="lane")
    assert lane_dev.kernel.startswith("render_kernel")
    lane, st_lane = lane_dev.render_to_host(p)
    for img in (auto, seg, whole, small):
        assert np.array_equal(img.view(np.uint32), lane.view(np.uint32))
    assert st_auto.as_dict() == st_lane.as_dict() == st_seg.as_dict() == st_whole.as_dict() == st_small.as_dict()
    # an eighth of the frame (what one GPU of eight renders of the fixed frame)
    import torch
    from vimg_amd import dist as vdist
    p8 = s.default_params(samples=16, tile_rank=3, tile_world=8)
    slab, _ = d.render(p8)
    stride = vdist.shard_stride_pixels(1800, 800, 8)
    gathered = np.zeros((8, stride, 3), dtype=np.float32)
    gathered[3, :slab.shape[0]] = slab.cpu().numpy()
    mine = vdist.assemble_numpy(gathered, 1800, 800, 8)
    mask = np.zeros((8, stride, 3), dtype=np.float32)
    mask[3, :slab.shape[0]] = 1
    own = vdist.assemble_numpy(mask, 1800, 800, 8)[..., 0] > 0
    assert own.sum() == slab.shape[0]
    assert np.array_equal(mine[own].view(np.uint32), lane[own].view(np.uint32))


def test_error_paths():
    from vimg_amd import hip
    s = scenes.json_scene("disney_spheres.json", res=(32, 16))
    d = _dev(s)
    with pytest.raises(hip.HipError):
        d.render_to_host(s.default_params(integrator="material", depth=0))
    with pytest.raises(hip.HipError):
        d.render_to_host(s.default_params(samples=0))
    with pytest.raises(hip.HipError):
        d.render_to_host(s.default_params(tile_rank=2, tile_world=2))
    with pytest.raises(hip.HipError):
        d.trace_pixel(s.default_params(), 32, 0)


@pytest.mark.parametrize("name", ["config3", "config4", "config5"])
def test_config_standins_match_oracle(name):
    """Synthetic stand-ins of BASELINE configs 3-5 (their assets are not in the reference tree):
    Disney array under an importance-sampled env map; displaced meshes + normal map + HDRI +
    thin lens; brick field + height field with mip-mapped textures, normal maps, RG map."""
    s = {"config3": lambda: scenes.config3_scene(res=(96, 72), env=(128, 64)),
         "config4": lambda: scenes.config4_scene(res=(96, 54), n_lat=48, env=(128, 64)),
         "config5": lambda: scenes.config5_scene(res=(96, 54), n=64, tex=64)}[name]()
    p = s.default_params(samples=8, depth=12)
    cpu, cst, _ = O.render(s, p)
    gpu, gst = _dev(s).render_to_host(p)
    _compare_images(gpu, cpu, name, min_exact=0.99)
    assert abs(gst.rays - cst.rays) <= max(16, 2e-3 * cst.rays)
    assert gst.nan_samples == cst.nan_samples


def test_config2_full_render_against_the_references_own_picture():
    """End to end on BASELINE config 2: 1800x800, 512 spp on the GPU, AgX + sRGB + 8-bit (host
    post chain), against the reference author's render of the same scene and settings
    (renders/disney_spheres_agx_512.png, stored as 4x4 block means).  Both are Monte-Carlo
    estimates with different rounding lotteries (builds of the reference itself span
    0.9 % in radiance), so the comparison is statistical."""
    import os
    import vimg_amd
    s = scenes.json_scene("disney_spheres.json")
    p = s.default_params()
    assert p.samples == 512
    img, st = _dev(s).render_to_host(p)
    assert st.nan_samples == 0
    ours = vimg_amd.tonemap_to_rgb8(img, 1).astype(np.float32)
    ours = ours.reshape(200, 4, 450, 4, 3).mean(axis=(1, 3))
    ref = np.load(os.path.join(scenes.SCENES, "..", "renders",
                               "disney_spheres_agx_512_ds4.npy")).astype(np.float32)
    diff = np.abs(ours - ref)
    print("mean 8-bit level ours", ours.mean(axis=(0, 1)), "ref", ref.mean(axis=(0, 1)),
          "mean |diff|", diff.mean(), "p99", np.percentile(diff, 99))
    assert np.allclose(ours.mean(axis=(0, 1)), ref.mean(axis=(0, 1)), rtol=0.012)
    assert diff.mean() < 2.0             # of 255 levels
    assert np.percentile(diff, 99) < 8.0


def test_cornell_spheres_against_the_references_own_pictures_on_the_gpu():
    """The reference-held triplet renders/sphere_{mis,mat,ref}.png (cornell_box_spheres, 800x800):
    the HIP path with the mis integrator against sphere_mis and sphere_ref, with the material
    integrator against sphere_mat and sphere_ref (what the bounds mean: test_oracle_pins)."""
    from test_oracle_pins import check_against_sphere_triplet
    s = scenes.json_scene("cornell_box_spheres.json")
    d = _dev(s)
    mis, st = d.render_to_host(s.default_params(samples=100))           # the scene file's own 100 spp
    assert st.nan_samples == 0
    check_against_sphere_triplet(mis, "mis", "HIP mis 100 spp")
    r_mis = check_against_sphere_triplet(mis, "ref", "HIP mis 100 spp")
    mat, st = d.render_to_host(s.default_params(samples=256, integrator="material"))
    assert st.nan_samples == 0 and st.shadow_rays == 0
    # BSDF sampling alone is a high-variance estimator here and clamp + sRGB bias a noisy image dark
    # by an amount that depends on the sample count (oracle, same scene: 16 spp 0.52, 100 spp 0.90,
    # 256 spp 1.0 of the converged picture's level; sphere_mat itself sits 6 % under sphere_ref), so
    # the bounds are wide: same picture, level between the author's two
    check_against_sphere_triplet(mat, "mat", "HIP material 256 spp", level=(0.98, 1.13), min_corr=0.98, max_mad=8.0)
    r_mat = check_against_sphere_triplet(mat, "ref", "HIP material 256 spp", level=(0.93, 1.06), min_corr=0.98, max_mad=8.0)
    assert np.all(r_mat < r_mis + 0.01)


@pytest.mark.parametrize("scene_name", ["cornell", "glass_in_box", "feature"])
def test_material_integrator_matches_oracle(scene_name):
    """material_integrator (reference src/integrators/mat_integrator.cpp): BSDF sampling only,
    eval_div_pdf forms, roulette break returns black, Dielectric transmits here (unlike mis)."""
    s = {"cornell": lambda: scenes.json_scene("cornell_box_spheres.json", res=(80, 80)),
         "glass_in_box": lambda: scenes.json_scene("glass_in_box.json", res=(96, 72)),
         "feature": lambda: scenes.feature_scene(res=(96, 64))}[scene_name]()
    p = s.default_params(integrator="material", samples=16, depth=24)
    cpu, cst, _ = O.render(s, p)
    gpu, gst = _dev(s).render_to_host(p)
    _compare_images(gpu, cpu, "material " + scene_name, min_exact=0.995)
    assert gst.shadow_rays == 0 and cst.shadow_rays == 0
    assert abs(gst.rays - cst.rays) <= max(8, 1e-3 * cst.rays)


@pytest.mark.parametrize("tonemapper", [0, 1, 2, 3])
def test_post_chain_on_gpu_is_byte_exact(tonemapper):
    import torch
    from vimg_amd import hip
    s = scenes.json_scene("disney_spheres.json", res=(360, 160))
    hdr, _ = _dev(s).render(s.default_params(samples=8), stats=True)
    hdr = hdr.clone()
    hdr[0, 0, 1] = float("nan")          # NaN pixel -> magenta
    hdr[0, 1] = 0.0                      # black
    hdr[0, 2] = 1000.0                   # far above white
    got = hip.post_rgb8(hdr, tonemapper).cpu().numpy()
    want = O.post_rgb8(hdr.cpu().numpy(), tonemapper)
    # Reinhard maps a NaN-luminance pixel to black (change_luminance's l_in > 0 test fails),
    # the other three carry the NaN to the magenta marker
    assert got[0, 0].tolist() == ([0, 0, 0] if tonemapper == 2 else [255, 0, 255])
    assert np.array_equal(got, want)


def test_cpp_host_program_end_to_end(tmp_path):
    """v-img_amd/bin/vimg-amd: the C++ host (JSON loading, SAH BVH, PNG) around the C ABI —
    same picture as the Python path, and the -d single-pixel flag."""
    import os
    import subprocess
    import vimg_amd
    from vimg_amd import hip
    from PIL import Image
    exe = os.path.join(vimg_amd.abi.PKG_DIR, "bin", "vimg-amd")
    scene = os.path.join(scenes.SCENES, "cornell_box_spheres.json")
    out = str(tmp_path / "cli.png")
    r = subprocess.run([exe, "-f", scene, "-s", "4", "-c", "1", "-b", "1", "-o", out],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Mrays/s" in r.stdout
    s = scenes.json_scene("cornell_box_spheres.json")
    hdr, _ = _dev(s).render(s.default_params(samples=4))
    want = hip.post_rgb8(hdr, 1).cpu().numpy()
    assert np.array_equal(np.asarray(Image.open(out)), want)
    r = subprocess.run([exe, "-f", scene, "-s", "4", "-b", "1", "-d", "400 300"],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "Value of pixel in linear space" in r.stdout
    px = _dev(s).trace_pixel(s.default_params(samples=4), 400, 300)
    vals = [float(v) for v in r.stdout.split("(")[-1].split(")")[0].split(",")]
    assert np.allclose(vals, px, rtol=1e-6)
    assert subprocess.run([exe, "-f", "/no/such.json"], capture_output=True).returncode == 1


def test_axis_aligned_rays_take_the_exact_slab_path():
    """Rays with a zero direction component make 0 * inf = NaN possible in the slab test; the
    reference's (a<b)?b:a selects then propagate the NaN differently from v_min/v_max.  The
    kernel switches to the select form for such rays: hits and occlusion must equal the oracle's,
    including origins that lie exactly on bounding-box planes."""
    s = scenes.json_scene("disney_spheres.json")
    d = _dev(s)
    rng = np.random.default_rng(11)
    dirs = np.array([[0, 0, -1], [0, 0, 1], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0],
                     [0, 0.6, -0.8], [0.6, 0, -0.8], [0.6, 0.8, 0]], dtype=np.float32)
    # origins: random interior points, plus coordinates snapped onto box planes of the scene
    planes = np.array([-650, 650, -277.5, 277.5, 277, -77.5, -300, 300, 0, -177.5, -200, 40],
                      dtype=np.float32)
    o = rng.uniform(-600, 600, (3000, 3)).astype(np.float32)
    o[:, 1] = rng.uniform(-270, 270, 3000)
    o[:, 2] = rng.uniform(-270, 270, 3000)
    snap = rng.random((3000, 3)) < 0.4
    o[snap] = rng.choice(planes, snap.sum())
    rays = np.concatenate([o, dirs[rng.integers(0, len(dirs), 3000)]], 1)
    hg, hc = d.probe(O.PROBE_CLOSEST_HIT, rays), O.probe(s, O.PROBE_CLOSEST_HIT, rays)
    assert np.array_equal(hg[:, :4], hc[:, :4])                     # hit flag, t, prim, material
    assert hc[:, 0].sum() > 1000
    occ = np.concatenate([rays, rng.uniform(1, 1500, (3000, 1)).astype(np.float32)], 1)
    assert np.array_equal(d.probe(O.PROBE_OCCLUDED, occ), O.probe(s, O.PROBE_OCCLUDED, occ))


def test_scene_validation_rejects_broken_tables():
    """vimg_hip_scene_upload checks every index before anything reaches a kernel."""
    import ctypes as C
    from vimg_amd import abi, hip
    s = scenes.json_scene("disney_spheres.json", res=(32, 16))
    lib = abi.hip_lib()

    def upload(mutate):
        view = abi.Scene()
        C.memmove(C.byref(view), s.view, C.sizeof(view))
        keep = mutate(view)
        h = C.c_void_p()
        rc = lib.vimg_hip_scene_upload(C.byref(view), C.byref(h))
        if rc == 0:
            lib.vimg_hip_scene_free(h)
        return rc, lib.vimg_hip_last_error().decode(), keep

    def bad_prim(v):
        arr = (abi.Prim * v.num_prims)(*[v.prims[i] for i in range(v.num_prims)])
        arr[3].index = 9999
        v.prims = C.cast(arr, C.POINTER(abi.Prim))
        return arr

    def bad_node(v):
        n = v.bvh.num_nodes
        arr = (abi.BVHNode * n)(*[v.bvh.nodes[i] for i in range(n)])
        arr[1].first_index = 1          # a cycle: node 1 points at itself
        v.bvh.nodes = C.cast(arr, C.POINTER(abi.BVHNode))
        return arr

    def bad_light(v):
        arr = (abi.Light * v.num_lights)(*[v.lights[i] for i in range(v.num_lights)])
        arr[0].prim = 500
        v.lights = C.cast(arr, C.POINTER(abi.Light))
        return arr

    def bad_material(v):
        arr = (abi.Material * v.num_materials)(*[v.materials[i] for i in range(v.num_materials)])
        arr[0].tex = 77
        v.materials = C.cast(arr, C.POINTER(abi.Material))
        return arr

    for mut in (bad_prim, bad_node, bad_light, bad_material):
        rc, msg, _ = upload(mut)
        assert rc == -1 and msg, (mut.__name__, rc, msg)
    assert upload(lambda v: None)[0] == 0
    # mis needs a light
    def no_lights(v):
        v.num_lights = 0
    view = abi.Scene()
    C.memmove(C.byref(view), s.view, C.sizeof(view))
    view.num_lights = 0
    h = C.c_void_p()
    assert lib.vimg_hip_scene_upload(C.byref(view), C.byref(h)) == 0
    p = s.default_params(samples=1)
    out = np.zeros((16, 32, 3), np.float32)
    assert lib.vimg_hip_render_to_host(h, C.byref(p), out.ctypes.data_as(abi.Pf32), None) == -1
    lib.vimg_hip_scene_free(h)


# ------------------------------------------------------------------ pre-step on the GPU (8f rank 3)
@pytest.mark.parametrize("w,h,wrap_u,wrap_v", [(64, 64, 1, 1), (96, 40, 0, 2), (33, 17, 2, 0),
                                               (1, 1, 0, 0), (2, 300, 1, 1), (1024, 512, 1, 0)])
def test_precompute_mip_chain_is_byte_identical_to_the_host_build(w, h, wrap_u, wrap_v):
    """GPU mip chain (8-tap filter over bilinear fetches, reference src/image_texture.cpp:60-160)
    against libvimg_host's loops: same float expression tree, so the same bytes, on square, ragged,
    degenerate and large images and for all three wrap modes."""
    from vimg_amd import abi, hip, host
    rng = np.random.default_rng(w * 1000 + h)
    img = (rng.random((h, w, 3), dtype=np.float32) ** 2 * 4).astype(np.float32)   # HDR-ish range
    s = host.HostScene()
    t = s.add_texture_image(img, wrap_u, wrap_v)
    s.add_material("lambertian", tex=s.add_texture_const((0.5, 0.5, 0.5)))
    s.add_sphere((0, 0, 0), 1.0, 0)
    s.set_camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40, (8, 8))
    s.build_bvh()
    v = s.view.contents
    tex = v.textures[t]
    host_texels = np.ctypeslib.as_array(v.texels, (v.num_texels, 3))
    gpu, levels = hip.build_mip_chain(img, wrap_u, wrap_v)
    assert len(levels) == tex.num_levels
    base = tex.level_offset[0]
    for l, (off, lw, lh) in enumerate(levels):
        assert tex.level_offset[l] - base == off
        assert (lw, lh) == (max(w >> l, 1), max(h >> l, 1))
    want = host_texels[base:base + gpu.shape[0]]
    assert np.array_equal(gpu.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("w,h", [(32, 16), (200, 100), (65, 3), (1, 1), (2048, 1024)])
def test_precompute_env_cdfs_are_byte_identical_to_the_host_build(w, h):
    """Env-map importance tables (reference include/rng/sampling.h:113-135,168-197): the float
    prefix sums are sequential per row on both sides, so equality is exact; includes an image with
    black rows (uniform fallback) and a zero image."""
    from vimg_amd import abi, hip, host
    rng = np.random.default_rng(w + 7 * h)
    img = (rng.random((h, w, 3), dtype=np.float32) ** 4 * 30).astype(np.float32)
    if h > 2:
        img[1] = 0                       # a row of zero luminance: uniform conditional
    if (w, h) == (65, 3):
        img[:] = 0                       # everything black: uniform marginal too

    def host_tables():
        s = host.HostScene()
        t = s.add_texture_image(img)
        s.set_background_envmap(t)
        s.add_material("lambertian", tex=s.add_texture_const((0.5, 0.5, 0.5)))
        s.add_sphere((0, 0, 0), 1.0, 0)
        s.set_camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40, (8, 8))
        s.build_bvh()
        v = s.view.contents
        cdf = np.ctypeslib.as_array(v.cdf_pool, (v.num_cdf,))
        r0, c0 = v.background.row_cdf_offset, v.background.col_cdf_offset
        texels = np.ctypeslib.as_array(v.texels, (v.num_texels, 3)).copy()
        return cdf[r0:r0 + h + 1].copy(), cdf[c0:c0 + h * (w + 1)].reshape(h, w + 1).copy(), texels

    row_h, col_h, tex_h = host_tables()
    row_g, col_g = hip.build_env_cdfs(img)
    assert np.array_equal(row_g.view(np.uint32), row_h.view(np.uint32))
    assert np.array_equal(col_g.view(np.uint32), col_h.view(np.uint32))
    # and through the hook: the host library assembling a scene with the GPU builders installed
    hip.install_gpu_precompute(True)
    try:
        row_i, col_i, tex_i = host_tables()
    finally:
        hip.install_gpu_precompute(False)
    assert np.array_equal(row_i.view(np.uint32), row_h.view(np.uint32))
    assert np.array_equal(col_i.view(np.uint32), col_h.view(np.uint32))
    assert np.array_equal(tex_i.view(np.uint32), tex_h.view(np.uint32))


def test_precompute_8bit_conversions_and_render_with_gpu_built_tables():
    from vimg_amd import hip, host
    rng = np.random.default_rng(11)
    vals = rng.integers(0, 256, size=(257, 129, 3), dtype=np.uint8)
    assert np.array_equal(hip.lut8_to_float(vals, host.srgb8_lut()).view(np.uint32),
                          host.srgb8_to_linear(vals).view(np.uint32))
    assert np.array_equal(hip.rgb8_to_normal(vals, 0.7).view(np.uint32),
                          host.rgb8_to_normal(vals, 0.7).view(np.uint32))
    # a scene assembled with the GPU builders renders the same bits as one assembled on the host
    ref = scenes.feature_scene(res=(48, 32))
    hip.install_gpu_precompute(True)
    try:
        dev = scenes.feature_scene(res=(48, 32))
    finally:
        hip.install_gpu_precompute(False)
    p = ref.default_params(samples=4, depth=6)
    a, _ = _dev(ref).render_to_host(p)
    b, _ = _dev(dev).render_to_host(p)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


# ------------------------------------------------------------------ heatmap integrator (8f rank 2)
@pytest.mark.parametrize("scene_name,bvh", [("disney_spheres.json", "sweep"), ("glass_in_box.json", "binned"),
                                            ("feature", "sweep"), ("config4", "sweep")])
def test_heatmap_matches_oracle_bit_for_bit(scene_name, bvh):
    """vimg_hip_render_heatmap (reference heatmap_img + BVH::hit<float>): the cost is an integer
    count of node visits and primitive tests and the colour map is plain float arithmetic, so the
    picture is bit-identical; whole image and 3 shards."""
    from vimg_amd import abi, dist as vdist
    if scene_name == "feature":
        s = scenes.feature_scene(res=(72, 40), envmap=False, lens=True)
    elif scene_name == "config4":
        s = scenes.config4_scene(res=(96, 54), n_lat=48, env=(128, 64))
    else:
        s = scenes.json_scene(scene_name, res=(100, 60),
                              bvh=abi.BVH_SWEEP if bvh == "sweep" else abi.BVH_BINNED)
    p = s.default_params(samples=4)
    d = _dev(s)
    for factor in (-1.0, 7.5):
        cpu, counts = O.heatmap(s, p, factor=factor)
        gpu = d.render_heatmap(p, factor).cpu().numpy()
        assert np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32)), scene_name
    assert counts.max() > 0
    w, h = s.resolution
    slabs = []
    for r in range(3):
        pr = s.default_params(samples=4, tile_rank=r, tile_world=3)
        slabs.append(d.render_heatmap(pr, 7.5).cpu().numpy())
    stride = max(x.shape[0] for x in slabs)
    padded = np.zeros((3, stride, 3), dtype=np.float32)
    for r, x in enumerate(slabs):
        padded[r, :x.shape[0]] = x
    assert np.array_equal(vdist.assemble_numpy(padded, w, h, 3), gpu)


# ------------------------------------------------------------------ GPU BVH builder (8f rank 4)
@pytest.mark.parametrize("builder", ["lbvh", "ploc"])
@pytest.mark.parametrize("name", ["one sphere", "two prims", "coincident", "disney_spheres", "big_mesh", "config4"])
def test_gpu_builders_give_a_valid_tree_and_the_same_picture(name, builder):
    """vimg_hip_build_lbvh (Morton order + radix tree + bottom-up boxes, one primitive per leaf) and
    vimg_hip_build_ploc (locally-ordered clustering + SAH leaves of up to 8), emitted in the
    reference's BVH layout.  The tree must satisfy the layout's invariants; GPU kernels and oracle
    then walk the same tree, so their images are bit-identical as with the host-built trees; and
    the picture does not depend on which tree was built (ties aside)."""
    from test_host_and_abi import _check_tree
    from vimg_amd import hip, host
    if name == "coincident":
        # 40 spheres around ONE centre (all Morton codes equal, every split plane degenerate: the builders
        # fall back to splits by position) beside a light and a floor sphere
        s = host.HostScene()
        m = s.add_material("lambertian", tex=s.add_texture_const((0.6, 0.7, 0.5)))
        lt = s.add_material("diffuse_light", emit=(6, 6, 6))
        for k in range(40):
            s.add_sphere((0.0, 0.5, 0.0), 0.3 + 0.01 * k, m)
        s.add_sphere((0.5, 3.0, 0.5), 0.8, lt)
        s.add_sphere((0.0, -100.0, 0.0), 100.0, m)
        s.set_camera((0, 1.5, 6), (0, 0.6, 0), (0, 1, 0), 40, (40, 32))
        s.set_render_defaults("mis", 4, 8)
        s.build_bvh()
    elif name in ("one sphere", "two prims"):
        s = host.HostScene()
        m = s.add_material("lambertian", tex=s.add_texture_const((0.7, 0.6, 0.5)))
        lt = s.add_material("diffuse_light", emit=(5, 5, 5))
        s.add_sphere((0, 0, 0), 1.0, lt if name == "one sphere" else m)
        if name == "two prims":
            s.add_sphere((0.5, 2.5, 0.5), 0.7, lt)
        s.set_camera((0, 1, 6), (0, 0.5, 0), (0, 1, 0), 40, (40, 32))
        s.set_render_defaults("mis", 4, 8)
        s.build_bvh()
    elif name == "disney_spheres":
        s = scenes.json_scene("disney_spheres.json", res=(96, 48))
    elif name == "big_mesh":
        s = scenes.big_mesh_scene(res=(96, 64))
    else:
        s = scenes.config4_scene(res=(96, 54), n_lat=48, env=(128, 64))
    p = s.default_params(samples=4)
    sah_img, _ = _dev(s).render_to_host(p)
    s.build_bvh_with(hip.lbvh_builder() if builder == "lbvh" else hip.ploc_builder())
    depth = _check_tree(s, leaf_max=1 if builder == "lbvh" else 8)
    n = s.view.contents.num_prims
    if builder == "lbvh":
        assert s.view.contents.bvh.num_nodes == 2 * n - 1
    assert s.view.contents.bvh.num_nodes <= 2 * n - 1 and depth <= 64
    cpu, cst, _ = O.render(s, p)
    gpu, gst = _dev(s).render_to_host(p)
    _compare_images(gpu, cpu, f"{name} ({builder})", min_exact=0.995)
    assert gst.paths == cst.paths and abs(gst.rays - cst.rays) <= max(8, 2e-3 * cst.rays)
    same = (gpu.view(np.uint32) == sah_img.view(np.uint32)).all(axis=-1).mean()
    assert same > 0.97 and abs(gpu.mean() - sah_img.mean()) <= 0.03 * abs(sah_img.mean()) + 1e-6


def _tree_cost(s):
    """Surface-area cost of the scene's tree under the reference's model (traversal 0.5, intersection 1:
    include/bvh.h:17-20): sum over nodes of area / root area x (0.5 for a node with children, its
    primitive count for a leaf)."""
    nodes, bb, _, _ = s.bvh_arrays()
    first, count = nodes[:, 0].astype(np.int64), nodes[:, 1].astype(np.int64)

    def area(lo, hi):
        d = (hi - lo).astype(np.float64)
        return d[..., 0] * d[..., 1] + d[..., 0] * d[..., 2] + d[..., 1] * d[..., 2]
    a = np.zeros(len(nodes))
    a[0] = area(bb[0], bb[2])
    inner = np.nonzero(count == 0)[0]
    base = 2 * first[inner] + 2
    a[first[inner]] = area(bb[base], bb[base + 2])
    a[first[inner] + 1] = area(bb[base + 1], bb[base + 3])
    return float((a * np.where(count == 0, 0.5, count)).sum() / a[0])


@pytest.mark.parametrize("builder", ["lbvh", "ploc"])
def test_gpu_builders_are_deterministic_and_ploc_is_close_to_the_sweep(builder):
    """The builders' kernels number nodes with atomic counters, but the tree they describe - and so the
    arrays in the reference's layout - must not depend on the order waves ran in: two builds give the
    same bytes.  And the quality of vimg_hip_build_ploc is made by its kernels (agglomeration, leaves
    ended by the cost model, the top rebuilt by binned SAH): on a 27 K-triangle scene its tree costs no
    more than 1.05 x the host's sweep-SAH tree (src/bvh/sweep_bvh.cpp:74-216, the bar) under the
    reference's cost model, where the plain LBVH costs more than PLOC."""
    from vimg_amd import hip
    s = scenes.config4_scene(res=(96, 54), n_lat=96, env=(64, 32))
    sweep = _tree_cost(s)
    fn = hip.lbvh_builder() if builder == "lbvh" else hip.ploc_builder()
    s.build_bvh_with(fn)
    first = [np.array(x, copy=True) for x in s.bvh_arrays()[:3]] + [s.bvh_arrays()[3]]
    cost = _tree_cost(s)
    s.build_bvh_with(fn)
    again = s.bvh_arrays()
    assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(first[:3], again[:3])) and first[3] == again[3]
    if builder == "ploc":
        assert cost <= 1.05 * sweep, (cost, sweep)
        s.build_bvh_with(hip.lbvh_builder())
        assert _tree_cost(s) > cost


@pytest.mark.parametrize("leaf_cap", [10 ** 9, 300])
def test_leaves_of_more_than_127_primitives(leaf_cap):
    """A caller's builder may hand over leaves larger than the 7-bit count of the packed child
    reference (this repository's builders stop at 8).  The upload chains such a leaf (127 primitives
    per link, the leaf's own box on both sides) so that its primitives are still tested in
    obj_indices order: the image equals the oracle's on the SAME tree - a single leaf with all 420
    primitives of the scene, then a median split into two leaves of ~250 - on the lane-bound and the
    pooled kernel; only the event counts gain the chains' node visits."""
    import ctypes as C
    from vimg_amd import abi
    BUILDER = C.CFUNCTYPE(C.c_int, C.c_uint32, abi.Pf32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                          C.c_void_p, abi.Pf32, C.POINTER(C.c_uint32))

    def big_leaves(n, bounds6, num_nodes, max_depth, nodes_p, bb_p, obj_p):
        b = np.ctypeslib.as_array(bounds6, (n, 6)).copy()
        nodes = np.ctypeslib.as_array(C.cast(nodes_p, C.POINTER(C.c_uint32)), (2 * n - 1, 2))
        bb = np.ctypeslib.as_array(bb_p, (2 * (2 * n - 1) + 3, 3))
        obj = np.ctypeslib.as_array(obj_p, (n,))
        centre = (b[:, :3] + b[:, 3:]) * 0.5
        state = {"next": 1, "pos": 0, "depth": 0}

        def box(ids):
            return b[ids, :3].min(0), b[ids, 3:].max(0)

        def build(node, ids, d):
            state["depth"] = max(state["depth"], d)
            if len(ids) <= leaf_cap:
                nodes[node] = (state["pos"], len(ids))
                obj[state["pos"]:state["pos"] + len(ids)] = ids
                state["pos"] += len(ids)
                return
            lo, hi = box(ids)
            axis = int(np.argmax(hi - lo))
            order = ids[np.argsort(centre[ids, axis], kind="stable")]
            halves = (order[:len(ids) // 2], order[len(ids) // 2:])
            first = state["next"]
            state["next"] += 2
            nodes[node] = (first, 0)
            for k in (0, 1):
                c_lo, c_hi = box(halves[k])
                bb[2 * first + 2 + k], bb[2 * first + 4 + k] = c_lo, c_hi
            build(first, halves[0], d + 1)
            build(first + 1, halves[1], d + 1)

        ids = np.arange(n)
        bb[0], bb[2] = box(ids)
        build(0, ids, 1)
        num_nodes[0], max_depth[0] = state["next"], state["depth"]
        return 0

    cb = BUILDER(big_leaves)
    s = scenes.big_mesh_scene(res=(64, 48), n=14)       # 2 * 14 * 14 = 392 triangles + walls and a light
    n = s.view.contents.num_prims
    assert n > 2 * 127
    p = s.default_params(samples=4)
    sah, _ = _dev(s).render_to_host(p)
    s.build_bvh_with(C.cast(cb, C.c_void_p))
    bvh = s.view.contents.bvh
    biggest = max(bvh.nodes[i].obj_count for i in range(bvh.num_nodes))
    assert biggest > 127 and (biggest == n) == (leaf_cap > n)
    cpu, cst, _ = O.render(s, p)
    for sched in ("lane", "cu"):
        gpu, gst = _dev_opts(s, scheduler=sched, **({"pool_slots": 40} if sched == "cu" else {})).render_to_host(p)
        _compare_images(gpu, cpu, f"leaves of up to {biggest} primitives ({sched})", min_exact=0.995)
        assert gst.paths == cst.paths and gst.rays == cst.rays and gst.prim_tests <= cst.prim_tests
        assert gst.internal_visits > cst.internal_visits          # the chains' links
    same = (gpu.view(np.uint32) == sah.view(np.uint32)).all(axis=-1).mean()
    assert same > 0.97


def test_launch_policy_picks_the_builds_design_md_names():
    """vimg_hip.hip:make_launch: every launch - a full frame, its shards down to an
    eighth, a frame of test size, trace_pixel - gets the CU scheduler, in the build for trees that sit
    in LDS or the one for trees in global memory; the lane-bound kernel by name.  (What each of them
    renders is the business of the parity tests above.)"""
    s = scenes.json_scene("disney_spheres.json")                       # 1800 x 800
    d = _dev(s)
    for tw in (1, 2, 4, 8):
        assert d.kernel_for(s.default_params(samples=4, tile_world=tw)) == "render_cu_kernel<false>", tw
    small = _dev(scenes.json_scene("disney_spheres.json", res=(136, 72)))
    assert small.kernel_for(s.default_params(samples=4)) == "render_cu_kernel<false>"
    deep = scenes.config4_scene(n_lat=48, env=(128, 64))               # 1366 x 768, tree beyond LDS
    dd = _dev(deep)
    for tw in (1, 2, 4, 8):
        assert dd.kernel_for(deep.default_params(samples=4, tile_world=tw)) == "render_cu_kernel<true,deep>", tw
    by_name = _dev_opts(s, scheduler="lane")
    assert by_name.kernel_for(s.default_params(samples=4)).startswith("render_kernel<false")


def test_a_tree_deeper_than_the_lds_stack():
    """The stack bound of the boundary is 94 levels; the pooled kernels keep at most the first 32
    entries of a lane's stack in LDS and the rest in global memory (by policy: a tree that deep would
    otherwise leave its workgroup no LDS for path slots).  A caller's builder hands over a
    "caterpillar" - every internal node has one leaf child and the rest of the primitives as the
    other - 49 levels deep; the image and the event counts must be the oracle's on the SAME tree, on
    the lane-bound kernel and on both pooled builds with their default stack split."""
    import ctypes as C
    from vimg_amd import abi
    BUILDER = C.CFUNCTYPE(C.c_int, C.c_uint32, abi.Pf32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                          C.c_void_p, abi.Pf32, C.POINTER(C.c_uint32))

    def caterpillar(n, bounds6, num_nodes, max_depth, nodes_p, bb_p, obj_p):
        b = np.ctypeslib.as_array(bounds6, (n, 6)).copy()
        nodes = np.ctypeslib.as_array(C.cast(nodes_p, C.POINTER(C.c_uint32)), (2 * n - 1, 2))
        bb = np.ctypeslib.as_array(bb_p, (2 * (2 * n - 1) + 3, 3))
        obj = np.ctypeslib.as_array(obj_p, (n,))
        order = np.argsort((b[:, 0] + b[:, 3]) + 0.37 * (b[:, 2] + b[:, 5]), kind="stable")
        obj[:] = order
        bb[0], bb[2] = b[:, :3].min(0), b[:, 3:].max(0)
        node, nxt = 0, 1
        for i in range(n - 1):              # node: leaf {order[i]} | everything behind it
            first = nxt
            nxt += 2
            nodes[node] = (first, 0)
            rest = order[i + 1:]
            bb[2 * first + 2], bb[2 * first + 4] = b[order[i], :3], b[order[i], 3:]
            bb[2 * first + 3], bb[2 * first + 5] = b[rest, :3].min(0), b[rest, 3:].max(0)
            nodes[first] = (i, 1)
            node = first + 1
        nodes[node] = (n - 1, 1)
        num_nodes[0], max_depth[0] = nxt, n
        return 0

    cb = BUILDER(caterpillar)
    s = scenes.big_mesh_scene(res=(64, 48), n=4)        # 2 * 4 * 4 = 32 triangles + walls and a light
    s.build_bvh_with(C.cast(cb, C.c_void_p))
    bvh = s.view.contents.bvh
    assert 34 < bvh.max_depth <= 92, bvh.max_depth
    p = s.default_params(samples=6)
    cpu, cst, _ = O.render(s, p)
    lane, lst = _dev_opts(s, scheduler="lane").render_to_host(p)
    _compare_images(lane, cpu, f"caterpillar tree of depth {bvh.max_depth}")
    for k in ("closest_rays", "shadow_rays", "internal_visits", "leaf_visits", "prim_tests"):
        a, b = getattr(lst, k), getattr(cst, k)
        assert abs(a - b) <= max(64, 1e-3 * b), (k, a, b)
    for opts in (dict(scheduler="cu"), dict(scheduler="cu", lds_stack=4, pool_segments=2)):
        d = _dev_opts(s, **opts)
        sched = str(opts)
        assert "deep" in d.kernel, d.kernel
        gpu, gst = d.render_to_host(p)
        assert np.array_equal(gpu.view(np.uint32), lane.view(np.uint32)), (sched, d.kernel)
        assert gst.as_dict() == lst.as_dict(), sched


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` with no rendezvous in the environment must start the ranks itself
    (fresh child processes; the driver invokes it exactly like this) and relay one JSON line.
    Rehearsal form: both ranks on this GPU, host-staged gather (gloo), strong scaling on a small
    frame, assembled image verified against the single-GPU frame."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--one-device", "--spp", "8", "--steps", "1", "--warmup", "0", "--no-cpu", "--verify",
                        "--res", "456", "200"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["weak"]["value"] > 0 and out["weak"]["resolution"] == [648, 280]      # the sqrt(2)-grown frame rides along
    assert "verify: assembled frame is bit-identical" in r.stderr
    assert 0 < out["efficiency"] <= 1.5
