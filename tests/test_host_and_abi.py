"""Host side (libvimg_host.so) and the C ABI surface — no GPU needed."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
from PIL import Image

import scenes
import vimg_amd
from vimg_amd import abi, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(vimg_(?:hip|host)_[a-z0-9_]+)\s*\(", text))


def test_hip_library_exports_every_declared_symbol():
    declared = _declared("vimg_hip.h")
    assert declared == set(abi.HIP_SYMBOLS), declared ^ set(abi.HIP_SYMBOLS)
    lib = abi.hip_lib()                       # loads on a machine without a GPU
    for name in declared:
        assert hasattr(lib, name), name


def test_host_library_exports_every_declared_symbol():
    declared = _declared("vimg_host.h")
    assert declared == set(abi.HOST_SYMBOLS), declared ^ set(abi.HOST_SYMBOLS)
    lib = abi.host_lib()
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match_the_c_headers():
    # sizes the C compiler gives the boundary structs (checked against a tiny C probe)
    import subprocess, tempfile
    src = '#include <stdio.h>\n#include "vimg_scene.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",' \
          'sizeof(VimgCamera),sizeof(VimgMesh),sizeof(VimgSphere),sizeof(VimgMaterial),sizeof(VimgTexture),' \
          'sizeof(VimgTextureRG),sizeof(VimgBackground),sizeof(VimgBVH),sizeof(VimgScene),' \
          'sizeof(VimgRenderParams),sizeof(VimgRenderStats));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "p.c"), "-o",
                        os.path.join(d, "p")], check=True)
        got = [int(v) for v in subprocess.run([os.path.join(d, "p")], capture_output=True,
                                              text=True, check=True).stdout.split()]
    want = [C.sizeof(t) for t in (abi.Camera, abi.Mesh, abi.Sphere, abi.Material, abi.Texture,
                                  abi.TextureRG, abi.Background, abi.BVH, abi.Scene,
                                  abi.RenderParams, abi.RenderStats)]
    assert got == want


def test_hip_path_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for the GPU-less container")
    from vimg_amd import hip
    s = scenes.json_scene("disney_spheres.json", res=(32, 16))
    with pytest.raises(hip.HipError) as e:
        hip.DeviceScene(s)
    assert "[-2]" in str(e.value)            # VIMG_E_DEVICE, no silent CPU fallback
    with pytest.raises(hip.HipError):
        hip.device_count()


def test_product_package_never_touches_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "v-img_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle/" not in text and "liboracle" not in text and "oracle_lib" not in text, f


# ------------------------------------------------------------------------------- JSON loader
def test_json_loader_semantics():
    s = scenes.json_scene("disney_spheres.json")
    v = s.view.contents
    p = s.default_params()
    assert (v.camera.res_x, v.camera.res_y, v.camera.vfov_deg) == (1800, 800, 25.0)
    assert p.samples == 512 and p.depth == 0xFFFFFFFF          # "depth": -1 -> uint32 wrap
    assert v.num_prims == 18 and v.num_tris == 12 and v.num_spheres == 6 and v.num_meshes == 6
    assert v.num_materials == 10 and v.num_lights == 2
    # emissive triangles are registered last to first (mesh_loading.cpp:96-103)
    assert [(v.lights[i].type, v.lights[i].prim) for i in range(2)] == [(0, 11), (0, 10)]
    # "background" is parsed and ignored: black, not a light (json_scene.cpp:202-206)
    assert v.background.type == abi.BG_CONST and list(v.background.col) == [0, 0, 0]
    assert v.camera.aperture_radius == 0.0 and v.camera.focal_dist == 1.0
    # quad vertices go through scale -> rotate -> translate in file order
    verts = np.ctypeslib.as_array(v.vertices, (v.num_vertices, 3))
    assert np.allclose(verts[:4, 2], -277.5) and np.allclose(np.abs(verts[:4, 0]), 650)


def test_json_loader_errors_and_defaults():
    with pytest.raises(host.HostError):
        vimg_amd.HostScene.from_json("/nonexistent/scene.json")
    with pytest.raises(host.HostError):
        vimg_amd.HostScene.from_json_text("{ not json")
    with pytest.raises(host.HostError):
        vimg_amd.HostScene.from_json_text(json.dumps({"materials": [], "surfaces": []}))
    base = {"camera": {"transform": {"from": [0, 0, 5], "at": [0, 0, 0]}},
            "materials": [{"type": "lambertian", "name": "w", "albedo": [1, 1, 1]},
                          {"type": "diffuse_light", "name": "l", "albedo": [3, 3, 3]}],
            "surfaces": [{"type": "sphere", "mat_name": "w", "center": [0, 0, 0]},
                         {"type": "sphere", "mat_name": "l", "center": [0, 3, 0], "radius": 0.5}]}
    s = vimg_amd.HostScene.from_json_text(json.dumps(base))
    p = s.default_params()
    v = s.view.contents
    assert (v.camera.res_x, v.camera.res_y, v.camera.vfov_deg) == (500, 500, 40.0)
    assert (p.samples, p.depth, p.integrator) == (30, 30, abi.INTEGRATOR_S_NORMAL)
    assert v.spheres[0].radius == 1.0 and v.num_lights == 1 and v.lights[0].prim == 1
    bad = dict(base, integrator={"type": "normal"})       # README spelling: falls back to s_normal
    assert vimg_amd.HostScene.from_json_text(json.dumps(bad)).default_params().integrator == 0
    unknown = dict(base, surfaces=[{"type": "sphere", "mat_name": "nope", "center": [0, 0, 0]}])
    with pytest.raises(host.HostError):
        vimg_amd.HostScene.from_json_text(json.dumps(unknown))
    badmat = dict(base, materials=[{"type": "plastic", "name": "w"}])
    with pytest.raises(host.HostError):
        vimg_amd.HostScene.from_json_text(json.dumps(badmat))


# ------------------------------------------------------------------------------- BVH builders
def test_json_mesh_surface_loads_obj_positions_and_fans_polygons(tmp_path):
    """"type": "mesh" (reference src/scene_loading/json_scene.cpp:366-385 + load_from_obj,
    src/geometry/mesh_loading.cpp:21-65): positions only, path relative to the scene file, every
    face a fan, 1-based / negative indices, v/vt/vn tokens; the transform is applied to positions."""
    (tmp_path / "assets").mkdir()
    (tmp_path / "scenes").mkdir()
    obj = """# unit cube
v -0.5 -0.5 -0.5
v  0.5 -0.5 -0.5
v  0.5  0.5 -0.5
v -0.5  0.5 -0.5
v -0.5 -0.5  0.5
v  0.5 -0.5  0.5
v  0.5  0.5  0.5
v -0.5  0.5  0.5
vn 0 0 1
vt 0 0
f 1 4 3 2
f 5/1/1 6/1/1 7/1/1 8/1/1
f -8 -7 -3 -4
f 2//1 3//1 7//1 6//1
f 3 4 8 7
f 4 1 5
f 4 5 8
"""
    (tmp_path / "assets" / "cube.obj").write_text(obj)
    scene = {
        "camera": {"transform": {"from": [3, 2, 5], "at": [0, 0, 0], "up": [0, 1, 0]}, "vfov": 30,
                   "resolution": [32, 24]},
        "sampler": {"samples": 2, "depth": 3},
        "integrator": {"type": "mis"},
        "materials": [{"type": "lambertian", "name": "m", "texture": {"type": "constant", "albedo": [0.8, 0.8, 0.8]}},
                      {"type": "diffuse_light", "name": "l", "albedo": [3, 3, 3]}],
        "surfaces": [{"type": "mesh", "filename": "../assets/cube.obj", "mat_name": "m",
                      "transform": [{"translate": [0, 0.5, 0]}, {"scale": [1, 9, 4]}]},
                     {"type": "quad", "mat_name": "l", "transform": [{"translate": [0, 0, 9]}]}],
    }
    path = tmp_path / "scenes" / "s.json"
    path.write_text(json.dumps(scene))
    s = host.HostScene.from_json(str(path))
    v = s.view.contents
    assert v.num_meshes == 2 and v.num_prims == 12 + 2
    m = v.meshes[0]
    assert (m.num_vertices, m.has_normals, m.num_uv_sets, m.color_tex_uv) == (8, 0, 0, abi.NO_UV)
    verts = np.ctypeslib.as_array(v.vertices, (v.num_vertices, 3))[:8]
    want = np.array([[-.5, -.5, -.5], [.5, -.5, -.5], [.5, .5, -.5], [-.5, .5, -.5], [-.5, -.5, .5],
                     [.5, -.5, .5], [.5, .5, .5], [-.5, .5, .5]], dtype=np.float32)
    want = (want + np.float32([0, 0.5, 0])) * np.float32([1, 9, 4])     # translate, then scale
    assert np.array_equal(verts, want)
    tri = np.ctypeslib.as_array(v.tri_indices, (v.num_tris, 3))[:12]
    assert tri.tolist() == [[0, 3, 2], [0, 2, 1], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4],
                            [1, 2, 6], [1, 6, 5], [2, 3, 7], [2, 7, 6], [3, 0, 4], [3, 4, 7]]
    # errors: missing file, face before its vertices
    scene["surfaces"][0]["filename"] = "nope.obj"
    path.write_text(json.dumps(scene))
    with pytest.raises(host.HostError, match="cannot open"):
        host.HostScene.from_json(str(path))
    (tmp_path / "scenes" / "bad.obj").write_text("v 0 0 0\nf 1 2 3\n")
    scene["surfaces"][0]["filename"] = "bad.obj"
    path.write_text(json.dumps(scene))
    with pytest.raises(host.HostError, match="not defined"):
        host.HostScene.from_json(str(path))


@pytest.mark.parametrize("kind", [abi.BVH_SWEEP, abi.BVH_BINNED])
def test_bvh_invariants(kind):
    s = scenes.big_mesh_scene()
    s.build_bvh(kind)
    nodes, bb, obj, depth = s.bvh_arrays()
    v = s.view.contents
    n_prims = v.num_prims
    assert sorted(obj.tolist()) == list(range(n_prims))          # every primitive exactly once
    # primitive bounds
    verts = np.ctypeslib.as_array(v.vertices, (v.num_vertices, 3))
    lo = np.zeros((n_prims, 3), np.float32)
    hi = np.zeros((n_prims, 3), np.float32)
    for i in range(n_prims):
        pr = v.prims[i]
        if pr.type == abi.PRIM_TRIANGLE:
            m = v.meshes[v.tri_mesh[pr.index]]
            idx = [m.first_vertex + v.tri_indices[pr.index * 3 + k] for k in range(3)]
            lo[i], hi[i] = verts[idx].min(0), verts[idx].max(0)
        else:
            sp = v.spheres[pr.index]
            c = np.array(list(sp.center), np.float32)
            lo[i], hi[i] = c - sp.radius, c + sp.radius

    seen = np.zeros(len(nodes), bool)

    def visit(n, box_lo, box_hi, d):
        assert not seen[n]
        seen[n] = True
        first, count = int(nodes[n][0]), int(nodes[n][1])
        if count:
            assert count <= 8
            ids = obj[first:first + count]
            assert np.all(lo[ids] >= box_lo - 0) and np.all(hi[ids] <= box_hi + 0)
            return d
        base = 2 * first + 2
        dl = visit(first, bb[base], bb[base + 2], d + 1)
        dr = visit(first + 1, bb[base + 1], bb[base + 3], d + 1)
        # "right child should be larger" (half surface area)
        def hsa(a, b):
            e = b - a
            return e[0] * e[1] + e[0] * e[2] + e[1] * e[2]
        assert hsa(bb[base], bb[base + 2]) <= hsa(bb[base + 1], bb[base + 3])
        return max(dl, dr)

    import sys
    sys.setrecursionlimit(10000)
    assert visit(0, bb[0], bb[2], 1) == depth
    assert seen.all()


def _check_tree(s, leaf_max):
    """Invariants of the reference's BVH layout: every primitive in exactly one leaf, every box
    holds what hangs below it, siblings adjacent, max_depth = levels.  Returns the depth found."""
    nodes, bb, obj, depth = s.bvh_arrays()
    v = s.view.contents
    n_prims = v.num_prims
    assert sorted(obj.tolist()) == list(range(n_prims))
    verts = np.ctypeslib.as_array(v.vertices, (v.num_vertices, 3)) if v.num_vertices else None
    lo = np.zeros((n_prims, 3), np.float32)
    hi = np.zeros((n_prims, 3), np.float32)
    for i in range(n_prims):
        pr = v.prims[i]
        if pr.type == abi.PRIM_TRIANGLE:
            m = v.meshes[v.tri_mesh[pr.index]]
            idx = [m.first_vertex + v.tri_indices[pr.index * 3 + k] for k in range(3)]
            lo[i], hi[i] = verts[idx].min(0), verts[idx].max(0)
        else:
            sp = v.spheres[pr.index]
            c = np.array(list(sp.center), np.float32)
            lo[i], hi[i] = c - sp.radius, c + sp.radius
    seen = np.zeros(len(nodes), bool)
    stack = [(0, bb[0], bb[2], 1)]
    deepest = 0
    while stack:
        n, box_lo, box_hi, d = stack.pop()
        assert not seen[n]
        seen[n] = True
        deepest = max(deepest, d)
        first, count = int(nodes[n][0]), int(nodes[n][1])
        if count:
            assert count <= leaf_max
            ids = obj[first:first + count]
            assert np.all(lo[ids] >= box_lo) and np.all(hi[ids] <= box_hi)
            continue
        base = 2 * first + 2
        for k in (0, 1):
            c_lo, c_hi = bb[base + k], bb[base + 2 + k]
            assert np.all(c_lo >= box_lo) and np.all(c_hi <= box_hi)
            stack.append((first + k, c_lo, c_hi, d + 1))
    assert seen.all() and deepest == depth
    return depth


def test_build_bvh_with_a_supplied_builder():
    """vimg_host_build_bvh_with: the builder gets the primitive bounds the host builders use and
    fills the reference's arrays; here a median-split builder written in Python.  The image does
    not depend on the tree (different trees only reorder the tests; ties aside)."""
    import oracle_lib as O
    BUILDER = C.CFUNCTYPE(C.c_int, C.c_uint32, abi.Pf32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                          C.c_void_p, abi.Pf32, C.POINTER(C.c_uint32))

    def median_split(n, bounds6, num_nodes, max_depth, nodes_p, bb_p, obj_p):
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
            if len(ids) <= 2:
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

    cb = BUILDER(median_split)
    s = scenes.json_scene("cornell_box_spheres.json", res=(48, 48))
    p = s.default_params(samples=4)
    ref, _, _ = O.render(s, p, threads=2)
    s.build_bvh_with(C.cast(cb, C.c_void_p))
    assert _check_tree(s, leaf_max=2) >= 3
    img, _, _ = O.render(s, p, threads=2)
    same = (img.view(np.uint32) == ref.view(np.uint32)).all(axis=-1).mean()
    assert same > 0.98 and abs(img.mean() - ref.mean()) < 0.02 * ref.mean()
    with pytest.raises(host.HostError, match="builder failed"):
        s.build_bvh_with(C.cast(BUILDER(lambda *a: -1), C.c_void_p))


# ------------------------------------------------------------------------------- textures / env
def test_mip_chain_and_env_cdfs():
    s = scenes.feature_scene()
    v = s.view.contents
    imgs = [v.textures[i] for i in range(v.num_textures) if v.textures[i].type == abi.TEX_IMAGE]
    t = imgs[0]                                   # 32x32 -> ceil(log2 32) = 5 levels
    assert (t.width, t.height, t.num_levels) == (32, 32, 5)
    sizes = [max(32 >> l, 1) ** 2 for l in range(5)]
    offs = [t.level_offset[l] for l in range(5)]
    assert [offs[i + 1] - offs[i] for i in range(4)] == sizes[:4]
    tex = np.ctypeslib.as_array(v.texels, (v.num_texels, 3))
    lvl0 = tex[offs[0]:offs[0] + 1024].reshape(32, 32, 3)
    lvl1 = tex[offs[1]:offs[1] + 256].reshape(16, 16, 3)
    assert lvl1.min() >= 0                        # negative filter lobes are clamped
    assert abs(lvl1.mean() - lvl0.mean()) < 0.03  # the 8-tap filter has unit DC gain
    env = imgs[-1]
    assert (env.width, env.height) == (32, 16) and v.background.type == abi.BG_ENVMAP
    cdf = np.ctypeslib.as_array(v.cdf_pool, (v.num_cdf,))
    rows = cdf[v.background.row_cdf_offset:v.background.row_cdf_offset + 17]
    assert rows[0] == 0 and abs(rows[-1] - 1) < 1e-6 and np.all(np.diff(rows) >= 0)
    cols = cdf[v.background.col_cdf_offset:v.background.col_cdf_offset + 16 * 33].reshape(16, 33)
    assert np.all(cols[:, 0] == 0) and np.allclose(cols[:, -1], 1, atol=1e-6)
    assert np.all(np.diff(cols, axis=1) >= 0)
    # the env map is registered as a light (after the emissive quad's two triangles)
    assert v.lights[v.num_lights - 1].type == abi.LIGHT_BACKGROUND


def test_precompute_hooks_and_8bit_conversions():
    """vimg_host_set_precompute: an installed builder replaces the host loops (and its failure is
    an error, not a silent second path); the 8-bit conversions follow the reference's formulas."""
    rng = np.random.default_rng(5)
    img = rng.random((8, 16, 3), dtype=np.float32)
    calls = []
    MIP = C.CFUNCTYPE(C.c_int, C.c_uint32, C.c_uint32, abi.Pf32, C.c_uint32, C.c_uint32, abi.Pf32)
    CDF = C.CFUNCTYPE(C.c_int, abi.Pf32, C.c_uint32, C.c_uint32, abi.Pf32, abi.Pf32)

    def mip(w, h, level0, wu, wv, out):
        calls.append(("mip", w, h, wu, wv))
        n = w * h + (w // 2) * (h // 2) + (w // 4) * (h // 4)     # ceil(log2 8) = 3 levels
        np.ctypeslib.as_array(out, (n * 3,))[:] = 7.0
        return 0

    def cdf_fail(img_p, w, h, row, col):
        calls.append(("cdf", w, h))
        return -1

    mip_c, cdf_c = MIP(mip), CDF(cdf_fail)
    lib = abi.host_lib()
    lib.vimg_host_set_precompute(C.cast(mip_c, C.c_void_p), C.cast(cdf_c, C.c_void_p))
    try:
        s = host.HostScene()
        t = s.add_texture_image(img, abi.WRAP_CLAMP, abi.WRAP_MIRROR)
        assert calls == [("mip", 16, 8, abi.WRAP_CLAMP, abi.WRAP_MIRROR)]
        with pytest.raises(host.HostError, match="CDF builder failed"):
            s.set_background_envmap(t)
        assert calls[-1] == ("cdf", 16, 8)
    finally:
        lib.vimg_host_set_precompute(None, None)
    s2 = host.HostScene()
    s2.add_texture_image(img, abi.WRAP_CLAMP, abi.WRAP_MIRROR)     # host loops again
    assert len(calls) == 2

    lut = host.srgb8_lut()
    x = np.arange(256, dtype=np.float64) / 255.0
    want = np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)
    assert lut[0] == 0 and lut[255] == 1 and np.allclose(lut, want, rtol=1e-6, atol=1e-9)
    vals = rng.integers(0, 256, size=(5, 7, 3), dtype=np.uint8)
    assert np.array_equal(host.srgb8_to_linear(vals), lut[vals])
    nm = host.rgb8_to_normal(vals, scale=0.5)
    v = vals.astype(np.float64) / 127.5 - 1.0
    v[..., :2] *= 0.5
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    assert np.allclose(nm, v, atol=2e-7)


# ------------------------------------------------------------------------------- post chain
def test_tonemap_srgb_quantise_and_png(tmp_path):
    img = np.zeros((2, 4, 3), np.float32)
    img[0, 0] = 0.0
    img[0, 1] = 0.0031308 * 0.5          # linear segment
    img[0, 2] = 0.5
    img[0, 3] = 7.0                      # clamps to 1
    img[1, 0] = np.nan                   # magenta
    out = vimg_amd.tonemap_to_rgb8(img, 0)
    lin = 0.0031308 * 0.5 * 12.92
    assert out[0, 0].tolist() == [0, 0, 0]
    assert out[0, 1].tolist() == [int(255.999 * lin)] * 3
    assert out[0, 2].tolist() == [int(255.999 * (1.055 * 0.5 ** (1 / 2.4) - 0.055))] * 3
    assert out[0, 3].tolist() == [255, 255, 255] and out[1, 0].tolist() == [255, 0, 255]
    for tm in (1, 2, 3):                 # AgX, Reinhard, ACES: finite, in range, monotone in brightness
        ramp = np.linspace(0, 4, 64, dtype=np.float32)[None, :, None].repeat(3, 2)
        o = vimg_amd.tonemap_to_rgb8(ramp, tm)[0, :, 0].astype(int)
        assert np.all(np.diff(o) >= 0) and o[-1] > o[0]
    path = tmp_path / "t.png"
    vimg_amd.write_png(path, out)
    assert np.array_equal(np.asarray(Image.open(path)), out)
    with pytest.raises(host.HostError):
        vimg_amd.tonemap_to_rgb8(img, 9)
