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
    src = '#include <stdio.h>\n#include "vimg_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",' \
          'sizeof(VimgCamera),sizeof(VimgMesh),sizeof(VimgSphere),sizeof(VimgMaterial),sizeof(VimgTexture),' \
          'sizeof(VimgTextureRG),sizeof(VimgBackground),sizeof(VimgBVH),sizeof(VimgScene),' \
          'sizeof(VimgRenderParams),sizeof(VimgRenderStats),sizeof(VimgHipOptions));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "p.c"), "-o",
                        os.path.join(d, "p")], check=True)
        got = [int(v) for v in subprocess.run([os.path.join(d, "p")], capture_output=True,
                                              text=True, check=True).stdout.split()]
    want = [C.sizeof(t) for t in (abi.Camera, abi.Mesh, abi.Sphere, abi.Material, abi.Texture,
                                  abi.TextureRG, abi.Background, abi.BVH, abi.Scene,
                                  abi.RenderParams, abi.RenderStats, abi.HipOptions)]
    assert got == want
    # vimg_hip_options_default fills every field with VIMG_OPT_AUTO, which is what the ctypes mirror starts from
    lib = abi.hip_lib()
    a, b = abi.HipOptions(scheduler=2), abi.HipOptions()
    lib.vimg_hip_options_default(C.byref(a))
    assert bytes(a) == bytes(b) and a.struct_size == C.sizeof(abi.HipOptions) and a.scheduler == abi.OPT_AUTO


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


def _glm_mat_mul(a, b):
    """glm mat4 * mat4 on column lists, float32, terms added left to right."""
    f = np.float32
    r = []
    for j in range(4):
        col = []
        for i in range(4):
            acc = f(a[0][i]) * f(b[j][0])
            for k in (1, 2, 3):
                acc = f(acc + f(a[k][i]) * f(b[j][k]))
            col.append(f(acc))
        r.append(col)
    return r


def _reference_transform(entries):
    """get_transform of the reference (src/scene_loading/json_scene.cpp:67-121) restated in numpy
    float32: scale / rotate (quaternion x y z w, glm::toMat4) / translate, each op PRE-multiplying."""
    f = np.float32
    ident = lambda: [[f(1 if i == j else 0) for i in range(4)] for j in range(4)]
    x = ident()
    for e in entries:
        m = ident()
        if "scale" in e:
            sc = e["scale"] if isinstance(e["scale"], list) else [e["scale"]] * 3
            for k in range(3):
                m[k][k] = f(sc[k])
        elif "rotate" in e:
            qx, qy, qz, qw = (f(v) for v in e["rotate"])
            qxx, qyy, qzz = f(qx * qx), f(qy * qy), f(qz * qz)
            qxz, qxy, qyz = f(qx * qz), f(qx * qy), f(qy * qz)
            qwx, qwy, qwz = f(qw * qx), f(qw * qy), f(qw * qz)
            one, two = f(1), f(2)
            m[0] = [f(one - f(two * f(qyy + qzz))), f(two * f(qxy + qwz)), f(two * f(qxz - qwy)), f(0)]
            m[1] = [f(two * f(qxy - qwz)), f(one - f(two * f(qxx + qzz))), f(two * f(qyz + qwx)), f(0)]
            m[2] = [f(two * f(qxz + qwy)), f(two * f(qyz - qwx)), f(one - f(two * f(qxx + qyy))), f(0)]
        elif "translate" in e:
            m[3] = [f(e["translate"][0]), f(e["translate"][1]), f(e["translate"][2]), f(1)]
        x = _glm_mat_mul(m, x)
    return x


def _xform_point(m, p):
    """xform * vec4(p, 1), then /= w (create_quad_mesh, src/geometry/mesh_loading.cpp:67-77);
    glm mat * vec association (m0 v0 + m1 v1) + (m2 v2 + m3 v3)."""
    f = np.float32
    v = [f(p[0]), f(p[1]), f(p[2]), f(1)]
    r = [f(f(f(m[0][i] * v[0]) + f(m[1][i] * v[1])) + f(f(m[2][i] * v[2]) + f(m[3][i] * v[3]))) for i in range(4)]
    return [f(r[0] / r[3]), f(r[1] / r[3]), f(r[2] / r[3])]


def test_json_loader_against_a_scene_assembled_by_hand():
    """Loader-independent check of BASELINE config 2's input: the tables the JSON loader produces for
    scenes/disney_spheres.json must be, byte for byte, the tables obtained by (a) computing what
    the reference's loader computes from the same numbers here in numpy float32 - transform order,
    quad vertices and winding {0,2,1},{2,0,3}, primitive order, lights = the emissive quad's
    triangles in REVERSE order (add_tri_list_to_scene, mesh_loading.cpp:95-103), one ConstColor
    texture per Principled material, material parameter defaults (json_scene.cpp:289-314) - and
    (b) assembling the scene through the HostScene.add_* API from those numbers."""
    with open(os.path.join(scenes.SCENES, "disney_spheres.json")) as f:
        d = json.load(f)
    ref = scenes.json_scene("disney_spheres.json")
    v = ref.view.contents

    # ---- (a) expected tables, computed here
    quad_local = [(-1, -1, 0), (-1, 1, 0), (1, 1, 0), (1, -1, 0)]
    exp_vertices, exp_tri, exp_tri_mesh, exp_prims, exp_lights, exp_spheres = [], [], [], [], [], []
    mat_index = {m["name"]: i for i, m in enumerate(d["materials"])}
    mat_type = {m["name"]: m["type"] for m in d["materials"]}
    n_mesh = 0
    for surf in d["surfaces"]:
        if surf["type"] == "quad":
            m = _reference_transform(surf.get("transform", []))
            exp_vertices += [_xform_point(m, p) for p in quad_local]
            first_prim = len(exp_prims)
            for t in ((0, 2, 1), (2, 0, 3)):
                exp_tri.append(t)
                exp_tri_mesh.append(n_mesh)
                exp_prims.append((abi.PRIM_TRIANGLE, len(exp_tri) - 1))
            if mat_type[surf["mat_name"]] == "diffuse_light":
                exp_lights += [(abi.LIGHT_PRIM, first_prim + 1), (abi.LIGHT_PRIM, first_prim)]
            n_mesh += 1
        else:
            exp_spheres.append((surf["center"], surf.get("radius", 1.0), mat_index[surf["mat_name"]]))
            exp_prims.append((abi.PRIM_SPHERE, len(exp_spheres) - 1))
    got_vertices = np.ctypeslib.as_array(v.vertices, (v.num_vertices, 3))
    assert np.array_equal(got_vertices.view(np.uint32), np.asarray(exp_vertices, np.float32).view(np.uint32))
    assert np.ctypeslib.as_array(v.tri_indices, (v.num_tris, 3)).tolist() == [list(t) for t in exp_tri]
    assert np.ctypeslib.as_array(v.tri_mesh, (v.num_tris,)).tolist() == exp_tri_mesh
    assert [(v.prims[i].type, v.prims[i].index) for i in range(v.num_prims)] == exp_prims
    assert [(v.lights[i].type, v.lights[i].prim) for i in range(v.num_lights)] == exp_lights
    assert v.num_spheres == len(exp_spheres)
    for i, (c, r, m) in enumerate(exp_spheres):
        sp = v.spheres[i]
        assert (list(sp.center), sp.radius, sp.material) == ([np.float32(x) for x in c], np.float32(r), m)
    uv = np.ctypeslib.as_array(v.uvs, (v.num_uvs, 2))
    assert uv.tolist() == [[0, 0], [0, 1], [1, 1], [1, 0]] * n_mesh
    assert v.background.type == abi.BG_CONST and list(v.background.col) == [0, 0, 0]      # quirk Q3
    assert (v.camera.res_x, v.camera.res_y, v.camera.vfov_deg, v.camera.aperture_radius) == (1800, 800, float(d["camera"]["vfov"]), 0.0)

    # ---- (b) the same scene through the add_* API (meshes from the vertices computed above)
    s = host.HostScene()
    cam = d["camera"]
    s.set_camera(cam["transform"]["from"], cam["transform"]["at"], cam["transform"]["up"], cam["vfov"],
                 cam["resolution"])
    s.set_render_defaults(d["integrator"]["type"], d["sampler"]["samples"], d["sampler"]["depth"])
    for m in d["materials"]:
        if m["type"] == "lambertian":
            s.add_material("lambertian", tex=s.add_texture_const(m["albedo"]))
        elif m["type"] == "diffuse_light":
            s.add_material("diffuse_light", emit=m["albedo"])
        else:
            s.add_material("principled", tex=s.add_texture_const(m["base_color"]),
                           roughness=m.get("roughness", 0.5), anisotropic=m.get("anisotropic", 0.0),
                           eta=m.get("eta", 1.5), subsurface=m.get("subsurface", 0.0),
                           metallic=m.get("metallic", 0.0), spec_trans=m.get("spec_trans", 0.0),
                           specular=m.get("specular", 0.5), spec_tint=m.get("spec_tint", 0.0),
                           sheen=m.get("sheen", 0.0), sheen_tint=m.get("sheen_tint", 0.5),
                           clearcoat=m.get("clearcoat", 0.0), clearcoat_gloss=m.get("clearcoat_gloss", 1.0))
    k = 0
    for surf in d["surfaces"]:
        if surf["type"] == "quad":
            s.add_mesh(np.asarray(exp_vertices[4 * k:4 * k + 4], np.float32), [[0, 2, 1], [2, 0, 3]],
                       mat_index[surf["mat_name"]], uv_sets=[[[0, 0], [0, 1], [1, 1], [1, 0]]], color_uv=0)
            k += 1
        else:
            s.add_sphere(surf["center"], surf.get("radius", 1.0), mat_index[surf["mat_name"]])
    s.set_background_const((0, 0, 0), add_to_lights=False)
    s.build_bvh(abi.BVH_SWEEP)
    w = s.view.contents

    def table(ptr, n, ctype):
        return C.string_at(ptr, n * C.sizeof(ctype)) if n else b""
    for name, cnt, ctype in (("prims", "num_prims", abi.Prim), ("meshes", "num_meshes", abi.Mesh),
                             ("spheres", "num_spheres", abi.Sphere),
                             ("textures", "num_textures", abi.Texture), ("lights", "num_lights", abi.Light)):
        assert getattr(v, cnt) == getattr(w, cnt), name
        assert table(getattr(v, name), getattr(v, cnt), ctype) == table(getattr(w, name), getattr(w, cnt), ctype), name
    # materials: the fields their type reads (the loader leaves the others zero, the API at its defaults)
    used = {abi.MAT_LAMBERTIAN: ("tex",), abi.MAT_DIFFUSE_LIGHT: ("emit",),
            abi.MAT_PRINCIPLED: ("tex", "mr_tex", "normal_map", "metallic_factor", "roughness_factor",
                                 "specular_transmission", "subsurface", "specular", "specular_tint",
                                 "anisotropic", "sheen", "sheen_tint", "clearcoat", "clearcoat_gloss", "eta")}
    assert v.num_materials == w.num_materials == len(d["materials"])
    for i in range(v.num_materials):
        a, b = v.materials[i], w.materials[i]
        assert a.type == b.type
        for fld in used[a.type]:
            fa, fb = getattr(a, fld), getattr(b, fld)
            assert (list(fa) == list(fb)) if fld == "emit" else (fa == fb), (i, fld)
    assert v.num_vertices == w.num_vertices and v.num_tris == w.num_tris and v.num_uvs == w.num_uvs
    assert C.string_at(v.vertices, v.num_vertices * 12) == C.string_at(w.vertices, w.num_vertices * 12)
    assert C.string_at(v.tri_indices, v.num_tris * 12) == C.string_at(w.tri_indices, w.num_tris * 12)
    assert C.string_at(v.uvs, v.num_uvs * 8) == C.string_at(w.uvs, w.num_uvs * 8)
    assert bytes(v.camera) == bytes(w.camera) and bytes(v.background) == bytes(w.background)
    for a, b in zip(ref.bvh_arrays(), s.bvh_arrays()):
        assert np.array_equal(a, b)
    pr, ps = ref.default_params(), s.default_params()
    assert (pr.integrator, pr.samples, pr.depth) == (ps.integrator, ps.samples, ps.depth) == (abi.INTEGRATOR_MIS, 512, 0xFFFFFFFF)


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
def test_json_mesh_surface_loads_obj_positions_and_splits_quads_like_tinyobj(tmp_path):
    """"type": "mesh" (reference src/scene_loading/json_scene.cpp:366-385 + load_from_obj,
    src/geometry/mesh_loading.cpp:21-65): positions only, path relative to the scene file,
    1-based / negative indices, v/vt/vn tokens; the transform is applied to positions.  Quads are
    split as the reference's bundled tinyobjloader splits them (include/tiny_obj_loader.h:1488-1583):
    along the shorter diagonal of the UNTRANSFORMED positions, [0,1,2],[0,2,3] only when
    |v2-v0|^2 < |v3-v1|^2 and [0,1,3],[1,2,3] otherwise - so every square (a tie) takes the second
    form; polygons with more than four vertices are refused."""
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
v 0 0 2
v 3 0 2
v 1 1 2
v 0 1 2
f 9 10 11 12
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
    assert v.num_meshes == 2 and v.num_prims == 14 + 2
    m = v.meshes[0]
    assert (m.num_vertices, m.has_normals, m.num_uv_sets, m.color_tex_uv) == (12, 0, 0, abi.NO_UV)
    verts = np.ctypeslib.as_array(v.vertices, (v.num_vertices, 3))[:8]
    want = np.array([[-.5, -.5, -.5], [.5, -.5, -.5], [.5, .5, -.5], [-.5, .5, -.5], [-.5, -.5, .5],
                     [.5, -.5, .5], [.5, .5, .5], [-.5, .5, .5]], dtype=np.float32)
    want = (want + np.float32([0, 0.5, 0])) * np.float32([1, 9, 4])     # translate, then scale
    assert np.array_equal(verts, want)
    tri = np.ctypeslib.as_array(v.tri_indices, (v.num_tris, 3))[:14]
    assert tri.tolist() == [[0, 3, 1], [3, 2, 1], [4, 5, 7], [5, 6, 7], [0, 1, 4], [1, 5, 4],   # squares: ties
                            [1, 2, 5], [2, 6, 5], [2, 3, 6], [3, 7, 6], [3, 0, 4], [3, 4, 7],
                            [8, 9, 10], [8, 10, 11]]               # |v2-v0|^2 = 2 < |v3-v1|^2 = 10
    # errors: polygon with more than four vertices, missing file, face before its vertices
    (tmp_path / "scenes" / "penta.obj").write_text("v 0 0 0\nv 1 0 0\nv 2 1 0\nv 1 2 0\nv 0 1 0\nf 1 2 3 4 5\n")
    scene["surfaces"][0]["filename"] = "penta.obj"
    path.write_text(json.dumps(scene))
    with pytest.raises(host.HostError, match="more than 4 vertices"):
        host.HostScene.from_json(str(path))
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


@pytest.mark.parametrize("w,h,wrap_u,wrap_v", [(32, 32, abi.WRAP_REPEAT, abi.WRAP_REPEAT), (40, 24, abi.WRAP_CLAMP, abi.WRAP_MIRROR),
                                              (17, 64, abi.WRAP_MIRROR, abi.WRAP_CLAMP)])
def test_mip_chain_is_byte_identical_to_an_independent_restatement(w, h, wrap_u, wrap_v):
    """The host library's mip chain (and through the byte-exactness tests of test_gpu_parity, the
    GPU pre-step's) against tests/prestep_ref.py, a numpy restatement of the reference's loop
    (src/image_texture.cpp:60-158: level count, 8-tap filter, bilinear taps, wrap modes, clamp of
    negative lobes) - not against the product's own code."""
    import prestep_ref as R
    rng = np.random.default_rng(w * 131 + h)
    img = (rng.random((h, w, 3), dtype=np.float32) * 1.5).astype(np.float32)
    s = host.HostScene()
    t = s.add_texture_image(img, wrap_u, wrap_v)
    s.add_material("lambertian", tex=t)
    s.add_sphere((0, 0, 0), 1.0, 0)
    s.build_bvh()
    v = s.view.contents
    tex = v.textures[t]
    want = R.mip_chain(img, wrap_u, wrap_v)
    assert tex.num_levels == len(want)
    texels = np.ctypeslib.as_array(v.texels, (v.num_texels, 3))
    for l, lvl in enumerate(want):
        lh, lw = lvl.shape[:2]
        got = texels[tex.level_offset[l]:tex.level_offset[l] + lw * lh].reshape(lh, lw, 3)
        assert np.array_equal(got.view(np.uint32), lvl.view(np.uint32)), (l, np.abs(got - lvl).max())


def test_env_cdfs_are_byte_identical_to_an_independent_restatement():
    """ArraySampling2D / ArraySampling1D (include/rng/sampling.h:107-197) restated in numpy
    (tests/prestep_ref.py): running float sums, normalisation by the row integral, sin(pi v) in
    double - against the host library's tables of an env-map scene."""
    import prestep_ref as R
    rng = np.random.default_rng(77)
    img = (rng.random((24, 48, 3), dtype=np.float32) ** 3 * 20).astype(np.float32)
    img[5, :, :] = 0          # a row with zero integral: uniform conditional
    s = host.HostScene()
    t = s.add_texture_image(img)
    s.set_background_envmap(t)
    s.add_material("lambertian", tex=s.add_texture_const((0.5, 0.5, 0.5)))
    s.add_sphere((0, 0, 0), 1.0, 0)
    s.build_bvh()
    v = s.view.contents
    cdf = np.ctypeslib.as_array(v.cdf_pool, (v.num_cdf,))
    rows = cdf[v.background.row_cdf_offset:v.background.row_cdf_offset + 25]
    cols = cdf[v.background.col_cdf_offset:v.background.col_cdf_offset + 24 * 49].reshape(24, 49)
    want_rows, want_cols = R.env_cdfs(img)
    assert np.array_equal(cols.view(np.uint32), want_cols.view(np.uint32))
    assert np.array_equal(rows.view(np.uint32), want_rows.view(np.uint32))
    assert np.array_equal(cols[5], np.arange(49, dtype=np.float32) / np.float32(48))


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


def test_bench_roofline_refuses_a_stale_pmc_pass(tmp_path):
    """bench.py divides the instruction counts of a committed rocprofv3 PMC pass by the time it
    measures live.  The pass is only used when it was taken on this workload, this kernel and this
    build of the library (sha256); anything else flips the line to "pmc": "stale" (with the reason) and to the
    labelled algorithmic figure; a shard's counts scale with its rays."""
    import importlib
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    wl, kern, sha = "disney_spheres.json, mis integrator, 512 spp, 1800x800", "render_cu_kernel<false>", "ab" * 32
    tie = {"workload": wl, "kernel_name": kern, "library_sha256": sha, "rays_per_launch": 4.0e9}
    (tmp_path / "valu.json").write_text(json.dumps(dict(tie, valu_wave_insts_per_launch=2.0e11, valu_lane_utilization=0.5)))
    (tmp_path / "traffic.json").write_text(json.dumps(dict(tie, hbm_bytes_per_launch=4.0e11)))
    fresh = bench.build_roofline(300.0, 2.0e12, 4.0e9, kern, wl, sha, profiles_dir=str(tmp_path))
    assert fresh["pmc"] == "fresh" and fresh["unit"] == "Tlane-op/s" and fresh["traffic"] == int(4.0e11)
    want = 2.0e11 / 0.3 * 64 * 0.5 / bench.VALU_LANE_PEAK
    assert abs(fresh["frac"] - want) < 1e-3 and abs(fresh["valu"]["issue_frac"] - 2.0e11 / 0.3 / bench.VALU_ISSUE_PEAK) < 1e-3
    # an eighth of the frame: an eighth of the instructions and of the traffic
    shard = bench.build_roofline(100.0, 2.5e11, 5.0e8, kern, wl, sha, profiles_dir=str(tmp_path))
    assert shard["pmc"] == "fresh" and abs(shard["valu"]["wave_insts_per_launch"] - 2.5e10) < 1 and shard["traffic"] == int(5.0e10)
    for other in (dict(lib_sha256="cd" * 32), dict(kernel_name="render_pool4_kernel<false,group>"), dict(workload=wl.replace("512", "64"))):
        args = dict(kernel_name=kern, workload=wl, lib_sha256=sha)
        args.update(other)
        stale = bench.build_roofline(300.0, 2.0e12, 4.0e9, args["kernel_name"], args["workload"], args["lib_sha256"],
                                     profiles_dir=str(tmp_path))
        assert stale["pmc"] == "stale" and stale["pmc_detail"]["valu"].startswith("stale: "), stale
        assert stale["unit"] == "GB/s" and stale["traffic"] is None and "valu" not in stale
    absent = bench.build_roofline(300.0, 2.0e12, 4.0e9, kern, wl, sha, profiles_dir=str(tmp_path / "none"))
    assert absent["pmc"] == "absent" and absent["unit"] == "GB/s"
    # the fingerprint is the sha256 of the file
    import hashlib
    f = tmp_path / "lib.so"
    f.write_bytes(b"code object")
    assert bench.library_fingerprint(str(f)) == hashlib.sha256(b"code object").hexdigest()
    assert bench.library_fingerprint(str(tmp_path / "missing.so")) is None


def test_no_called_function_reads_the_kernel_argument_segment():
    """Guard of a fault that happened once (round 2, gpurun_out/s1/t.log: "Fatal Python error: Aborted" in
    render_to_host): under code object v5 a non-inlined device function gets no
    kernel-argument pointer - __builtin_amdgcn_kernarg_segment_ptr() folds to null there and the first
    scene access faults.  Rules held by the sources: the builtin appears in ONE place (cu_kargs of
    render_cu_kernel.h); everything that calls cu_kargs is force-inlined into the kernel (VD) or is the
    kernel; that header defines no __noinline__ function; and the __noinline__ vertex stage of the
    development build's render_pool4_kernel takes the address of its argument block as (k_lo, k_hi)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "v-img_amd", "csrc")
    users = {}
    for name in sorted(os.listdir(csrc)):
        text = open(os.path.join(csrc, name)).read()
        code = re.sub(r"//[^\n]*", "", text)
        if "__builtin_amdgcn_kernarg_segment_ptr" in code:
            users[name] = code.count("__builtin_amdgcn_kernarg_segment_ptr")
    assert users == {"render_cu_kernel.h": 1}, users
    cu = re.sub(r"//[^\n]*", "", open(os.path.join(csrc, "render_cu_kernel.h")).read())
    assert "__noinline__" not in cu
    # every function of the header whose body calls cu_kargs() is VD (= __device__ __forceinline__) or __global__
    heads = [(m.start(), m.group(0)) for m in re.finditer(r"^(?:VD|__global__)[^\n;{]*\n?[^\n;{]*\{", cu, flags=re.M)]
    for m in re.finditer(r"cu_kargs\(\)", cu):
        before = [h for h in heads if h[0] < m.start()]
        assert before, "cu_kargs() outside a function"
        assert before[-1][1].startswith(("VD", "__global__")), before[-1][1]
    assert len(list(re.finditer(r"cu_kargs\(\)", cu))) >= 4      # (definition, two stages, the kernel)
    # the development build's non-inlined callee: block address through registers, never kernel arguments
    p4 = re.sub(r"//[^\n]*", "", open(os.path.join(csrc, "render_pool4_kernel.h")).read())
    sigs = re.findall(r"__noinline__\s+\w+\s+(\w+)\s*\(([^)]*)\)", p4)
    assert sigs and all(args.strip().startswith("uint32_t k_lo, uint32_t k_hi") for _, args in sigs), sigs
    assert all("DScene" not in args and "RenderArgs" not in args for _, args in sigs)


def _kernel_notes(obj_path):
    """{kernel symbol: {private_segment_fixed_size, vgpr_spill_count, vgpr_count}} of the gfx950 code
    object bundled in one compiled .hip unit (objcopy + clang-offload-bundler + llvm-readelf)."""
    import re, shutil, subprocess, tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [shutil.which("objcopy"), os.path.join(llvm, "clang-offload-bundler"), os.path.join(llvm, "llvm-readelf")]
    if not all(t and os.path.exists(t) for t in tools) or not os.path.exists(obj_path):
        return None
    with tempfile.TemporaryDirectory() as tmp:
        fat, elf = os.path.join(tmp, "unit.fatbin"), os.path.join(tmp, "unit.elf")
        subprocess.run([tools[0], "-O", "binary", "--only-section=.hip_fatbin", obj_path, fat], check=True)
        subprocess.run([tools[1], "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--output={elf}"], check=True, capture_output=True)
        notes = subprocess.run([tools[2], "--notes", elf], check=True, capture_output=True, text=True).stdout
    out, fields = {}, {}
    for line in notes.splitlines():
        m = re.match(r"\s+\.(name|private_segment_fixed_size|vgpr_spill_count|vgpr_count):\s+(\S+)", line)
        if m:
            fields[m.group(1)] = m.group(2)
        if line.strip().startswith(".wavefront_size") and "name" in fields:
            out[fields["name"]] = {k: int(v) for k, v in fields.items() if k != "name"}
            fields = {}
    return out


def test_the_untextured_render_kernels_use_no_scratch():
    """VERDICT r2 item 2 ("Scratch_Size 0 for the default C2 kernel ... vgpr_spill_count 0 in the ISA notes"):
    the builds config 2 is rendered with - render_cu_kernel<false, deep or not>, rays queued late
    (k_cu.hip) and early (k_cu_early.hip) - keep everything in their 128 registers.  Read from the code
    objects `make` left under build/hip (what the last 48 B were)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = 0
    for unit in ("k_cu.o", "k_cu_early.o"):
        notes = _kernel_notes(os.path.join(root, "build", "hip", unit))
        if notes is None:
            pytest.skip("no build/hip objects or no binutils / llvm tools here")
        plain = {k: v for k, v in notes.items() if "render_cu_kernelILb0E" in k}      # TEX = false
        assert len(plain) == 2, sorted(notes)
        for name, n in plain.items():
            assert n["private_segment_fixed_size"] == 0 and n["vgpr_spill_count"] == 0 and n["vgpr_count"] <= 128, (name, n)
            seen += 1
    assert seen == 4
