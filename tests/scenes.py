"""Scene builders shared by the CPU and GPU tests (inputs only; no reference code)."""
import os

import numpy as np

import vimg_amd
from vimg_amd import abi

SCENES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes")


def json_scene(name, res=None, bvh=abi.BVH_SWEEP):
    """Load one of the reference's JSON scenes, optionally at a smaller resolution."""
    path = os.path.join(SCENES, name)
    if res is None:
        return vimg_amd.HostScene.from_json(path, bvh=bvh)
    import json
    with open(path) as f:
        d = json.load(f)
    d["camera"]["resolution"] = [int(res[0]), int(res[1])]
    return vimg_amd.HostScene.from_json_text(json.dumps(d), bvh=bvh)


def odyssey_without_monolith(res=None):
    """The reference's third known-answer scene, scenes/MIS_light_tests/odyssey_mis.json (a depth-1
    floor under a large QUAD light, the only analytic pin of triangle-light sampling), without its
    monolith: that surface is the mesh ../../assets/cube.obj, which the reference tree does not hold
    (README.md:58).  The floor between the glowing wall (x = -12) and the monolith (x >= -1 whatever
    the cube's vertices within [-1, 1]^3 are) sees the whole wall with or without it, and at depth 1
    nothing the monolith reflects arrives there: those pixels are a known answer."""
    import json
    with open(os.path.join(SCENES, "MIS_light_tests", "odyssey_mis.json")) as f:
        d = json.load(f)
    kept = [sf for sf in d["surfaces"] if sf["type"] != "mesh"]
    assert len(kept) == 2 and len(d["surfaces"]) == 3
    d["surfaces"] = kept
    if res is not None:
        d["camera"]["resolution"] = [int(res[0]), int(res[1])]
    return vimg_amd.HostScene.from_json_text(json.dumps(d))


def _grid_mesh(n, size, height_fn, uv_scale=1.0):
    xs = np.linspace(-size, size, n + 1, dtype=np.float32)
    gx, gz = np.meshgrid(xs, xs, indexing="ij")
    gy = height_fn(gx, gz).astype(np.float32)
    verts = np.stack([gx, gy, gz], -1).reshape(-1, 3)
    uv = np.stack([(gx / (2 * size) + 0.5) * uv_scale, (gz / (2 * size) + 0.5) * uv_scale],
                  -1).reshape(-1, 2).astype(np.float32)
    idx = []
    for i in range(n):
        for j in range(n):
            a, b = i * (n + 1) + j, i * (n + 1) + j + 1
            c, d = (i + 1) * (n + 1) + j, (i + 1) * (n + 1) + j + 1
            idx += [[a, b, d], [a, d, c]]
    idx = np.asarray(idx, dtype=np.uint32)
    # smooth normals from the height field gradient
    eps = 1e-3
    dydx = (height_fn(gx + eps, gz) - height_fn(gx - eps, gz)) / (2 * eps)
    dydz = (height_fn(gx, gz + eps) - height_fn(gx, gz - eps)) / (2 * eps)
    nrm = np.stack([-dydx, np.ones_like(dydx), -dydz], -1).reshape(-1, 3)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return verts, idx, nrm.astype(np.float32), uv


def feature_scene(res=(96, 64), seed=7, envmap=True, lens=True, rg_shape=(8, 8)):
    """Procedural scene exercising every feature on the path that the JSON format cannot
    express: mesh with interpolated normals and two uv sets, mip-mapped image texture, normal
    map, metallic-roughness map, checkerboard, Principled + Lambertian + glass + emissive quad,
    env-map importance sampling (or a constant emissive background), thin lens."""
    rng = np.random.default_rng(seed)
    s = vimg_amd.HostScene()
    s.set_camera((0.0, 1.6, 4.2), (0.0, 0.3, 0.0), (0, 1, 0), 38.0, res,
                 aperture_radius=0.03 if lens else 0.0, focal_dist=4.3)
    s.set_render_defaults("mis", 8, 12)

    # textures
    img = rng.random((32, 32, 3), dtype=np.float32) * 0.8 + 0.1
    img[8:24, 8:24] *= 0.3
    t_img = s.add_texture_image(img, abi.WRAP_REPEAT, abi.WRAP_MIRROR)
    nm = rng.normal(0, 0.25, (16, 16, 3)).astype(np.float32)
    nm[..., 2] = 1.0
    nm /= np.linalg.norm(nm, axis=2, keepdims=True)
    t_nm = s.add_texture_image(nm, abi.WRAP_REPEAT, abi.WRAP_REPEAT)
    rg = rng.random((rg_shape[1], rg_shape[0], 2), dtype=np.float32) * 0.8 + 0.1    # (width, height) of the map
    t_rg = s.add_texture_rg(rg, abi.WRAP_REPEAT, abi.WRAP_CLAMP)
    t_white = s.add_texture_const((0.73, 0.73, 0.73))
    t_checker = s.add_texture_checker(8, 8, (0.8, 0.8, 0.8), (0.15, 0.15, 0.2))
    t_blue = s.add_texture_const((0.2, 0.3, 0.8))

    m_floor = s.add_material("lambertian", tex=t_checker)
    m_wall = s.add_material("lambertian", tex=t_white)
    m_light = s.add_material("diffuse_light", emit=(12.0, 11.0, 9.0))
    m_tex = s.add_material("principled", tex=t_img, mr_tex=t_rg, normal_map=t_nm, metallic=0.9,
                           roughness=0.8, specular=0.5, clearcoat=0.6, clearcoat_gloss=0.4,
                           sheen=0.3, anisotropic=0.2, subsurface=0.2)
    m_glass = s.add_material("principled", tex=t_blue, metallic=0.0, roughness=0.15,
                             spec_trans=1.0, eta=1.45)
    m_diel = s.add_material("dielectric", ior=1.5)
    m_lamb_img = s.add_material("lambertian", tex=t_img)

    def xf(scale, rot_x_deg, trans):
        a = np.deg2rad(rot_x_deg)
        r = np.array([[1, 0, 0, 0], [0, np.cos(a), -np.sin(a), 0], [0, np.sin(a), np.cos(a), 0],
                      [0, 0, 0, 1]], dtype=np.float32)
        sc = np.diag([scale[0], scale[1], scale[2], 1]).astype(np.float32)
        t = np.eye(4, dtype=np.float32)
        t[:3, 3] = trans
        return (t @ r @ sc).T.reshape(16)   # column-major

    s.add_quad(xf((4, 4, 1), -90, (0, 0, 0)), m_floor)            # floor, normal +y
    s.add_quad(xf((4, 2.5, 1), 0, (0, 2.5, -3.0)), m_wall)        # back wall
    s.add_quad(xf((0.8, 0.6, 1), 90, (0.3, 3.2, 0.2)), m_light)   # light, facing down

    verts, idx, nrm, uv = _grid_mesh(10, 0.9, lambda x, z: 0.35 + 0.12 * np.sin(3 * x) * np.cos(2.5 * z))
    verts = verts + np.array([-1.1, 0.0, 0.3], dtype=np.float32)
    uv2 = (uv * 2.0).astype(np.float32)
    s.add_mesh(verts, idx, m_tex, normals=nrm, uv_sets=[uv, uv2], color_uv=0, normal_uv=1, mr_uv=0)
    v2, i2, n2, uv_b = _grid_mesh(4, 0.5, lambda x, z: 0.9 + 0.0 * x)
    v2 = v2 + np.array([1.4, 0.0, -1.2], dtype=np.float32)
    s.add_mesh(v2, i2, m_lamb_img, normals=None, uv_sets=[uv_b], color_uv=0)

    s.add_sphere((0.9, 0.45, 0.7), 0.45, m_glass)
    s.add_sphere((-0.1, 0.3, 1.3), 0.3, m_diel)
    s.add_sphere((0.2, 0.55, -0.9), 0.55, m_tex)

    if envmap:
        h, w = 16, 32
        yy = (np.arange(h, dtype=np.float32)[:, None] + 0.5) / h
        xx = (np.arange(w, dtype=np.float32)[None, :] + 0.5) / w
        sky = np.stack([0.3 + 0.4 * (1 - yy) + 0 * xx, 0.4 + 0.4 * (1 - yy) + 0 * xx,
                        0.6 + 0.4 * (1 - yy) + 0 * xx], -1).astype(np.float32)
        sun = np.exp(-(((xx - 0.3) / 0.06) ** 2 + ((yy - 0.25) / 0.08) ** 2)).astype(np.float32)
        env = sky + 25.0 * sun[..., None]
        t_env = s.add_texture_image(env, abi.WRAP_CLAMP, abi.WRAP_CLAMP)
        s.set_background_envmap(t_env, radiance_scale=0.8)
    else:
        s.set_background_const((0.35, 0.4, 0.5), add_to_lights=True)
    s.build_bvh(abi.BVH_SWEEP)
    return s


def big_mesh_scene(res=(128, 96), n=48, seed=3):
    """A few thousand triangles (deeper BVH, global-memory nodes beyond the LDS copy)."""
    s = vimg_amd.HostScene()
    s.set_camera((0.0, 2.2, 4.5), (0.0, 0.2, 0.0), (0, 1, 0), 40.0, res)
    s.set_render_defaults("mis", 8, 16)
    t_a = s.add_texture_const((0.7, 0.6, 0.5))
    t_c = s.add_texture_checker(16, 16, (0.9, 0.9, 0.9), (0.2, 0.2, 0.25))
    m_a = s.add_material("principled", tex=t_a, metallic=0.3, roughness=0.4, clearcoat=0.5)
    m_c = s.add_material("lambertian", tex=t_c)
    m_l = s.add_material("diffuse_light", emit=(20.0, 20.0, 20.0))
    verts, idx, nrm, uv = _grid_mesh(n, 2.5, lambda x, z: 0.25 * np.sin(2.2 * x) * np.sin(1.7 * z))
    s.add_mesh(verts, idx, m_c, normals=nrm, uv_sets=[uv], color_uv=0)
    v2, i2, n2, uv2 = _grid_mesh(n // 2, 0.8, lambda x, z: 1.0 + 0.3 * np.cos(3 * x + z))
    s.add_mesh(v2, i2, m_a, normals=n2, uv_sets=[uv2], color_uv=0)
    s.add_sphere((1.5, 0.8, 0.5), 0.5, m_a)
    lv = np.array([[-0.7, 3.0, -0.7], [0.7, 3.0, -0.7], [0.7, 3.0, 0.7], [-0.7, 3.0, 0.7]],
                  dtype=np.float32)
    s.add_mesh(lv, np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32), m_l)
    s.build_bvh(abi.BVH_SWEEP)
    return s


# --------------------------------------------------------------------------------------------
# Synthetic stand-ins for BASELINE configs 3-5 (their assets are not in the reference tree):
# same feature mix, procedural data with fixed seeds.
def _grid_mesh_fast(n, size, height_fn, center=(0, 0, 0), uv_scale=1.0):
    xs = np.linspace(-size, size, n + 1, dtype=np.float32)
    gx, gz = np.meshgrid(xs, xs, indexing="ij")
    gy = height_fn(gx, gz).astype(np.float32)
    verts = np.stack([gx, gy, gz], -1).reshape(-1, 3) + np.asarray(center, np.float32)
    uv = np.stack([(gx / (2 * size) + 0.5) * uv_scale, (gz / (2 * size) + 0.5) * uv_scale],
                  -1).reshape(-1, 2).astype(np.float32)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a = (i * (n + 1) + j).ravel()
    b, c, d = a + 1, a + n + 1, a + n + 2
    idx = np.stack([np.stack([a, b, d], 1), np.stack([a, d, c], 1)], 1).reshape(-1, 3).astype(np.uint32)
    eps = 1e-3
    dydx = (height_fn(gx + eps, gz) - height_fn(gx - eps, gz)) / (2 * eps)
    dydz = (height_fn(gx, gz + eps) - height_fn(gx, gz - eps)) / (2 * eps)
    nrm = np.stack([-dydx, np.ones_like(dydx), -dydz], -1).reshape(-1, 3)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return verts.astype(np.float32), idx, nrm.astype(np.float32), uv


def _uv_sphere(n_lat, n_lon, radius, center, displace=None):
    th = np.linspace(0, np.pi, n_lat + 1, dtype=np.float64)
    ph = np.linspace(0, 2 * np.pi, n_lon + 1, dtype=np.float64)
    T, P = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], -1)
    r = radius * (1.0 + (displace(d) if displace else 0.0))
    verts = (d * r[..., None]).reshape(-1, 3) + np.asarray(center)
    nrm = d.reshape(-1, 3)
    uv = np.stack([P / (2 * np.pi), T / np.pi], -1).reshape(-1, 2)
    i, j = np.meshgrid(np.arange(n_lat), np.arange(n_lon), indexing="ij")
    a = (i * (n_lon + 1) + j).ravel()
    b, c, e = a + 1, a + n_lon + 1, a + n_lon + 2
    idx = np.stack([np.stack([a, b, c], 1), np.stack([b, e, c], 1)], 1).reshape(-1, 3).astype(np.uint32)
    return verts.astype(np.float32), idx, nrm.astype(np.float32), uv.astype(np.float32)


def _sky(h, w, sun=(0.3, 0.25), sun_gain=40.0):
    yy = (np.arange(h, dtype=np.float32)[:, None] + 0.5) / h
    xx = (np.arange(w, dtype=np.float32)[None, :] + 0.5) / w
    sky = np.stack([0.25 + 0.5 * (1 - yy) + 0 * xx, 0.35 + 0.5 * (1 - yy) + 0 * xx,
                    0.55 + 0.45 * (1 - yy) + 0 * xx], -1).astype(np.float32)
    s = np.exp(-(((xx - sun[0]) / 0.03) ** 2 + ((yy - sun[1]) / 0.04) ** 2)).astype(np.float32)
    return sky + sun_gain * s[..., None]


def config3_scene(res=(1366, 1024), k=5, env=(1024, 512)):
    """Disney-BSDF array: k x k analytic spheres, Principled parameters swept over the grid, on
    a checkerboard floor under a procedural lat-long env map (mis, importance sampled)."""
    s = vimg_amd.HostScene()
    s.set_camera((0.0, 6.5, 11.0), (0.0, 0.6, 0.0), (0, 1, 0), 40.0, res)
    s.set_render_defaults("mis", 512, 0xFFFFFFFF)
    t_floor = s.add_texture_checker(8, 8, (0.8, 0.8, 0.8), (0.2, 0.2, 0.2))
    m_floor = s.add_material("lambertian", tex=t_floor)
    q = np.diag([8.0, 8.0, 1.0, 1.0]).astype(np.float32)
    rot = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]], np.float32)   # +z -> +y
    s.add_quad((rot @ q).T.reshape(16), m_floor)
    for i in range(k):
        for j in range(k):
            u, v = i / (k - 1), j / (k - 1)
            t = s.add_texture_const((0.15 + 0.7 * u, 0.25, 0.85 - 0.6 * v))
            m = s.add_material("principled", tex=t, metallic=u, roughness=0.05 + 0.9 * v,
                               spec_trans=1.0 if (i + j) % 5 == 0 else 0.0,
                               clearcoat=1.0 if (i + j) % 3 == 0 else 0.0, clearcoat_gloss=0.7,
                               sheen=v, anisotropic=0.5 * u, subsurface=0.5 * (1 - u), eta=1.5)
            s.add_sphere(((i - (k - 1) / 2) * 2.4, 0.9, (j - (k - 1) / 2) * 2.4), 0.9, m)
    t_env = s.add_texture_image(_sky(env[1], env[0]), abi.WRAP_CLAMP, abi.WRAP_CLAMP)
    s.set_background_envmap(t_env, radiance_scale=1.0)
    s.build_bvh(abi.BVH_SWEEP)
    return s


def config4_scene(res=(1366, 768), n_lat=280, env=(2048, 1024), seed=0xC4):
    """Two displaced spheres (~2 x 2*n_lat*2*n_lat triangles) with smooth normals, a normal map
    on one, Principled metallic-roughness, procedural HDRI with importance sampling, thin lens."""
    rng = np.random.default_rng(seed)
    s = vimg_amd.HostScene()
    s.set_camera((0.0, 1.4, 6.0), (0.0, 1.0, 0.0), (0, 1, 0), 35.0, res, aperture_radius=0.06,
                 focal_dist=5.6)
    s.set_render_defaults("mis", 512, 0xFFFFFFFF)
    coef = rng.normal(size=(6, 3))

    def disp(d):
        out = 0
        for c in coef:
            out = out + 0.03 * np.sin(4 * (d @ c))
        return out
    nm = rng.normal(0, 0.2, (256, 256, 3)).astype(np.float32)
    nm[..., 2] = 1.0
    nm /= np.linalg.norm(nm, axis=2, keepdims=True)
    t_nm = s.add_texture_image(nm)
    t_a = s.add_texture_const((0.9, 0.6, 0.3))
    t_b = s.add_texture_const((0.6, 0.65, 0.7))
    t_g = s.add_texture_checker(16, 16, (0.7, 0.7, 0.7), (0.3, 0.3, 0.3))
    m_a = s.add_material("principled", tex=t_a, normal_map=t_nm, metallic=1.0, roughness=0.35)
    m_b = s.add_material("principled", tex=t_b, metallic=0.0, roughness=0.5, clearcoat=1.0)
    m_g = s.add_material("lambertian", tex=t_g)
    for (c, m, nuv) in [((-1.2, 1.0, 0.0), m_a, 0), ((1.2, 1.0, -0.8), m_b, abi.NO_UV)]:
        v, idx, n, uv = _uv_sphere(n_lat, 2 * n_lat, 1.0, c, disp)
        s.add_mesh(v, idx, m, normals=n, uv_sets=[(uv * 4).astype(np.float32)], color_uv=0, normal_uv=nuv)
    v, idx, n, uv = _grid_mesh_fast(64, 12.0, lambda x, z: 0.0 * x)
    s.add_mesh(v, idx, m_g, normals=None, uv_sets=[uv], color_uv=0)
    t_env = s.add_texture_image(_sky(env[1], env[0], sun=(0.62, 0.3)), abi.WRAP_CLAMP, abi.WRAP_CLAMP)
    s.set_background_envmap(t_env, radiance_scale=1.0)
    s.build_bvh(abi.BVH_SWEEP)
    return s


def config5_scene(res=(1366, 768), n=500, tex=1024, seed=0xC5):
    """~1 M triangles: a brick field (instanced boxes) + a displaced height field, 4 mip-mapped
    colour textures, 2 normal maps, 1 metallic-roughness map, constant background light."""
    rng = np.random.default_rng(seed)
    s = vimg_amd.HostScene()
    s.set_camera((0.0, 7.0, 13.0), (0.0, 0.5, 0.0), (0, 1, 0), 38.0, res)
    s.set_render_defaults("mis", 2048, 0xFFFFFFFF)

    def noise_tex(sz, base):
        img = rng.random((sz // 8, sz // 8, 3), dtype=np.float32)
        img = np.kron(img, np.ones((8, 8, 1), np.float32))                       # low-pass
        return (0.25 + 0.6 * img) * np.asarray(base, np.float32)
    t_cols = [s.add_texture_image(noise_tex(tex, b)) for b in
              [(0.9, 0.5, 0.4), (0.5, 0.8, 0.5), (0.5, 0.6, 0.9), (0.9, 0.85, 0.6)]]
    t_nms = []
    for _ in range(2):
        nm = rng.normal(0, 0.15, (tex // 2, tex // 2, 3)).astype(np.float32)
        nm[..., 2] = 1.0
        nm /= np.linalg.norm(nm, axis=2, keepdims=True)
        t_nms.append(s.add_texture_image(nm))
    t_rg = s.add_texture_rg(rng.random((256, 256, 2), dtype=np.float32) * 0.8 + 0.1)
    mats = [s.add_material("principled", tex=t_cols[0], normal_map=t_nms[0], mr_tex=t_rg,
                           metallic=1.0, roughness=1.0),
            s.add_material("lambertian", tex=t_cols[1]),
            s.add_material("principled", tex=t_cols[2], normal_map=t_nms[1], metallic=0.1,
                           roughness=0.4, clearcoat=0.5),
            s.add_material("lambertian", tex=t_cols[3])]
    # height field: 2*n*n triangles
    v, idx, nrm, uv = _grid_mesh_fast(n, 9.0, lambda x, z: 0.35 * np.sin(1.3 * x) * np.cos(1.1 * z)
                                      + 0.05 * np.sin(9 * x + 4 * z), uv_scale=6.0)
    s.add_mesh(v, idx, mats[0], normals=nrm, uv_sets=[uv, (uv * 0.5).astype(np.float32)],
               color_uv=0, normal_uv=0, mr_uv=1)
    # bricks: 12 triangles each
    cube_v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32)
    cube_i = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7],
                       [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], np.uint32)
    cube_uv = (cube_v[:, [0, 2]] * 0.5 + 0.5).astype(np.float32)
    nb = 40
    per_mat = [[] for _ in range(3)]
    for bi in range(nb):
        for bj in range(nb):
            c = np.array([(bi - nb / 2) * 0.42, 1.2 + 0.25 * rng.random(), (bj - nb / 2) * 0.42], np.float32)
            per_mat[(bi + bj) % 3].append(cube_v * np.array([0.18, 0.1, 0.18], np.float32) + c)
    for k, blocks in enumerate(per_mat):
        vv = np.concatenate(blocks)
        ii = np.concatenate([cube_i + 8 * b for b in range(len(blocks))])
        uu = np.tile(cube_uv, (len(blocks), 1))
        s.add_mesh(vv, ii, mats[1 + k], normals=None, uv_sets=[uu], color_uv=0,
                   normal_uv=0 if k == 1 else abi.NO_UV)
    s.set_background_const((0.9, 0.95, 1.1), add_to_lights=True)
    s.build_bvh(abi.BVH_SWEEP)
    return s
