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


def feature_scene(res=(96, 64), seed=7, envmap=True, lens=True):
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
    rg = rng.random((8, 8, 2), dtype=np.float32) * 0.8 + 0.1
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
