"""Host side of the boundary: scene loading / construction, BVH build, post (libvimg_host.so).

Mirrors what the reference's ``main`` does before and after the hot path
(reference src/main.cpp:116-209 and :304-372); nothing here is timed.
"""
import ctypes as C

import numpy as np

from . import _abi as abi


def _fp(a):
    return a.ctypes.data_as(abi.Pf32)


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


class HostError(RuntimeError):
    pass


class HostScene:
    """Owns a ``VimgHostScene``; ``.view`` is the POD ``VimgScene`` the render libraries take."""

    def __init__(self, handle=None):
        self._lib = abi.host_lib()
        self._h = C.c_void_p(handle if handle is not None else self._lib.vimg_host_scene_new())
        self._keep = []

    # ---- construction -------------------------------------------------------------------
    @classmethod
    def from_json(cls, path, bvh=abi.BVH_SWEEP):
        lib = abi.host_lib()
        h = C.c_void_p()
        if lib.vimg_host_scene_from_json_file(str(path).encode(), C.byref(h)) != 0:
            raise HostError(lib.vimg_host_last_error().decode())
        s = cls(h.value)
        s.build_bvh(bvh)
        return s

    @classmethod
    def from_json_text(cls, text, bvh=abi.BVH_SWEEP):
        lib = abi.host_lib()
        h = C.c_void_p()
        if lib.vimg_host_scene_from_json_text(text.encode(), C.byref(h)) != 0:
            raise HostError(lib.vimg_host_last_error().decode())
        s = cls(h.value)
        s.build_bvh(bvh)
        return s

    def _check(self, rc):
        if rc < 0:
            raise HostError(self._lib.vimg_host_last_error().decode())
        return rc

    def set_camera(self, look_from, look_at, up, vfov_deg, res, aperture_radius=0.0,
                   focal_dist=1.0):
        f, a, u = _f32(look_from), _f32(look_at), _f32(up)
        self._lib.vimg_host_set_camera_lookat(self._h, _fp(f), _fp(a), _fp(u), vfov_deg,
                                              int(res[0]), int(res[1]), aperture_radius,
                                              focal_dist)

    def set_render_defaults(self, integrator="mis", samples=16, depth=30):
        self._lib.vimg_host_set_render_defaults(self._h, abi.INTEGRATORS[integrator], samples,
                                                depth & 0xFFFFFFFF)

    def add_texture_const(self, rgb):
        c = _f32(rgb)
        return self._check(self._lib.vimg_host_add_texture_const(self._h, _fp(c)))

    def add_texture_checker(self, w, h, col_a, col_b):
        a, b = _f32(col_a), _f32(col_b)
        return self._check(self._lib.vimg_host_add_texture_checker(self._h, w, h, _fp(a), _fp(b)))

    def add_texture_image(self, rgb, wrap_u=abi.WRAP_REPEAT, wrap_v=abi.WRAP_REPEAT):
        img = _f32(rgb)
        h, w = img.shape[0], img.shape[1]
        return self._check(self._lib.vimg_host_add_texture_image(self._h, w, h, _fp(img), wrap_u,
                                                                 wrap_v))

    def add_texture_rg(self, rg, wrap_u=abi.WRAP_REPEAT, wrap_v=abi.WRAP_REPEAT):
        img = _f32(rg)
        h, w = img.shape[0], img.shape[1]
        return self._check(self._lib.vimg_host_add_texture_rg(self._h, w, h, _fp(img), wrap_u,
                                                              wrap_v))

    def add_material(self, kind, tex=-1, mr_tex=-1, normal_map=-1, emit=(0, 0, 0), ior=1.5,
                     metallic=0.0, roughness=0.5, spec_trans=0.0, subsurface=0.0, specular=0.5,
                     spec_tint=0.0, anisotropic=0.0, sheen=0.0, sheen_tint=0.5, clearcoat=0.0,
                     clearcoat_gloss=1.0, eta=1.5):
        m = abi.Material()
        m.type = {"lambertian": abi.MAT_LAMBERTIAN, "dielectric": abi.MAT_DIELECTRIC,
                  "diffuse_light": abi.MAT_DIFFUSE_LIGHT, "principled": abi.MAT_PRINCIPLED}[kind]
        m.tex, m.mr_tex, m.normal_map = tex, mr_tex, normal_map
        m.emit[0], m.emit[1], m.emit[2] = emit
        m.ior = ior
        m.metallic_factor, m.roughness_factor = metallic, roughness
        m.specular_transmission, m.subsurface, m.specular = spec_trans, subsurface, specular
        m.specular_tint, m.anisotropic, m.sheen, m.sheen_tint = spec_tint, anisotropic, sheen, \
            sheen_tint
        m.clearcoat, m.clearcoat_gloss, m.eta = clearcoat, clearcoat_gloss, eta
        return self._check(self._lib.vimg_host_add_material(self._h, C.byref(m)))

    def add_mesh(self, vertices, indices, material, normals=None, uv_sets=(), color_uv=abi.NO_UV,
                 normal_uv=abi.NO_UV, mr_uv=abi.NO_UV):
        v = _f32(vertices, (-1, 3))
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        n = _f32(normals, (-1, 3)) if normals is not None else None
        uvs = [_f32(u, (-1, 2)) for u in uv_sets]
        arr = (abi.Pf32 * max(len(uvs), 1))(*[_fp(u) for u in uvs])
        return self._check(self._lib.vimg_host_add_mesh(
            self._h, v.shape[0], _fp(v), _fp(n) if n is not None else None, len(uvs), arr,
            idx.shape[0], idx.ctypes.data_as(C.POINTER(abi.u32)), material, color_uv, normal_uv,
            mr_uv))

    def add_quad(self, xform_colmajor, material):
        m = _f32(xform_colmajor, (16,))
        return self._check(self._lib.vimg_host_add_quad(self._h, _fp(m), material))

    def add_sphere(self, center, radius, material):
        c = _f32(center)
        return self._check(self._lib.vimg_host_add_sphere(self._h, _fp(c), radius, material))

    def set_background_const(self, rgb, add_to_lights=True):
        c = _f32(rgb)
        self._lib.vimg_host_set_background_const(self._h, _fp(c), int(add_to_lights))

    def set_background_envmap(self, env_tex, world_to_env=None, env_to_world=None,
                              radiance_scale=1.0):
        eye = np.eye(4, dtype=np.float32).reshape(16)
        w2e = _f32(world_to_env, (16,)) if world_to_env is not None else eye
        e2w = _f32(env_to_world, (16,)) if env_to_world is not None else eye
        self._check(self._lib.vimg_host_set_background_envmap(self._h, env_tex, _fp(w2e),
                                                              _fp(e2w), radiance_scale))

    def build_bvh(self, kind=abi.BVH_SWEEP):
        self._check(self._lib.vimg_host_build_bvh(self._h, kind))

    def build_bvh_with(self, builder_fn_ptr):
        """Build the BVH with a caller-supplied builder (a C function pointer with the
        vimg_bvh_builder_fn signature, e.g. vimg_amd.hip.lbvh_builder())."""
        self._check(self._lib.vimg_host_build_bvh_with(self._h, builder_fn_ptr))
        return self

    # ---- views ---------------------------------------------------------------------------
    @property
    def view(self):
        p = self._lib.vimg_host_scene_view(self._h)
        if not p:
            raise HostError(self._lib.vimg_host_last_error().decode())
        return p

    @property
    def resolution(self):
        cam = self.view.contents.camera
        return cam.res_x, cam.res_y

    def default_params(self, **override):
        p = abi.RenderParams()
        self._lib.vimg_host_default_params(self._h, C.byref(p))
        return make_params(p, **override)

    def bvh_arrays(self):
        b = self.view.contents.bvh
        n = b.num_nodes
        nodes = np.ctypeslib.as_array(C.cast(b.nodes, C.POINTER(abi.u32)), (n, 2)).copy()
        bb = np.ctypeslib.as_array(b.bb_mins_maxes, ((2 * n + 3), 3)).copy()
        obj = np.ctypeslib.as_array(b.obj_indices, (self.view.contents.num_prims,)).copy()
        return nodes, bb, obj, int(b.max_depth)

    def close(self):
        if self._h:
            self._lib.vimg_host_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_params(base=None, integrator=None, samples=None, depth=None, tile_rank=None,
                tile_world=None):
    p = abi.RenderParams()
    if base is not None:
        C.memmove(C.byref(p), C.byref(base), C.sizeof(p))
    else:
        p.integrator, p.samples, p.depth, p.tile_rank, p.tile_world = abi.INTEGRATOR_MIS, 16, 30, \
            0, 1
    if integrator is not None:
        p.integrator = abi.INTEGRATORS[integrator] if isinstance(integrator, str) else integrator
    if samples is not None:
        p.samples = samples
    if depth is not None:
        p.depth = depth & 0xFFFFFFFF
    if tile_rank is not None:
        p.tile_rank = tile_rank
    if tile_world is not None:
        p.tile_world = tile_world
    return p


def tonemap_to_rgb8(rgb, tonemapper=1):
    """clamp(0) / AgX(1) / Reinhard(2) / ACES(3) + sRGB + quantise (reference src/main.cpp:304-356)."""
    lib = abi.host_lib()
    img = _f32(rgb)
    h, w = img.shape[0], img.shape[1]
    out = np.empty((h, w, 3), dtype=np.uint8)
    if lib.vimg_host_tonemap_to_rgb8(_fp(img), w, h, tonemapper,
                                     out.ctypes.data_as(C.POINTER(C.c_uint8))) != 0:
        raise HostError(lib.vimg_host_last_error().decode())
    return out


def write_png(path, rgb8):
    lib = abi.host_lib()
    img = np.ascontiguousarray(rgb8, dtype=np.uint8)
    if lib.vimg_host_write_png(str(path).encode(), img.ctypes.data_as(C.POINTER(C.c_uint8)),
                               img.shape[1], img.shape[0]) != 0:
        raise HostError(lib.vimg_host_last_error().decode())


def srgb8_lut():
    lut = np.empty(256, dtype=np.float32)
    abi.host_lib().vimg_host_srgb8_lut(lut.ctypes.data_as(abi.Pf32))
    return lut


def srgb8_to_linear(values_u8):
    a = np.ascontiguousarray(values_u8, dtype=np.uint8)
    out = np.empty(a.shape, dtype=np.float32)
    abi.host_lib().vimg_host_srgb8_to_linear(a.ctypes.data_as(C.POINTER(C.c_uint8)), a.size,
                                             out.ctypes.data_as(abi.Pf32))
    return out


def rgb8_to_normal(rgb8, scale=1.0):
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    out = np.empty(a.shape, dtype=np.float32)
    abi.host_lib().vimg_host_rgb8_to_normal(a.ctypes.data_as(C.POINTER(C.c_uint8)), a.size // 3,
                                            scale, out.ctypes.data_as(abi.Pf32))
    return out
