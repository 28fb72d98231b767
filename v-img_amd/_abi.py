"""ctypes mirror of include/vimg_scene.h, include/vimg_host.h and include/vimg_hip.h.

Only declarations live here: struct layouts, library loading and argtypes.  The product
libraries are loaded from ``v-img_amd/lib`` (built in-tree by ``make``); a missing library is an
error, never a fallback (the HIP path must fail loudly when its extension is absent).
"""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(PKG_DIR, "lib")
REPO_ROOT = os.path.dirname(PKG_DIR)

NO_UV = 255
MAX_UV_SETS = 4
MAX_MIP_LEVELS = 15

INTEGRATOR_S_NORMAL, INTEGRATOR_G_NORMAL, INTEGRATOR_MATERIAL, INTEGRATOR_MIS = 0, 1, 2, 3
INTEGRATORS = {"s_normal": 0, "g_normal": 1, "material": 2, "mis": 3}
PRIM_TRIANGLE, PRIM_SPHERE = 0, 1
MAT_LAMBERTIAN, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_PRINCIPLED = 0, 1, 2, 3
TEX_CONST, TEX_CHECKER, TEX_IMAGE = 0, 1, 2
WRAP_CLAMP, WRAP_MIRROR, WRAP_REPEAT = 0, 1, 2
LIGHT_PRIM, LIGHT_BACKGROUND = 0, 1
BG_CONST, BG_ENVMAP = 0, 1
BVH_BINNED, BVH_SWEEP = 0, 1

f32, u32, i32, u64, i64 = C.c_float, C.c_uint32, C.c_int32, C.c_uint64, C.c_int64


class Camera(C.Structure):
    _fields_ = [("cam_to_world", f32 * 16), ("vfov_deg", f32), ("res_x", i32), ("res_y", i32),
                ("aperture_radius", f32), ("focal_dist", f32)]


class Prim(C.Structure):
    _fields_ = [("type", u32), ("index", u32)]


class Mesh(C.Structure):
    _fields_ = [("first_vertex", u32), ("num_vertices", u32), ("has_normals", u32),
                ("num_uv_sets", u32), ("uv_offset", u32 * MAX_UV_SETS), ("color_tex_uv", u32),
                ("normal_tex_uv", u32), ("metallic_roughness_tex_uv", u32), ("material", u32)]


class Sphere(C.Structure):
    _fields_ = [("center", f32 * 3), ("radius", f32), ("material", u32)]


class Material(C.Structure):
    _fields_ = [("type", u32), ("tex", i32), ("mr_tex", i32), ("normal_map", i32),
                ("emit", f32 * 3), ("ior", f32), ("metallic_factor", f32),
                ("roughness_factor", f32), ("specular_transmission", f32), ("subsurface", f32),
                ("specular", f32), ("specular_tint", f32), ("anisotropic", f32), ("sheen", f32),
                ("sheen_tint", f32), ("clearcoat", f32), ("clearcoat_gloss", f32), ("eta", f32)]


class Texture(C.Structure):
    _fields_ = [("type", u32), ("col_a", f32 * 3), ("col_b", f32 * 3), ("width", u32),
                ("height", u32), ("num_levels", u32), ("wrap_u", u32), ("wrap_v", u32),
                ("level_offset", u64 * MAX_MIP_LEVELS)]


class TextureRG(C.Structure):
    _fields_ = [("width", u32), ("height", u32), ("wrap_u", u32), ("wrap_v", u32),
                ("offset", u64)]


class Light(C.Structure):
    _fields_ = [("type", u32), ("prim", u32)]


class Background(C.Structure):
    _fields_ = [("type", u32), ("col", f32 * 3), ("env_tex", i32), ("world_to_env", f32 * 16),
                ("env_to_world", f32 * 16), ("radiance_scale", f32), ("row_cdf_offset", u64),
                ("col_cdf_offset", u64)]


class BVHNode(C.Structure):
    _fields_ = [("first_index", u32), ("obj_count", u32)]


class BVH(C.Structure):
    _fields_ = [("num_nodes", u32), ("max_depth", u32), ("nodes", C.POINTER(BVHNode)),
                ("bb_mins_maxes", C.POINTER(f32)), ("obj_indices", C.POINTER(u32))]


class Scene(C.Structure):
    _fields_ = [
        ("camera", Camera), ("background", Background),
        ("num_prims", u32), ("prims", C.POINTER(Prim)),
        ("num_tris", u32), ("tri_indices", C.POINTER(u32)), ("tri_mesh", C.POINTER(u32)),
        ("num_meshes", u32), ("meshes", C.POINTER(Mesh)),
        ("num_vertices", u32), ("vertices", C.POINTER(f32)), ("normals", C.POINTER(f32)),
        ("num_uvs", u64), ("uvs", C.POINTER(f32)),
        ("num_spheres", u32), ("spheres", C.POINTER(Sphere)),
        ("num_materials", u32), ("materials", C.POINTER(Material)),
        ("num_textures", u32), ("textures", C.POINTER(Texture)),
        ("num_texels", u64), ("texels", C.POINTER(f32)),
        ("num_rg_textures", u32), ("rg_textures", C.POINTER(TextureRG)),
        ("num_rg_texels", u64), ("rg_texels", C.POINTER(f32)),
        ("num_lights", u32), ("lights", C.POINTER(Light)),
        ("num_cdf", u64), ("cdf_pool", C.POINTER(f32)),
        ("bvh", BVH),
    ]


class RenderParams(C.Structure):
    _fields_ = [("integrator", u32), ("samples", u32), ("depth", u32), ("tile_rank", u32),
                ("tile_world", u32)]


class RenderStats(C.Structure):
    _fields_ = [("paths", u64), ("closest_rays", u64), ("shadow_rays", u64),
                ("internal_visits", u64), ("leaf_visits", u64), ("prim_tests", u64),
                ("sphere_tests", u64), ("nan_samples", u64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}

    @property
    def rays(self):
        return int(self.closest_rays) + int(self.shadow_rays)


OPT_AUTO = -1
SCHED_LANE, SCHED_POOL, SCHED_STAGE, SCHED_POOL4, SCHED_POOL4G, SCHED_CU = 1, 2, 3, 4, 5, 6
SCHEDULERS = {"lane": SCHED_LANE, "pool": SCHED_POOL, "stage": SCHED_STAGE, "pool4": SCHED_POOL4, "pool4g": SCHED_POOL4G, "cu": SCHED_CU}


class HipOptions(C.Structure):
    """VimgHipOptions (include/vimg_hip.h): every field -1 = the library's policy."""
    _fields_ = [("struct_size", u32), ("scheduler", i32), ("waves_per_simd", i32),
                ("lds_budget_kb", i32), ("pool_slots", i32), ("pool_segments", i32),
                ("pool_refill", i32), ("pool_vbatch", i32), ("pool_classes", i32),
                ("pool_starve", i32), ("pool_boxmin", i32), ("lds_leaf", i32),
                ("stage_slots", i32), ("stage_seg_len", i32), ("stage_wchunk", i32),
                ("stage_walk_quota", i32), ("pool4_rays", i32), ("lds_stack", i32), ("pool_gbreak", i32),
                ("cu_waves", i32), ("cu_walkers", i32), ("cu_flex", i32), ("cu_lowwater", i32), ("cu_patience", i32),
                ("cu_join", i32), ("cu_sleep", i32)]

    def __init__(self, **kw):
        super().__init__()
        self.struct_size = C.sizeof(HipOptions)
        for name, _ in self._fields_[1:]:
            setattr(self, name, OPT_AUTO)
        for k, v in kw.items():
            if k == "scheduler" and isinstance(v, str):
                v = SCHEDULERS[v]
            if k not in dict(self._fields_):
                raise TypeError(f"unknown option {k}")
            setattr(self, k, v)


PScene = C.POINTER(Scene)
PParams = C.POINTER(RenderParams)
PStats = C.POINTER(RenderStats)
Pf32 = C.POINTER(f32)

# name -> (restype, argtypes); also the list of symbols include/vimg_host.h declares
HOST_SYMBOLS = {
    "vimg_host_scene_from_json_file": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "vimg_host_scene_from_json_text": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "vimg_host_scene_new": (C.c_void_p, []),
    "vimg_host_scene_free": (None, [C.c_void_p]),
    "vimg_host_set_camera_lookat": (None, [C.c_void_p, Pf32, Pf32, Pf32, f32, C.c_int, C.c_int,
                                           f32, f32]),
    "vimg_host_set_render_defaults": (None, [C.c_void_p, u32, u32, u32]),
    "vimg_host_add_texture_const": (C.c_int, [C.c_void_p, Pf32]),
    "vimg_host_add_texture_checker": (C.c_int, [C.c_void_p, u32, u32, Pf32, Pf32]),
    "vimg_host_add_texture_image": (C.c_int, [C.c_void_p, u32, u32, Pf32, u32, u32]),
    "vimg_host_add_texture_rg": (C.c_int, [C.c_void_p, u32, u32, Pf32, u32, u32]),
    "vimg_host_add_material": (C.c_int, [C.c_void_p, C.POINTER(Material)]),
    "vimg_host_add_mesh": (C.c_int, [C.c_void_p, u32, Pf32, Pf32, u32, C.POINTER(Pf32), u32,
                                     C.POINTER(u32), u32, u32, u32, u32]),
    "vimg_host_add_quad": (C.c_int, [C.c_void_p, Pf32, u32]),
    "vimg_host_add_sphere": (C.c_int, [C.c_void_p, Pf32, f32, u32]),
    "vimg_host_set_background_const": (None, [C.c_void_p, Pf32, C.c_int]),
    "vimg_host_set_background_envmap": (C.c_int, [C.c_void_p, C.c_int, Pf32, Pf32, f32]),
    "vimg_host_set_precompute": (None, [C.c_void_p, C.c_void_p]),
    "vimg_host_srgb8_lut": (None, [Pf32]),
    "vimg_host_srgb8_to_linear": (None, [C.POINTER(C.c_uint8), C.c_uint64, Pf32]),
    "vimg_host_rgb8_to_normal": (None, [C.POINTER(C.c_uint8), C.c_uint64, f32, Pf32]),
    "vimg_host_build_bvh": (C.c_int, [C.c_void_p, C.c_int]),
    "vimg_host_build_bvh_with": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vimg_host_scene_view": (PScene, [C.c_void_p]),
    "vimg_host_default_params": (None, [C.c_void_p, PParams]),
    "vimg_host_tonemap_to_rgb8": (C.c_int, [Pf32, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(C.c_uint8)]),
    "vimg_host_write_png": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint8), C.c_int, C.c_int]),
    "vimg_host_last_error": (C.c_char_p, []),
}

# the list of symbols include/vimg_hip.h declares
HIP_SYMBOLS = {
    "vimg_hip_init": (C.c_int, [C.c_int]),
    "vimg_hip_device_count": (C.c_int, []),
    "vimg_hip_options_default": (None, [C.POINTER(HipOptions)]),
    "vimg_hip_scene_upload": (C.c_int, [PScene, C.POINTER(C.c_void_p)]),
    "vimg_hip_scene_upload_opts": (C.c_int, [PScene, C.POINTER(HipOptions), C.POINTER(C.c_void_p)]),
    "vimg_hip_scene_free": (C.c_int, [C.c_void_p]),
    "vimg_hip_shard_pixels": (i64, [C.c_void_p, PParams]),
    "vimg_hip_render": (C.c_int, [C.c_void_p, PParams, C.c_void_p, C.c_void_p, PStats]),
    "vimg_hip_render_async": (C.c_int, [C.c_void_p, PParams, C.c_void_p, C.c_void_p]),
    "vimg_hip_check": (C.c_int, [C.c_void_p]),
    "vimg_hip_render_to_host": (C.c_int, [C.c_void_p, PParams, Pf32, PStats]),
    "vimg_hip_trace_pixel": (C.c_int, [C.c_void_p, PParams, C.c_int, C.c_int, Pf32]),
    "vimg_hip_render_heatmap": (C.c_int, [C.c_void_p, PParams, f32, C.c_void_p, C.c_void_p]),
    "vimg_hip_assemble_shards": (C.c_int, [C.c_void_p, u32, i64, C.c_void_p, C.c_void_p,
                                           C.c_void_p]),
    "vimg_hip_time_renders": (C.c_int, [C.c_void_p, PParams, C.c_void_p, C.c_int, Pf32]),
    "vimg_hip_post_rgb8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p]),
    "vimg_hip_mip_chain_texels": (C.c_uint64, [u32, u32, C.POINTER(u32)]),
    "vimg_hip_build_mip_chain": (C.c_int, [u32, u32, Pf32, u32, u32, Pf32]),
    "vimg_hip_build_env_cdfs": (C.c_int, [Pf32, u32, u32, Pf32, Pf32]),
    "vimg_hip_lut8_to_float": (C.c_int, [C.POINTER(C.c_uint8), C.c_uint64, Pf32, Pf32]),
    "vimg_hip_rgb8_to_normal": (C.c_int, [C.POINTER(C.c_uint8), C.c_uint64, f32, Pf32]),
    "vimg_hip_build_lbvh": (C.c_int, [u32, Pf32, C.POINTER(u32), C.POINTER(u32), C.c_void_p, Pf32,
                                     C.POINTER(u32)]),
    "vimg_hip_build_ploc": (C.c_int, [u32, Pf32, C.POINTER(u32), C.POINTER(u32), C.c_void_p, Pf32,
                                     C.POINTER(u32)]),
    "vimg_hip_scene_bytes": (i64, [C.c_void_p]),
    "vimg_hip_scene_kernel": (C.c_char_p, [C.c_void_p]),
    "vimg_hip_launch_kernel": (C.c_char_p, [C.c_void_p, PParams]),
    "vimg_hip_last_error": (C.c_char_p, []),
}


def _bind(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_host_lib = None
_hip_lib = None


def host_lib():
    global _host_lib
    if _host_lib is None:
        path = os.path.join(LIB_DIR, "libvimg_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make host` (or __graft_entry__.build())")
        _host_lib = _bind(C.CDLL(path), HOST_SYMBOLS)
    return _host_lib


def hip_lib():
    """The HIP extension.  No CPU fallback exists: a missing library raises."""
    global _hip_lib
    if _hip_lib is None:
        # VIMG_HIP_LIB selects another build of the same library (A/B measurements only)
        path = os.environ.get("VIMG_HIP_LIB") or os.path.join(LIB_DIR, "libvimg_hip.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make hip` (or __graft_entry__.build()); "
                               "there is no CPU fallback for the render path")
        # One HIP runtime per process: the torch wheel ships its own libamdhip64 / ROCr, and a
        # second copy in the same process cannot open the GPU.  Importing torch first makes the
        # loader resolve our DT_NEEDED libamdhip64.so.7 to the copy torch already mapped.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _hip_lib = _bind(C.CDLL(path), HIP_SYMBOLS)
    return _hip_lib
