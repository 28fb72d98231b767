"""GPU side of the boundary: ctypes mirror of include/vimg_hip.h (libvimg_hip.so, gfx950).

Mirrors the reference's hot-path entry points: ``render`` = scene_integrator
(reference include/integrators.h:36-153), ``trace_pixel`` = trace_pixel (:181-220).
torch is used only as the owner of device memory and streams; there is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _abi as abi
from .host import HostScene, make_params


class HipError(RuntimeError):
    pass


def _lib():
    return abi.hip_lib()


def _check(rc):
    if rc < 0:
        raise HipError(f"[{rc}] " + _lib().vimg_hip_last_error().decode())
    return rc


_side = None


class _Ordered:
    """The stream a launch with torch tensors goes to.  `stream=None` means torch's CURRENT stream,
    never the library's private one (which is non-blocking and would not be ordered against the
    fill of `out` or against torch consumers of the result).  The legacy null stream has no
    handle the library could tell from "no stream given", so work for it goes to a side stream
    that waits for the null stream before the launch and that the null stream waits for after."""

    def __init__(self, stream):
        import torch
        global _side
        self.cur = stream if stream is not None else torch.cuda.current_stream()
        self.side = None
        if self.cur.cuda_stream == 0:
            if _side is None:
                _side = torch.cuda.Stream()
            self.side = _side

    def __enter__(self):
        if self.side is not None:
            self.side.wait_stream(self.cur)
            return C.c_void_p(self.side.cuda_stream)
        return C.c_void_p(self.cur.cuda_stream)

    def __exit__(self, *exc):
        if self.side is not None:
            self.cur.wait_stream(self.side)
        return False


def device_count():
    return _check(_lib().vimg_hip_device_count())


def init(device=0):
    _check(_lib().vimg_hip_init(device))


class DeviceScene:
    """A scene resident in HBM (vimg_hip_scene_upload_opts).  `options`: abi.HipOptions, or keyword
    arguments for one (scheduler="lane" | "pool" | "stage", pool_segments=..., ...); nothing given
    = the library's policy."""

    def __init__(self, host_scene: HostScene, options=None, **opt_kw):
        self._lib = _lib()
        h = C.c_void_p()
        if options is None and opt_kw:
            options = abi.HipOptions(**opt_kw)
        self.options = options
        _check(self._lib.vimg_hip_scene_upload_opts(host_scene.view, C.byref(options) if options is not None else None,
                                                    C.byref(h)))
        self._h = h
        self.resolution = host_scene.resolution

    @property
    def bytes(self):
        return int(self._lib.vimg_hip_scene_bytes(self._h))

    @property
    def kernel(self):
        return self._lib.vimg_hip_scene_kernel(self._h).decode()

    def kernel_for(self, params):
        """Name of the kernel a launch with these parameters gets (the scheduler is chosen per launch)."""
        return self._lib.vimg_hip_launch_kernel(self._h, C.byref(params)).decode()

    def shard_pixels(self, params):
        return _check(self._lib.vimg_hip_shard_pixels(self._h, C.byref(params)))

    def render(self, params, out=None, stats=True, stream=None):
        """Blocking render into a torch CUDA tensor (allocated when ``out`` is None).

        tile_world == 1: returns [H, W, 3] in the reference layout (row 0 = top).
        tile_world  > 1: returns the shard's compact [shard_pixels, 3] buffer."""
        import torch
        w, h = self.resolution
        if out is None:
            if params.tile_world == 1:
                out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
            else:
                out = torch.zeros((self.shard_pixels(params), 3), dtype=torch.float32,
                                  device="cuda")
        st = abi.RenderStats()
        with _Ordered(stream) as sp:
            _check(self._lib.vimg_hip_render(self._h, C.byref(params), C.c_void_p(out.data_ptr()), sp,
                                             C.byref(st) if stats else None))
        return (out, st) if stats else out

    def render_async(self, params, out, stream=None):
        with _Ordered(stream) as sp:
            _check(self._lib.vimg_hip_render_async(self._h, C.byref(params),
                                                   C.c_void_p(out.data_ptr()), sp))

    def check(self):
        """After render_async and a synchronisation of its stream: raises HipError when a launch of this
        scene gave its frame up (the kernel's watchdog), before the frame is used, gathered or timed."""
        _check(self._lib.vimg_hip_check(self._h))

    def render_to_host(self, params, stats=True):
        """Render and copy the framebuffer to a numpy array [H, W, 3] (no torch needed)."""
        w, h = self.resolution
        out = np.empty((h, w, 3), dtype=np.float32)
        st = abi.RenderStats()
        _check(self._lib.vimg_hip_render_to_host(self._h, C.byref(params),
                                                 out.ctypes.data_as(abi.Pf32),
                                                 C.byref(st) if stats else None))
        return (out, st) if stats else out

    def render_heatmap(self, params, factor=-1.0, out=None, stream=None):
        """BVH traversal-cost picture (reference heatmap_img) into a torch CUDA tensor [H, W, 3]
        (or the compact shard slab when params.tile_world > 1)."""
        import torch
        w, h = self.resolution
        if out is None:
            n = w * h if params.tile_world == 1 else self.shard_pixels(params)
            out = torch.empty((n, 3), dtype=torch.float32, device="cuda")
            if params.tile_world == 1:
                out = out.view(h, w, 3)
        with _Ordered(stream) as sp:
            _check(self._lib.vimg_hip_render_heatmap(self._h, C.byref(params), factor,
                                                     C.c_void_p(out.data_ptr()), sp))
        return out

    def trace_pixel(self, params, x, y):
        out = np.zeros(3, dtype=np.float32)
        _check(self._lib.vimg_hip_trace_pixel(self._h, C.byref(params), x, y,
                                              out.ctypes.data_as(abi.Pf32)))
        return out

    def assemble_shards(self, gathered, world, shard_stride_pixels, out=None, stream=None):
        import torch
        w, h = self.resolution
        if out is None:
            out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        with _Ordered(stream) as sp:
            _check(self._lib.vimg_hip_assemble_shards(self._h, world, shard_stride_pixels,
                                                      C.c_void_p(gathered.data_ptr()),
                                                      C.c_void_p(out.data_ptr()), sp))
        return out

    def time_renders(self, params, out, steps):
        ms = np.zeros(steps, dtype=np.float32)
        _check(self._lib.vimg_hip_time_renders(self._h, C.byref(params),
                                               C.c_void_p(out.data_ptr()), steps,
                                               ms.ctypes.data_as(abi.Pf32)))
        return ms

    def probe(self, kind, inputs):
        """Unit-level test hook (vimg_hip_probe): per item n_in floats in, n_out floats out."""
        n_io = {1: (4, 8), 2: (6, 28), 3: (7, 1), 4: (12, 5), 5: (8, 7), 6: (4, 10), 7: (5, 4), 8: (1, 5)}
        fn = self._lib.vimg_hip_probe
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, abi.Pf32, abi.Pf32]
        n_in, n_out = n_io[kind]
        a = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, n_in)
        out = np.zeros((a.shape[0], n_out), dtype=np.float32)
        _check(fn(self._h, kind, a.shape[0], a.ctypes.data_as(abi.Pf32),
                  out.ctypes.data_as(abi.Pf32)))
        return out

    def close(self):
        if self._h:
            self._lib.vimg_hip_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def post_rgb8(image, tonemapper=1, stream=None):
    """Tonemap + sRGB + 8-bit quantise a [H, W, 3] float32 CUDA tensor on the GPU
    (vimg_hip_post_rgb8); returns a [H, W, 3] uint8 CUDA tensor."""
    import torch
    h, w = image.shape[0], image.shape[1]
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=image.device)
    with _Ordered(stream) as sp:
        _check(_lib().vimg_hip_post_rgb8(C.c_void_p(image.data_ptr()), w, h, tonemapper,
                                         C.c_void_p(out.data_ptr()), sp))
    return out


__all__ = ["DeviceScene", "post_rgb8", "HipError", "device_count", "init", "make_params"]


# ---- the pre-step of the path on the GPU (include/vimg_hip.h) ----------
def build_mip_chain(level0, wrap_u=abi.WRAP_REPEAT, wrap_v=abi.WRAP_REPEAT):
    """All mip levels of an [H, W, 3] float image, level 0 first: (flat [texels, 3] array,
    list of per-level (offset_in_texels, w, h))."""
    lib = abi.hip_lib()
    img = np.ascontiguousarray(level0, dtype=np.float32)
    h, w = img.shape[:2]
    n_levels = abi.u32(0)
    texels = int(lib.vimg_hip_mip_chain_texels(w, h, C.byref(n_levels)))
    out = np.empty((texels, 3), dtype=np.float32)
    _check(lib.vimg_hip_build_mip_chain(w, h, img.ctypes.data_as(abi.Pf32), wrap_u, wrap_v,
                                        out.ctypes.data_as(abi.Pf32)))
    levels, off, lw, lh = [], 0, w, h
    for _ in range(n_levels.value):
        levels.append((off, lw, lh))
        off += lw * lh
        lw, lh = max(lw // 2, 1), max(lh // 2, 1)
    return out, levels


def build_env_cdfs(img):
    """(row_cdf [H+1], col_cdfs [H, W+1]) of a lat-long [H, W, 3] float image."""
    lib = abi.hip_lib()
    a = np.ascontiguousarray(img, dtype=np.float32)
    h, w = a.shape[:2]
    row = np.empty(h + 1, dtype=np.float32)
    col = np.empty((h, w + 1), dtype=np.float32)
    _check(lib.vimg_hip_build_env_cdfs(a.ctypes.data_as(abi.Pf32), w, h, row.ctypes.data_as(abi.Pf32),
                                       col.ctypes.data_as(abi.Pf32)))
    return row, col


def lut8_to_float(values_u8, lut256):
    lib = abi.hip_lib()
    a = np.ascontiguousarray(values_u8, dtype=np.uint8)
    lut = np.ascontiguousarray(lut256, dtype=np.float32)
    assert lut.size == 256
    out = np.empty(a.shape, dtype=np.float32)
    _check(lib.vimg_hip_lut8_to_float(a.ctypes.data_as(C.POINTER(C.c_uint8)), a.size,
                                      lut.ctypes.data_as(abi.Pf32), out.ctypes.data_as(abi.Pf32)))
    return out


def rgb8_to_normal(rgb8, scale=1.0):
    lib = abi.hip_lib()
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    assert a.shape[-1] == 3
    out = np.empty(a.shape, dtype=np.float32)
    _check(lib.vimg_hip_rgb8_to_normal(a.ctypes.data_as(C.POINTER(C.c_uint8)), a.size // 3, scale,
                                       out.ctypes.data_as(abi.Pf32)))
    return out


def install_gpu_precompute(enable=True):
    """Make libvimg_host build mip chains and env-map CDFs with the GPU kernels (or, with
    enable=False, with its own loops again)."""
    host = abi.host_lib()
    if not enable:
        host.vimg_host_set_precompute(None, None)
        return
    lib = abi.hip_lib()
    host.vimg_host_set_precompute(C.cast(lib.vimg_hip_build_mip_chain, C.c_void_p),
                                  C.cast(lib.vimg_hip_build_env_cdfs, C.c_void_p))


def lbvh_builder():
    """Function pointer of the GPU LBVH builder for HostScene.build_bvh_with()."""
    return C.cast(abi.hip_lib().vimg_hip_build_lbvh, C.c_void_p)


def ploc_builder():
    """Function pointer of the GPU PLOC + SAH-leaf builder for HostScene.build_bvh_with()."""
    return C.cast(abi.hip_lib().vimg_hip_build_ploc, C.c_void_p)
