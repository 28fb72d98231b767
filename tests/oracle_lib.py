"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package (v-img_amd/)."""
import ctypes as C
import os

import numpy as np

import vimg_amd
from vimg_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE_CAMERA_RAY, PROBE_CLOSEST_HIT, PROBE_OCCLUDED, PROBE_BSDF_EVAL, PROBE_BSDF_SAMPLE, \
    PROBE_LIGHT_SAMPLE, PROBE_BACKGROUND = 1, 2, 3, 4, 5, 6, 7
PROBE_IO = {1: (4, 8), 2: (6, 28), 3: (7, 1), 4: (12, 5), 5: (8, 7), 6: (4, 10), 7: (5, 4)}

_libs = {}


def load(name="liboracle.so"):
    if name not in _libs:
        path = os.path.join(ROOT, "oracle", name)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make oracle`")
        lib = C.CDLL(path)
        lib.oracle_render.restype = C.c_int
        lib.oracle_render.argtypes = [abi.PScene, abi.PParams, C.c_int, abi.Pf32, abi.PStats]
        lib.oracle_trace_pixel.restype = C.c_int
        lib.oracle_trace_pixel.argtypes = [abi.PScene, abi.PParams, C.c_int, C.c_int, abi.Pf32]
        u64p = C.POINTER(C.c_uint64)
        lib.oracle_pcg32_srandom.argtypes = [u64p, C.c_uint64, C.c_uint64]
        lib.oracle_pcg32_random.restype = C.c_uint32
        lib.oracle_pcg32_random.argtypes = [u64p]
        lib.oracle_rand_float.restype = C.c_float
        lib.oracle_rand_float.argtypes = [u64p]
        lib.oracle_random_x_y_r2.argtypes = [C.c_uint32, abi.Pf32]
        lib.oracle_probe.restype = C.c_int
        lib.oracle_probe.argtypes = [abi.PScene, C.c_int, C.c_int, abi.Pf32, abi.Pf32]
        lib.oracle_uses_float_libm.restype = C.c_int
        lib.oracle_heatmap.restype = C.c_int
        lib.oracle_heatmap.argtypes = [abi.PScene, abi.PParams, C.c_float, C.c_int, abi.Pf32, abi.Pf32]
        lib.oracle_post_rgb8.restype = C.c_int
        lib.oracle_post_rgb8.argtypes = [abi.Pf32, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
        _libs[name] = lib
    return _libs[name]


def render(scene, params, threads=0, lib=None):
    """scene: vimg_amd.HostScene.  Returns (image[H,W,3] float32, stats, threads_used)."""
    lib = lib or load()
    w, h = scene.resolution
    out = np.zeros((h, w, 3), dtype=np.float32)
    stats = abi.RenderStats()
    used = lib.oracle_render(scene.view, C.byref(params), threads, out.ctypes.data_as(abi.Pf32),
                             C.byref(stats))
    if used < 0:
        raise RuntimeError("oracle_render rejected its arguments")
    return out, stats, used


def heatmap(scene, params, factor=-1.0, threads=0, lib=None):
    """Returns (turbo image [H,W,3], truncated per-pixel average cost [H,W])."""
    lib = lib or load()
    w, h = scene.resolution
    out = np.zeros((h, w, 3), dtype=np.float32)
    counts = np.zeros((h, w), dtype=np.float32)
    if lib.oracle_heatmap(scene.view, C.byref(params), factor, threads,
                          out.ctypes.data_as(abi.Pf32), counts.ctypes.data_as(abi.Pf32)) < 0:
        raise RuntimeError("oracle_heatmap rejected its arguments")
    return out, counts


def trace_pixel(scene, params, x, y, lib=None):
    lib = lib or load()
    out = np.zeros(3, dtype=np.float32)
    if lib.oracle_trace_pixel(scene.view, C.byref(params), x, y, out.ctypes.data_as(abi.Pf32)):
        raise RuntimeError("oracle_trace_pixel rejected its arguments")
    return out


def probe(scene, kind, inputs, lib=None):
    lib = lib or load()
    n_in, n_out = PROBE_IO[kind]
    a = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, n_in)
    out = np.zeros((a.shape[0], n_out), dtype=np.float32)
    if lib.oracle_probe(scene.view, kind, a.shape[0], a.ctypes.data_as(abi.Pf32),
                        out.ctypes.data_as(abi.Pf32)):
        raise RuntimeError("oracle_probe rejected its arguments")
    return out


class Pcg:
    def __init__(self, initstate, initseq=0, lib=None):
        self.lib = lib or load()
        self.si = (C.c_uint64 * 2)()
        self.lib.oracle_pcg32_srandom(self.si, initstate, initseq)

    @property
    def state(self):
        return int(self.si[0])

    @property
    def inc(self):
        return int(self.si[1])

    def u32(self):
        return int(self.lib.oracle_pcg32_random(self.si))

    def rand_float(self):
        return float(self.lib.oracle_rand_float(self.si))


def r2(n, lib=None):
    lib = lib or load()
    out = (C.c_float * 2)()
    lib.oracle_random_x_y_r2(n, out)
    return float(out[0]), float(out[1])


def post_rgb8(image, tonemapper, lib=None):
    lib = lib or load()
    img = np.ascontiguousarray(image, dtype=np.float32)
    out = np.empty(img.shape, dtype=np.uint8)
    if lib.oracle_post_rgb8(img.ctypes.data_as(abi.Pf32), img.shape[1], img.shape[0], tonemapper,
                            out.ctypes.data_as(C.POINTER(C.c_uint8))):
        raise RuntimeError("oracle_post_rgb8 rejected its arguments")
    return out
