"""v-img_amd — MI355X-native hot path of the v-img path tracer behind a C ABI.

Importable as ``vimg_amd`` (the directory name has a hyphen; ``vimg_amd.py`` at the repository
root registers this package under that name).

    host   : scene loading / construction, SAH BVH build, post      (libvimg_host.so, CPU)
    hip    : upload + render on the GPU                              (libvimg_hip.so, gfx950)
    dist   : tile sharding across ranks + the one RCCL gather
"""
from . import _abi as abi  # noqa: F401
from .host import HostScene, make_params, tonemap_to_rgb8, write_png  # noqa: F401

__all__ = ["abi", "HostScene", "make_params", "tonemap_to_rgb8", "write_png"]
