"""Tile sharding across ranks (one process per GPU) and the single gather at the end.

The reference already cuts the image into 8x8 tiles and deals them statically to its worker
threads with no communication (include/integrators.h:57-65,101).  Here the same tiles are dealt
to ranks: rank r of n renders tiles t with t % n == r into a compact buffer, the buffers are
gathered ONCE (all_gather of equal padded slabs: RCCL on GPUs, gloo in the CPU tests) and
de-interleaved into the reference image layout.  No collective sits on the data path.
"""
import numpy as np


def tile_grid(width, height):
    return (width + 7) // 8, (height + 7) // 8


def shard_tiles(width, height, rank, world):
    """Global ids of the 8x8 tiles (reference x-major order) owned by `rank`."""
    tx, ty = tile_grid(width, height)
    return np.arange(rank, tx * ty, world, dtype=np.int64)


def shard_stride_pixels(width, height, world):
    """Pixels per padded shard slab = 64 * ceil(tiles / world)."""
    tx, ty = tile_grid(width, height)
    return 64 * ((tx * ty + world - 1) // world)


def assemble_numpy(gathered, width, height, world):
    """Host restatement of the assemble kernel: [world, stride, 3] -> [H, W, 3].
    Used by the CPU (gloo) tests and as the checker of the device kernel."""
    tx, ty = tile_grid(width, height)
    stride = gathered.shape[1]
    out = np.zeros((height, width, 3), dtype=gathered.dtype)
    tiles = np.arange(tx * ty)
    rank, local = tiles % world, tiles // world
    for t in tiles:
        x0, y0 = (t // ty) * 8, (t % ty) * 8
        block = gathered[rank[t], local[t] * 64:(local[t] + 1) * 64].reshape(8, 8, 3)
        h = min(8, height - y0)
        w = min(8, width - x0)
        # within-tile index = ty_*8 + tx_, image row = H-1-y
        for yy in range(h):
            out[height - 1 - (y0 + yy), x0:x0 + w] = block[yy, :w]
    assert stride >= 64 * ((tx * ty + world - 1) // world)
    return out


def render_sharded(dev_scene, params_fn, rank, world, group=None):
    """Render this rank's tiles on the GPU, all_gather the compact slabs, assemble on every rank.

    params_fn(tile_rank, tile_world) -> VimgRenderParams.  Returns (image tensor, stats)."""
    import torch
    import torch.distributed as dist
    w, h = dev_scene.resolution
    params = params_fn(rank, world)
    if world == 1:
        return dev_scene.render(params)
    stride = shard_stride_pixels(w, h, world)
    slab = torch.zeros((stride, 3), dtype=torch.float32, device="cuda")
    _, stats = dev_scene.render(params, out=slab)
    # concatenated layout [world * stride, 3]: accepted by both RCCL and gloo
    gathered = torch.empty((world * stride, 3), dtype=torch.float32, device="cuda")
    dist.all_gather_into_tensor(gathered, slab, group=group)
    image = dev_scene.assemble_shards(gathered, world, stride)
    torch.cuda.synchronize()
    dev_scene.check()   # (a no-op after the blocking render above; the guard of callers that switch to render_async)
    return image, stats


def extract_shard(image, rank, world):
    """Inverse of assemble: cut the compact [stride, 3] slab of `rank` out of a full [H, W, 3]
    image (pixels outside the image stay 0).  Host helper for tests and CPU-side plumbing."""
    height, width = image.shape[0], image.shape[1]
    tx, ty = tile_grid(width, height)
    stride = shard_stride_pixels(width, height, world)
    slab = np.zeros((stride, 3), dtype=image.dtype)
    for local, t in enumerate(shard_tiles(width, height, rank, world)):
        x0, y0 = (t // ty) * 8, (t % ty) * 8
        h, w = min(8, height - y0), min(8, width - x0)
        block = np.zeros((8, 8, 3), dtype=image.dtype)
        for yy in range(h):
            block[yy, :w] = image[height - 1 - (y0 + yy), x0:x0 + w]
        slab[local * 64:(local + 1) * 64] = block.reshape(64, 3)
    return slab
