"""The N > 1 path on CPU: tile sharding + the single gather, world_size 2 over gloo.

The render itself is the GPU kernel in production; here each rank's shard is produced by the CPU
oracle so that the sharding arithmetic, the collective's shapes/padding and the reassembly are
exercised end to end without a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
import scenes
from vimg_amd import dist as vdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, res, spp, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = scenes.json_scene("disney_spheres.json", res=res)
    w, h = res
    part, st, _ = O.render(s, s.default_params(samples=spp, tile_rank=rank, tile_world=world),
                           threads=2)
    slab = torch.from_numpy(vdist.extract_shard(part, rank, world))
    stride = vdist.shard_stride_pixels(w, h, world)
    assert slab.shape == (stride, 3)
    gathered = torch.empty((world * stride, 3), dtype=torch.float32)   # concatenated slabs
    dist.all_gather_into_tensor(gathered, slab)            # the ONE collective of a frame
    gathered = gathered.view(world, stride, 3)
    rays = torch.tensor([st.rays, st.paths], dtype=torch.float64)
    dist.all_reduce(rays)
    image = vdist.assemble_numpy(gathered.numpy(), w, h, world)
    if rank == 0:
        np.savez(out_path, image=image, rays=rays.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("res", [(64, 40), (61, 37)])      # tile-aligned and ragged
def test_two_rank_gloo_render_equals_single_rank(tmp_path, res):
    spp = 2
    out_path = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(2, _free_port(), res, spp, out_path), nprocs=2, join=True)
    got = np.load(out_path)
    s = scenes.json_scene("disney_spheres.json", res=res)
    full, st, _ = O.render(s, s.default_params(samples=spp), threads=2)
    assert np.array_equal(got["image"], full)
    assert got["rays"].tolist() == [st.rays, st.paths]


def test_shard_helpers_round_trip():
    rng = np.random.default_rng(0)
    for (w, h, world) in [(123, 61, 3), (64, 64, 8), (9, 9, 2), (1800, 800, 8)]:
        if w * h > 200000:
            tx, ty = vdist.tile_grid(w, h)
            assert vdist.shard_stride_pixels(w, h, world) == 64 * ((tx * ty + world - 1) // world)
            assert sum(len(vdist.shard_tiles(w, h, r, world)) for r in range(world)) == tx * ty
            continue
        img = rng.random((h, w, 3), dtype=np.float32)
        slabs = np.stack([vdist.extract_shard(img, r, world) for r in range(world)])
        assert np.array_equal(vdist.assemble_numpy(slabs, w, h, world), img)
