"""N>1 path on CPU: world_size-2 gloo processes.  Each rank produces the pixels of its own tiles
(with the CPU oracle standing in for the GPU render — test infrastructure), the framebuffers are
combined with the product's reduce_framebuffer(), and rank 0 must hold the single-rank image bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tile, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from pooraytracer_amd import distributed, scenes
    data = scenes.cornell_box(ball_subdiv=1, width=80, height=56)
    full, _ = oracle.Oracle(data).render(spp=2, max_depth=4, seed=7)
    mask = distributed.owned_mask(80, 56, tile, rank, world)
    mine = np.where(mask[..., None], full, 0.0).astype(np.float32)  # what prt_render_device(rank, nranks) leaves in the fb
    fb = torch.from_numpy(mine.copy())
    distributed.reduce_framebuffer(fb, dst=0)
    if rank == 0:
        np.save(out_path, fb.numpy())
        np.save(out_path + ".full.npy", full.astype(np.float32))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,tile", [(2, 16), (2, 32)])
def test_tile_sharded_reduce_gloo(tmp_path, world, tile):
    out = str(tmp_path / "fb.npy")
    mp.spawn(_worker, args=(world, _free_port(), tile, out), nprocs=world, join=True)
    assert np.array_equal(np.load(out), np.load(out + ".full.npy"))


def test_tile_owner_map_partitions_image():
    from pooraytracer_amd import distributed
    for w, h, tile, n in [(80, 56, 16, 3), (1024, 1024, 32, 8), (1280, 720, 32, 8), (33, 17, 8, 2)]:
        owner = distributed.tile_owner_map(w, h, tile, n)
        assert owner.shape == (h, w) and owner.min() == 0 and owner.max() == min(n, ((w + tile - 1) // tile) * ((h + tile - 1) // tile)) - 1
        counts = np.bincount(owner.ravel(), minlength=n)
        assert counts.sum() == w * h
        if w * h >= 1024 * 720:  # round-robin tiles balance the big frames to within one tile row
            assert counts.max() - counts.min() <= tile * tile * 2
