"""Multi-GPU sharding of one frame: image tiles dealt round-robin over ranks, one
reduce(sum) of the float framebuffer to rank 0.  Works with any torch.distributed backend
(nccl = RCCL over xGMI on the GPUs; gloo in the CPU tests).  No data-path collective other than
the final reduce: tiles are independent, every rank holds the whole (small) scene.
"""
import numpy as np


def tile_owner_map(width, height, tile=32, nranks=1):
    """(H,W) int array: rank that renders each pixel.  Mirrors the device mapping in
    csrc/prt_device.h owned_to_pixel(): tile slot k = ty*tiles_x + kx belongs to rank k % nranks and
    covers tile (tx, ty) with tx = (kx + 3*ty) % tiles_x (rows rotated so a rank's tiles form diagonals)."""
    tile = max(8, (tile + 7) // 8 * 8)
    tiles_x = (width + tile - 1) // tile
    py, px = np.mgrid[0:height, 0:width]
    ty, tx = py // tile, px // tile
    kx = (tx - 3 * ty) % tiles_x
    k = ty * tiles_x + kx
    return (k % nranks).astype(np.int32)


def owned_mask(width, height, tile, rank, nranks):
    return tile_owner_map(width, height, tile, nranks) == rank


def reduce_framebuffer(fb, dst=0):
    """Sum the per-rank framebuffers onto `dst`.  Disjoint tiles => every element is x + 0 + ... + 0: the reduce is exact
    whatever its order, i.e. the frame is the union of the ranks' shares (a share equals the single-rank render's pixels up
    to how its launch grouped a pixel's samples into chunks)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM)
    return fb
