"""Tile sharding across GPUs and the final framebuffer gather.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the
CPU tests). The scene is replicated; the image is cut into 8x8-pixel tiles dealt round-robin
(tile_id % world == rank) because hair coverage is spatially clustered; every rank renders all
samples of its own tiles with NO data-path collective (pixels are independent: each owns its
PCG32 stream, pt.cpp:1942-1945). The only exchange is ONE gather of the packed float4 tiles to
rank 0 at the end, where a small kernel un-interleaves them. torch is plumbing here: device
buffers and the collective; all arithmetic of the path is in libyhair.so.
"""
import numpy as np

TILE = 8


def tiles_xy(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def shard_tiles(width, height, rank, world):
    tx, ty = tiles_xy(width, height)
    return np.arange(rank, tx * ty, world, dtype=np.int64)


def shard_pixels(width, height, rank, world):
    return int(len(shard_tiles(width, height, rank, world))) * TILE * TILE


def pack_tiles_host(image, rank, world):
    """Host restatement of k_pack (csrc/kernels.hip) for tests: image (H, W, 4) -> (ntiles*64, 4)."""
    h, w, _ = image.shape
    tx, _ = tiles_xy(w, h)
    tiles = shard_tiles(w, h, rank, world)
    out = np.zeros((len(tiles), TILE, TILE, 4), image.dtype)
    for k, t in enumerate(tiles):
        i0, j0 = (t % tx) * TILE, (t // tx) * TILE
        blk = image[j0:j0 + TILE, i0:i0 + TILE]
        out[k, :blk.shape[0], :blk.shape[1]] = blk
    return out.reshape(-1, 4)


def unpack_tiles_host(packed, src_rank, world, image):
    """Host restatement of k_unpack: scatters src_rank's packed tiles into image (H, W, 4)."""
    h, w, _ = image.shape
    tx, _ = tiles_xy(w, h)
    tiles = shard_tiles(w, h, src_rank, world)
    packed = np.asarray(packed).reshape(-1, TILE, TILE, 4)
    for k, t in enumerate(tiles):
        i0, j0 = (t % tx) * TILE, (t // tx) * TILE
        hh, ww = min(TILE, h - j0), min(TILE, w - i0)
        image[j0:j0 + hh, i0:i0 + ww] = packed[k, :hh, :ww]
    return image


def gather_framebuffer(packed, width, height, rank, world, ctx=None, dst=0, force_collective=False):
    """packed: torch tensor (shard_pixels, 4) float32 on this rank's device (or CPU under gloo).
    Returns the full (H, W, 4) torch tensor on rank `dst`, None elsewhere. One collective.
    force_collective: a world of one still goes through dist.gather (its process group must exist): the way a
    one-GPU box executes the RCCL branch."""
    import torch
    import torch.distributed as dist

    if world == 1 and not force_collective:
        shards = [packed]
    else:
        cap = max(shard_pixels(width, height, r, world) for r in range(world))
        send = torch.zeros((cap, 4), dtype=torch.float32, device=packed.device)
        send[: packed.shape[0]] = packed
        recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, recv, dst=dst)
        if rank != dst:
            return None
        shards = [recv[r][: shard_pixels(width, height, r, world)] for r in range(world)]
    image = torch.zeros((height, width, 4), dtype=torch.float32, device=packed.device)
    if image.is_cuda:
        # the context launches on its own (non-blocking) stream: torch's fill of `image` and the
        # collective's writes into the shards must have landed before its kernels touch them
        torch.cuda.synchronize()
    for r, shard in enumerate(shards):
        if ctx is not None and image.is_cuda:
            shard = shard.contiguous()
            ctx.unpack_tiles_device(shard.data_ptr(), r, world, image.data_ptr())
        else:
            img = image.numpy()
            unpack_tiles_host(shard.numpy(), r, world, img)
    return image
