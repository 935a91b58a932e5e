"""Multi-GPU tile sharding: one process per GPU, scene and BVH
replicated, rank r renders the 8x8-pixel tiles t with t % world == r, and the
disjoint slabs are exchanged with ONE all-gather (RCCL over xGMI when the
process group backend is "nccl"; gloo on CPU in the tests).

Every pixel's samples are folded on one GPU in iteration order, so the
gathered image is bit-identical for any world size.

The numpy functions mirror the device index math of work_to_pixel()
(csrc/render_kernels.h) and exist for CPU-side tests of the exchange.
"""
import numpy as np


def tile_grid(width, height):
    tx, ty = (width + 7) // 8, (height + 7) // 8
    return tx, ty, tx * ty


def slab_tiles(width, height, world):
    return (tile_grid(width, height)[2] + world - 1) // world


def slab_pixel_index(width, height, rank, world):
    """-> int64 array [slab_tiles*64]: row-major pixel index of each slab slot, -1 where the slot is padding."""
    tx, _, nt = tile_grid(width, height)
    w = np.arange(slab_tiles(width, height, world) * 64, dtype=np.int64)
    tile = (w >> 6) * world + rank
    k = w & 63
    px = (tile % tx) * 8 + (k & 7)
    py = (tile // tx) * 8 + (k >> 3)
    ok = (tile < nt) & (px < width) & (py < height)
    return np.where(ok, py * width + px, -1)


def pack_tiles_np(image, width, height, rank, world):
    """image [h*w,3] -> slab [slab_tiles*64,3] (zeros in padding slots)."""
    idx = slab_pixel_index(width, height, rank, world)
    slab = np.zeros((len(idx), 3), np.float32)
    ok = idx >= 0
    slab[ok] = image.reshape(-1, 3)[idx[ok]]
    return slab


def unpack_tiles_np(slabs, width, height, world, image=None):
    """slabs [world, slab_tiles*64, 3] -> image [h*w,3]."""
    if image is None:
        image = np.zeros((width * height, 3), np.float32)
    slabs = slabs.reshape(world, -1, 3)
    for r in range(world):
        idx = slab_pixel_index(width, height, r, world)
        ok = idx >= 0
        image[idx[ok]] = slabs[r][ok]
    return image


def gather_image(backend, image, world, group=None):
    """All-gather the shards of a device-resident image (torch tensor [h*w*3] float32 on the
    backend's device) into every rank's `image`.  No-op when world == 1."""
    if world == 1:
        return image
    import torch
    import torch.distributed as dist
    n = backend.slab_floats()
    slab = torch.empty(n, dtype=torch.float32, device=image.device)
    slabs = torch.empty(n * world, dtype=torch.float32, device=image.device)
    backend.pack_tiles(image.data_ptr(), slab.data_ptr())
    # the pack kernel runs on the backend's stream; the collective on torch's current stream
    backend.synchronize()
    if dist.get_backend(group) == "gloo":           # rehearsal on a box with fewer GPUs than ranks: stage through the host
        parts = [torch.empty(n, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(parts, slab.cpu(), group=group)
        slabs.copy_(torch.cat(parts))
    else:
        dist.all_gather_into_tensor(slabs, slab, group=group)
    torch.cuda.current_stream().synchronize()
    backend.unpack_tiles(slabs.data_ptr(), image.data_ptr())
    backend.synchronize()
    return image


def gather_image_cpu(slab, world, group=None):
    """gloo form used by the CPU tests: slab is a CPU torch tensor; returns [world*n]."""
    import torch
    import torch.distributed as dist
    out = [torch.empty_like(slab) for _ in range(world)]
    dist.all_gather(out, slab, group=group)
    return torch.cat(out)


# ---- sample-range shards (secondary mode): rank r renders the iterations s % world == r ----

def sample_shard_count(spp_begin, spp_end, rank, world):
    """Iterations of [spp_begin, spp_end) that rank `rank` of `world` renders (hpt_set_sample_shard)."""
    below = lambda s: 0 if s <= rank else (s - rank + world - 1) // world
    return below(spp_end) - below(spp_begin)


def reduce_samples(image, n_local, n_total, group=None):
    """Combines the per-rank running means of a sample-sharded render into the frame's mean, in place on every
    rank: image <- sum_r (n_r / N) * image_r, ONE all-reduce (RCCL when the group's backend is "nccl", gloo on
    CPU tensors in the tests).  `image` is this rank's mean over its n_local samples (a torch tensor, device or
    CPU); the backend's stream must have been synchronised.  fp32 sum in rank order of the collective: within
    1e-5 relative of the unsharded running mean, not bit-identical to it."""
    import torch.distributed as dist
    image.mul_(float(n_local) / float(n_total))
    dist.all_reduce(image, op=dist.ReduceOp.SUM, group=group)
    return image
