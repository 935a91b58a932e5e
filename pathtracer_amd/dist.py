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


class Exchange:
    """The exchange step of one backend: its slab buffers, allocated once, and the collectives in STREAM ORDER.

    With the RCCL backend and the backend launching on torch's current stream (`be.set_stream(
    torch.cuda.current_stream().cuda_stream)`), pack -> all_gather_into_tensor -> unpack needs no host
    synchronisation: the kernels are on the current stream, and torch orders its collective after the work already
    queued there and the stream's later work after the collective.  Any other arrangement (gloo rehearsal, a
    backend on a stream of its own) falls back to host synchronisation around the collective."""

    def __init__(self, backend, image, world, group=None):
        import torch
        import torch.distributed as dist
        self.backend, self.image, self.world, self.group = backend, image, world, group
        self.gloo = dist.get_backend(group) == "gloo"
        self.n = -1
        self._buffers()

    def _buffers(self):
        """(Re)allocates the slab buffers when the backend's shard, and with it the slab size, has changed."""
        import torch
        n = self.backend.slab_floats()
        if n == self.n:
            return
        self.n = n
        self.slab = torch.empty(n, dtype=torch.float32, device=self.image.device)
        self.slabs = torch.empty(n * self.world, dtype=torch.float32, device=self.image.device)
        self.host_parts = [torch.empty(n, dtype=torch.float32) for _ in range(self.world)] if self.gloo else None

    def _stream_ordered(self):
        import torch
        return (not self.gloo) and getattr(self.backend, "stream_handle", None) == torch.cuda.current_stream().cuda_stream

    def gather(self):
        """Tile shards: every rank's tiles of `image` -> the full image on every rank (ONE all-gather)."""
        import torch
        import torch.distributed as dist
        be = self.backend
        self._buffers()
        ordered = self._stream_ordered()
        be.pack_tiles(self.image.data_ptr(), self.slab.data_ptr())
        if self.gloo:                                   # rehearsal on a box with fewer GPUs than ranks: stage through the host
            be.synchronize()
            dist.all_gather(self.host_parts, self.slab.cpu(), group=self.group)
            self.slabs.copy_(torch.cat(self.host_parts))
            torch.cuda.current_stream().synchronize()
        else:
            if not ordered:
                be.synchronize()
            dist.all_gather_into_tensor(self.slabs, self.slab, group=self.group)
            if not ordered:
                torch.cuda.current_stream().synchronize()
        be.unpack_tiles(self.slabs.data_ptr(), self.image.data_ptr())
        if not ordered:
            be.synchronize()
        return self.image

    def reduce(self, n_local, n_total, frame=None):
        """Sample shards: this rank's running mean over its n_local samples -> the frame's mean (ONE all-reduce).
        In place by default; with `frame` (a tensor like image) the rank's own mean is kept -- what a progressive
        render needs, which goes on folding samples into it and reduces again every K iterations."""
        import torch
        out = self.image if frame is None else frame
        if self.gloo:
            self.backend.synchronize()
            host = self.image.cpu()
            reduce_samples(host, n_local, n_total, group=self.group)
            out.copy_(host)
            torch.cuda.current_stream().synchronize()
            return out
        ordered = self._stream_ordered()
        if not ordered:
            self.backend.synchronize()
        if frame is not None:
            frame.copy_(self.image)
        reduce_samples(out, n_local, n_total, group=self.group)
        if not ordered:
            torch.cuda.current_stream().synchronize()      # the backend's own stream may write the image next
        return out


class NativeExchange:
    """The same two exchanges through the library's own communicator (hpt_comm_init / hpt_gather_tiles /
    hpt_reduce_samples: RCCL called from C++, queued on the backend's stream, slab buffers owned by the context).
    torch.distributed only carries the 128-byte unique id from rank 0 to the others; with world == 1 it is not
    needed at all (a one-rank communicator: what the one-GPU box can execute of this path)."""

    def __init__(self, backend, image, world, rank=0, group=None):
        import torch
        self.backend, self.image, self.world, self.rank = backend, image, world, rank
        if world > 1:
            import torch.distributed as dist
            uid = backend.comm_unique_id() if rank == 0 else bytes(128)
            t = torch.frombuffer(bytearray(uid), dtype=torch.uint8).clone()
            on_device = dist.get_backend(group) == "nccl"
            if on_device:
                t = t.to(image.device)
            dist.broadcast(t, src=0, group=group)
            uid = bytes(t.cpu().numpy().tobytes())
        else:
            uid = backend.comm_unique_id()
        backend.comm_init(uid, rank, world)

    def gather(self):
        self.backend.gather_tiles(self.image.data_ptr())
        return self.image

    def reduce(self, n_local, n_total, frame=None):
        out = self.image if frame is None else frame
        self.backend.reduce_samples(self.image.data_ptr(), out.data_ptr(), n_local, n_total)
        return out


def gather_image(backend, image, world, group=None):
    """All-gather the shards of a device-resident image (torch tensor [h*w*3] float32 on the backend's device)
    into every rank's `image`.  No-op when world == 1.  The slab buffers are allocated once per backend."""
    if world == 1:
        return image
    ex = getattr(backend, "_exchange", None)
    # keyed on the tensor itself: the caching allocator may hand a later tensor (another size, another device) the same address
    if ex is None or ex.image is not image or ex.world != world or ex.group is not group:
        ex = backend._exchange = Exchange(backend, image, world, group)
    return ex.gather()


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
    CPU); the render must be complete in the collective's stream order (Exchange.reduce sees to that).  fp32 sum in rank order of the collective: within
    1e-5 relative of the unsharded running mean, not bit-identical to it."""
    import torch.distributed as dist
    image.mul_(float(n_local) / float(n_total))
    dist.all_reduce(image, op=dist.ReduceOp.SUM, group=group)
    return image
