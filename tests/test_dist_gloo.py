"""world_size-2 gloo test of the N>1 exchange: each rank holds only its tiles of the
image, one all-gather of the packed slabs rebuilds the full image on every rank."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, w, h, q):
    sys.path.insert(0, ROOT)
    from pathtracer_amd import dist as pd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = np.random.default_rng(7).random((w * h, 3), dtype=np.float32)       # what a 1-GPU render would hold
    mine = np.zeros_like(full)
    idx = pd.slab_pixel_index(w, h, rank, world)
    mine[idx[idx >= 0]] = full[idx[idx >= 0]]                                  # this rank rendered only its tiles
    slab = torch.from_numpy(pd.pack_tiles_np(mine, w, h, rank, world).reshape(-1))
    slabs = pd.gather_image_cpu(slab, world).numpy()
    out = pd.unpack_tiles_np(slabs, w, h, world)
    q.put((rank, bool(np.array_equal(out, full))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 100, 52, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
