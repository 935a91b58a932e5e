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


def _running_mean(samples):
    """pathtracer.cpp:200-217 in fp32: img = (1/it)*sample + ((it-1)/it)*img, it 1-based, first sample copied."""
    mean = samples[0].copy()
    for k in range(1, len(samples)):
        it = np.float32(k + 1)
        mean = (np.float32(1) / it) * samples[k] + (np.float32(k) / it) * mean
    return mean


def _sample_worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    from pathtracer_amd import dist as pd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    samples = np.random.default_rng(11).random((n_total, 40 * 24, 3), dtype=np.float32) * 4      # every rank: the same stack
    mine = samples[rank::world]                                    # hpt_set_sample_shard(rank, world)
    n_local = pd.sample_shard_count(0, n_total, rank, world)
    ok = n_local == len(mine)
    img = torch.from_numpy(_running_mean(mine))
    pd.reduce_samples(img, n_local, n_total)
    want = _running_mean(samples)
    ok = ok and bool(np.allclose(img.numpy(), want, rtol=1e-5, atol=1e-6))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sample_shard_reduce():
    """Sample-range shards: each rank's running mean over its iterations, one all-reduce of (n_r/N)-weighted
    images; 7 iterations over 2 ranks (4 + 3).  Tolerance 1e-5 relative: the summation order differs."""
    from pathtracer_amd import dist as pd
    assert [pd.sample_shard_count(0, 7, r, 2) for r in range(2)] == [4, 3]
    assert [pd.sample_shard_count(3, 10, r, 3) for r in range(3)] == [3, 2, 2]      # s = 3..9: {3,6,9}, {4,7}, {5,8}
    assert sum(pd.sample_shard_count(5, 1029, r, 8) for r in range(8)) == 1024
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sample_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
