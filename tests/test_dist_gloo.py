"""world_size-2 gloo test of the N>1 exchange: each rank holds only its tiles of the
image, one all-gather of the packed slabs rebuilds the full image on every rank."""
import os
import sys

import numpy as np
import pytest
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


def _device_worker(rank, world, port, w, h, spp, mode, q):
    """One rank of a 2-process render that shares ONE GPU: the device kernels hpt_pack_tiles / hpt_unpack_tiles
    (or the sample-shard fold) on both sides of a gloo collective, through dist.Exchange."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from pathtracer_amd import api, dist as pd, scenes
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    s = scenes.scene("cornell")
    bvh = api.build_bvh(s["verts"])
    be = api.Backend(w, h, device=0)
    be.load(dict(s, width=w, height=h), bvh)
    be.set_stream(torch.cuda.current_stream().cuda_stream)
    image = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
    ex = pd.Exchange(be, image, world)
    if mode == "tiles":
        be.set_shard(rank, world)
        be.render(0, spp, image.data_ptr())
        ex.gather(); ex.gather()                          # the buffers are reused: a second exchange changes nothing
    else:
        be.set_sample_shard(rank, world)
        be.render(0, spp, image.data_ptr())
        ex.reduce(pd.sample_shard_count(0, spp, rank, world), spp)
    be.synchronize()
    q.put((rank, image.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["tiles", "samples"])
def test_two_ranks_device_exchange_on_one_gpu(mode):
    """world_size 2 over gloo, both ranks on the one GPU of the box: each renders its shard through the C-ABI,
    packs with the DEVICE kernel, exchanges, unpacks with the device kernel.  Tile shards: bit-identical to the
    oracle's unsharded image on both ranks; sample shards: within 1e-5 (fp32 summation order)."""
    from conftest import bits, get_scene
    w, h, spp = 100, 52, 5
    _, _, o = get_scene("cornell", width=w, height=h)
    want = o.render(0, spp)[0].reshape(-1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_device_worker, args=(r, 2, port, w, h, spp, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in range(2):
        if mode == "tiles":
            assert np.array_equal(bits(res[r]), bits(want)), r
        else:
            assert np.allclose(res[r], want, rtol=1e-5, atol=1e-6), r
