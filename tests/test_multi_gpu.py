"""The N > 1 path behind the boundary (SURVEY.md §8b "multi-GPU handled inside the context", §8e) and the entry points
that start it: bench.py / pathtracer_amd.render with --gpus N, the library's own communicator (RCCL called from C++:
hpt_comm_*, hpt_gather_tiles, hpt_reduce_samples), hpt_group (one process, several GPUs), the progressive
"reduce every K" mode of BASELINE config 5, and checkpoints.

What a one-GPU box can execute: two ranks that SHARE the GPU over gloo (the launcher, the shard arithmetic, the device
pack/unpack kernels), and RCCL with ONE rank (communicator set-up, the collectives' launch path in stream order).  RCCL
between two GPUs first runs on the driver's multi-GPU node; DESIGN.md §5 says so.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, bits, get_scene


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


# ------------------------------------------------------------------------------------------------ CPU: argument handling
def test_bench_refuses_a_world_size_other_than_gpus():
    """bench.py never prints an n_gpus other than --gpus: under a launcher whose WORLD_SIZE differs it stops, also for
    WORLD_SIZE=1 (round 2 let `--gpus 8` run on one GPU and print n_gpus 1)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HPT_DIST_BACKEND")}
    import torch
    if torch.cuda.device_count() < 2:          # no launcher, fewer GPUs than ranks, RCCL asked for: refused before any rank starts
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode != 0 and "--gpus 2 but this node has" in r.stderr


def test_group_and_comm_fail_loudly_without_a_device():
    import ctypes as C
    import torch
    from pathtracer_amd import api
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.HptError, match="no HIP device"):
        api.Group(8, 8, [0])
    L = api.lib()
    g = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    assert L.hpt_group_create(8, 8, devs, 2, C.byref(g)) == -1 and b"named twice" in L.hpt_group_last_error(None)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_bench_gpus_2_starts_two_ranks_itself():
    """`python3 bench.py --gpus 2 ...` with no launcher around it: bench.py starts two ranks under torch.distributed.run
    before it touches the GPU (here they share the one GPU over gloo), prints n_gpus 2, and the two shards trace exactly
    the rays of the one-GPU run of the same frame."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    args = ["--steps", "1", "--warmup", "0", "--spp", "2", "--no-cpu-baseline", "--no-iter"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args, capture_output=True, text=True,
                         timeout=600, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, capture_output=True, text=True,
                         timeout=600, env=dict(env, HPT_DIST_BACKEND="gloo"), cwd=ROOT)
    assert two.returncode == 0, two.stderr[-3000:]
    a, b = _line(one.stdout), _line(two.stdout)
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2
    assert b["config"]["rays_per_frame"] == a["config"]["rays_per_frame"] > 0
    # lane-loads: the ranks' sum, not rank 0's share standing for all (per launch = per rank: the sum / 2)
    la, lb = a["roofline"]["useful_lane_loads_per_launch"], b["roofline"]["useful_lane_loads_per_launch"]
    assert abs(2 * lb - la) <= 1e-3 * la
    assert list(a["roofline"])[:4] == ["bound", "achieved", "peak", "frac"]


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["torch", "native"])
def test_bench_one_rank_exchange_through_rccl(exchange):
    """The exchange step with ONE rank over RCCL, as bench.py runs it at N > 1: torch.distributed's nccl process group
    (all_gather_into_tensor in stream order -- Exchange._stream_ordered() is the branch taken) or the library's own
    communicator, progressive with an exchange every 2 iterations.  Rays equal the plain run's."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HPT_DIST_BACKEND")}
    env["MASTER_PORT"] = str(36500 + os.getpid() % 2000)
    args = ["--gpus", "1", "--steps", "1", "--warmup", "1", "--spp", "4", "--scene", "bunny", "--no-cpu-baseline", "--no-iter", "--no-count"]
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert plain.returncode == 0, plain.stderr[-3000:]
    for shard in ("tiles", "samples"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--force-exchange", "--exchange", exchange,
                           "--reduce-every", "2", "--shard", shard], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        line = _line(r.stdout)
        assert line["config"]["rays_per_frame"] == _line(plain.stdout)["config"]["rays_per_frame"]
        assert line["progressive"]["slices"] == 2 and line["progressive"]["exchange_ms_per_slice"] > 0


def _render(be, torch, spp0, spp1, image):
    be.render(spp0, spp1, image.data_ptr())
    be.synchronize()


@pytest.mark.gpu
def test_library_communicator_one_rank():
    """hpt_comm_unique_id / hpt_comm_init / hpt_gather_tiles / hpt_reduce_samples with a one-rank RCCL communicator:
    the gather leaves the frame bit-identical, the reduction multiplies by n_local / n_total = 1 and leaves it
    bit-identical too (out of place and in place); shard mismatches and missing communicators are refused."""
    import torch
    from pathtracer_amd import api
    s, bvh, o = get_scene("cornell", width=72, height=40)
    want = o.render(0, 3)[0]
    be = api.Backend(72, 40)
    be.load(s, bvh)
    img = torch.zeros(72 * 40 * 3, dtype=torch.float32, device="cuda")
    with pytest.raises(api.HptError, match="no communicator"):
        be.gather_tiles(img.data_ptr())
    be.comm_init(be.comm_unique_id(), 0, 1)
    with pytest.raises(api.HptError, match="already"):
        be.comm_init(be.comm_unique_id(), 0, 1)
    _render(be, torch, 0, 3, img)
    be.gather_tiles(img.data_ptr()); be.gather_tiles(img.data_ptr())
    be.synchronize()
    assert np.array_equal(bits(img.cpu().numpy().reshape(-1, 3)), bits(want))
    frame = torch.full_like(img, 7.0)
    be.reduce_samples(img.data_ptr(), frame.data_ptr(), 3, 3)
    be.reduce_samples(img.data_ptr(), img.data_ptr(), 3, 3)
    be.synchronize()
    assert np.array_equal(bits(frame.cpu().numpy().reshape(-1, 3)), bits(want))
    assert np.array_equal(bits(img.cpu().numpy().reshape(-1, 3)), bits(want))
    be.set_shard(0, 2)
    with pytest.raises(api.HptError, match="tile shard 0 of 2"):
        be.gather_tiles(img.data_ptr())
    be.set_shard(0, 1)
    be.set_sample_shard(1, 2)
    with pytest.raises(api.HptError, match="sample shard 1 of 2"):
        be.reduce_samples(img.data_ptr(), img.data_ptr(), 1, 3)
    be.comm_destroy()
    be.close()


@pytest.mark.gpu
def test_group_of_one_device_matches_hpt_render():
    """hpt_group over one device: ncclCommInitAll, a stream of its own, hpt_render + pack + ncclAllGather + unpack per
    render; the frame is the oracle's bit for bit, also continued, and in sample-shard mode (one all-reduce per slice)."""
    from pathtracer_amd import api
    s, bvh, o = get_scene("cornell", width=72, height=40)
    g = api.Group(72, 40, [0])
    assert g.n == 1
    g.load(s, bvh)
    g.render(0, 2)
    g.render(2, 5)
    assert np.array_equal(bits(g.read_image()), bits(o.render(0, 5)[0]))
    st = g.stats()
    assert st["rays_primary"] == 72 * 40 * 3                       # of the last call: iterations 2..4
    g.set_sharding(api.SHARD_SAMPLES)
    g.render(0, 3)
    g.render(3, 4)
    assert np.array_equal(bits(g.read_image()), bits(o.render(0, 4)[0]))      # one member: its mean times 4/4
    with pytest.raises(api.HptError, match="continue at iteration 4"):
        g.render(2, 6)
    g.set_sharding(api.SHARD_TILES)
    g.render(0, 1)
    assert np.array_equal(bits(g.read_image()), bits(o.render(0, 1)[0]))
    g.close()
    with pytest.raises(api.HptError, match="named twice"):
        api.Group(8, 8, [0, 0])
    # the C++ facade over a group: same images as over one context
    pt = api.Pathtracer(72, 40, devices=[0])
    pt.set_scene(s["primitives"], s["materials"]); pt.set_camera(s["camera"]); pt.set_config(s["cfg"])
    pt.set_triangles(s["verts"].reshape(-1, 3))
    img = pt.render(3)
    img = pt.render()
    assert pt.get_iterations() == 4
    assert np.array_equal(bits(img.reshape(-1, 3)), bits(o.render(0, 4)[0]))
    pt.close()


@pytest.mark.gpu
def test_checkpoint_resume_is_bit_identical(tmp_path):
    """hpt_checkpoint_save / hpt_checkpoint_load: a render stopped after 3 iterations and continued in a NEW context
    from the file equals the uninterrupted render bit for bit (tile shard 1 of 2 as well); a file of another config,
    shard or size, or a damaged one, is refused."""
    import torch
    from pathtracer_amd import api
    s, bvh, o = get_scene("cornell", width=72, height=40)
    want = o.render(0, 7)[0]
    path = str(tmp_path / "c.ckpt")
    for shard in ((0, 1), (1, 2)):
        be = api.Backend(72, 40); be.load(s, bvh); be.set_shard(*shard)
        img = torch.zeros(72 * 40 * 3, dtype=torch.float32, device="cuda")
        be.render(0, 3, img.data_ptr())
        be.checkpoint_save(img.data_ptr(), 3, path)
        be.close()
        be = api.Backend(72, 40); be.load(s, bvh); be.set_shard(*shard)
        img2 = torch.zeros_like(img)
        assert be.checkpoint_load(img2.data_ptr(), path) == 3
        _render(be, torch, 3, 7, img2)
        got = img2.cpu().numpy().reshape(-1, 3)
        if shard == (0, 1):
            assert np.array_equal(bits(got), bits(want))
        else:
            from pathtracer_amd import dist as pd
            idx = pd.slab_pixel_index(72, 40, 1, 2)
            idx = idx[idx >= 0]
            assert np.array_equal(bits(got[idx]), bits(want[idx]))
        # refused: another shard, another seed, another size, a flipped byte
        be.set_shard(0, 3 if shard == (0, 1) else 2)
        with pytest.raises(api.HptError, match="shard"):
            be.checkpoint_load(img2.data_ptr(), path)
        be.set_shard(*shard)
        be.set_config(dict(s["cfg"], seed=s["cfg"]["seed"] + 1))
        with pytest.raises(api.HptError, match="seed"):
            be.checkpoint_load(img2.data_ptr(), path)
        be.set_config(s["cfg"])
        raw = bytearray(open(path, "rb").read())
        raw[200] ^= 1
        open(path + ".bad", "wb").write(raw)
        with pytest.raises(api.HptError, match="checksum"):
            be.checkpoint_load(img2.data_ptr(), path + ".bad")
        open(path + ".short", "wb").write(raw[:1000])
        with pytest.raises(api.HptError, match="truncated"):
            be.checkpoint_load(img2.data_ptr(), path + ".short")
        small = api.Backend(64, 40)
        with pytest.raises(api.HptError, match="72x40"):
            small.checkpoint_load(img2.data_ptr(), path)
        small.close()
        be.close()


def _write_obj(path, verts):
    v = verts.reshape(-1, 3)
    with open(path, "w") as f:
        for x, y, z in v:
            f.write("v %.9g %.9g %.9g\n" % (x, y, z))
        for i in range(len(v) // 3):
            f.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))


@pytest.mark.gpu
def test_cli_gpus_and_checkpoint(tmp_path):
    """hpt_cli --gpus 1 (an hpt_group in one process: RCCL from C++), --reduce-every, and --checkpoint/--resume give the
    PNG of the plain one-device run, byte for byte."""
    from PIL import Image
    s, bvh, _ = get_scene("cornell", width=64, height=64)
    obj = tmp_path / "cornell.obj"
    _write_obj(obj, s["verts"])
    light = [repr(x) for x in (0.0, -52.0, 5.0, 16.0, 12.0)]
    cli = os.path.join(ROOT, "pathtracer_amd", "hpt_cli")
    base = [cli, "--obj", str(obj), "--w", "64", "--h", "64", "--depth", "4", "--light"] + light

    def run(extra, out):
        r = subprocess.run(base + extra + ["--out", str(tmp_path / out)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        return np.asarray(Image.open(str(tmp_path / out)).convert("RGB")), _line(r.stdout)

    want, _ = run(["--spp", "6"], "a.png")
    got, line = run(["--spp", "6", "--gpus", "1"], "b.png")
    assert line["gpus"] == 1 and np.array_equal(got, want)
    got, _ = run(["--spp", "6", "--gpus", "1", "--reduce-every", "4"], "c.png")
    assert np.array_equal(got, want)
    got, _ = run(["--spp", "6", "--devices", "0", "--shard", "samples", "--reduce-every", "2"], "d.png")
    assert np.array_equal(got, want)
    ck = str(tmp_path / "r.ckpt")
    _, line = run(["--spp", "4", "--checkpoint", ck, "--reduce-every", "2"], "e.png")
    assert line["spp"] == 4
    got, line = run(["--spp", "6", "--checkpoint", ck, "--resume"], "f.png")
    assert line["spp"] == 6 and np.array_equal(got, want)


@pytest.mark.gpu
def test_render_entry_point_gpus_2_progressive_and_resume(tmp_path):
    """python -m pathtracer_amd.render --gpus 2 with no launcher (it starts the two ranks; they share this GPU over
    gloo): progressive with an exchange every 2 iterations, stopped after 4 with per-rank checkpoints, resumed to 6 --
    the PNG equals the one-process render's byte for byte (tile shards)."""
    from PIL import Image
    s, bvh, _ = get_scene("cornell", width=64, height=64)
    obj = tmp_path / "cornell.obj"
    _write_obj(obj, s["verts"])
    light = [repr(x) for x in (0.0, -52.0, 5.0, 16.0, 12.0)]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PYTHONPATH=ROOT, HPT_DIST_BACKEND="gloo")
    base = [sys.executable, "-m", "pathtracer_amd.render", "--obj", str(obj), "--w", "64", "--h", "64", "--depth", "4", "--light"] + light

    def run(extra, out):
        r = subprocess.run(base + extra + ["--out", str(tmp_path / out)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        return np.asarray(Image.open(str(tmp_path / out)).convert("RGB")), _line(r.stdout)

    want, _ = run(["--spp", "6"], "a.png")
    ck = str(tmp_path / "p.ckpt")
    _, line = run(["--spp", "6", "--gpus", "2", "--reduce-every", "2", "--checkpoint", ck, "--stop-after", "4"], "b.png")
    assert line["gpus"] == 2 and line["spp"] == 4
    assert os.path.exists(ck + ".rank0") and os.path.exists(ck + ".rank1")
    got, line = run(["--spp", "6", "--gpus", "2", "--reduce-every", "2", "--checkpoint", ck, "--resume"], "c.png")
    assert line["spp"] == 6 and np.array_equal(got, want)
    got, _ = run(["--spp", "6", "--gpus", "2", "--reduce-every", "2", "--shard", "samples"], "d.png")
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1


_GROUP_REHEARSAL = r'''
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
from conftest import bits, get_scene
from pathtracer_amd import api
s, bvh, o = get_scene("cornell", width=100, height=52)          # 12.5 x 6.5 tiles: ragged right and bottom edges, padded slabs
for n in (2, 3):
    g = api.Group(100, 52, [0] * n)
    g.load(s, bvh)
    g.render(0, 2); g.render(2, 5)
    want = o.render(0, 5)[0]
    assert np.array_equal(bits(g.read_image()), bits(want)), ("tiles", n)
    st = g.stats()
    assert st["rays_primary"] == 100 * 52 * 3, st                        # the members' shards of the last call, summed
    import ctypes
    g.synchronize()
    for m in range(n):                                                   # EVERY member holds the whole frame
        host = np.empty((100 * 52, 3), np.float32)
        assert api.lib().hpt_read_image(api.lib().hpt_group_ctx(g.g, m), ctypes.c_void_p(g.frame(m)), host.ctypes.data_as(ctypes.c_void_p)) == 0
        assert np.array_equal(bits(host), bits(want)), ("tiles, member", m)
    g.set_sharding(api.SHARD_SAMPLES)
    g.render(0, 4); g.render(4, 7)
    assert np.allclose(g.read_image(), o.render(0, 7)[0], rtol=1e-5, atol=1e-6), ("samples", n)
    g.render(0, 1)                                                       # one sample: the members without one contribute zero
    assert np.allclose(g.read_image(), o.render(0, 1)[0], rtol=1e-5, atol=1e-6), ("samples, 1 spp", n)
    g.close()
print("group rehearsal ok")
'''


@pytest.mark.gpu
def test_group_of_several_members_sharing_the_gpu_over_a_stand_in_for_rccl(tmp_path):
    """hpt_group's loops over SEVERAL members -- shards, slab offsets, one stream per member, ncclGroupStart/End around the
    members' collectives -- executed on the one GPU: the members share it (HPT_GROUP_SHARED_DEVICE=1) and a test stand-in
    for RCCL (tests/native/fake_rccl.hip, HPT_RCCL_LIB) copies between their buffers.  Tile shards of 2 and 3 members:
    the oracle's frame bit for bit; sample shards: within 1e-5.  hpt_cli --devices 0,0: the one-device PNG byte for byte.
    Says nothing about RCCL's own data movement."""
    from PIL import Image
    fake = os.path.join(ROOT, "tests", "native", "libfake_rccl.so")
    assert os.path.exists(fake), "make fake_rccl"
    env = dict(os.environ, HPT_RCCL_LIB=fake, HPT_GROUP_SHARED_DEVICE="1")
    r = subprocess.run([sys.executable, "-c", _GROUP_REHEARSAL % dict(root=ROOT)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "group rehearsal ok" in r.stdout, r.stderr[-3000:]
    s, bvh, _ = get_scene("cornell", width=64, height=64)
    obj = tmp_path / "cornell.obj"
    _write_obj(obj, s["verts"])
    light = [repr(x) for x in (0.0, -52.0, 5.0, 16.0, 12.0)]
    base = [os.path.join(ROOT, "pathtracer_amd", "hpt_cli"), "--obj", str(obj), "--w", "64", "--h", "64", "--depth", "4", "--spp", "5", "--light"] + light
    imgs = []
    for extra in ([], ["--devices", "0,0", "--reduce-every", "2"]):
        out = tmp_path / ("g%d.png" % len(imgs))
        r = subprocess.run(base + extra + ["--out", str(out)], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        imgs.append(np.asarray(Image.open(str(out)).convert("RGB")))
    assert _line(r.stdout)["gpus"] == 2 and np.array_equal(imgs[0], imgs[1])
