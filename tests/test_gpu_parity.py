"""GPU parity tests: the HIP render path, called through the C-ABI (pathtracer_amd/libhpt.so),
against the CPU oracle on the same inputs.

Bar: BIT-EXACT.  Everything on the path is fp32 + - * / sqrt in a fixed order plus integer
hashing, so hit ids, t, every radiance sample and every running mean
must match the oracle bit for bit (NaNs compare equal to NaNs: conftest.bits).
"""
import numpy as np
import pytest
import torch

from conftest import bits, get_scene
from pathtracer_amd import api, dist, scenes

pytestmark = pytest.mark.gpu


def backend_for(s, bvh):
    be = api.Backend(s["width"], s["height"])
    be.load(s, bvh)
    return be


def dev_image(be, fill=None):
    t = torch.zeros(be.width * be.height * 3, dtype=torch.float32, device="cuda")
    if fill is not None:
        t.copy_(torch.from_numpy(np.ascontiguousarray(fill, np.float32).reshape(-1)))
    return t


def mismatch(a, b):
    a, b = np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)
    assert a.shape == b.shape
    return int((bits(a) != bits(b)).sum())


CASES = [("cornell", dict(width=256, height=256)),                  # C1 size, SURVEY.md §7.3
         ("random", dict(n=2000, seed=1, width=128, height=128)),
         ("random", dict(n=12, seed=3, width=64, height=64)),       # smallest tree the builder accepts
         ("degenerate", dict(n_same=200, n_other=30, width=96, height=96)),
         ("bunny", dict(width=160, height=160)),
         ("sponza", dict(width=192, height=108))]


@pytest.mark.parametrize("name,kw", CASES)
def test_primary_hits_bit_exact(name, kw):
    s, bvh, o = get_scene(name, **kw)
    be = backend_for(s, bvh)
    for it in (0, 17):
        hit, t = be.primary_hits(it)
        ohit, ot, _ = o.primary_hits(it)
        assert np.array_equal(hit, ohit), "%d hit ids differ" % (hit != ohit).sum()
        assert mismatch(t, ot) == 0
    assert (ohit >= 0).any()


@pytest.mark.parametrize("name,kw", CASES[:4] + [("bunny", dict(width=96, height=96)), ("sponza", dict(width=96, height=54))])
def test_radiance_sample_bit_exact(name, kw):
    """kernel_execute: one sample per pixel, every bounce, RNG draw order, Russian roulette."""
    s, bvh, o = get_scene(name, **kw)
    be = backend_for(s, bvh)
    for it in (0, 5, 36):
        img = be.kernel_execute(it)
        oimg, cnt = o.kernel_execute(it)
        assert mismatch(img, oimg) == 0
        st = be.stats()
        assert (st["rays_primary"], st["rays_bounce"], st["paths"]) == (cnt["rays_primary"], cnt["rays_bounce"], cnt["paths"])


def test_hip_path_against_committed_golden_vectors():
    """The committed fixtures of tests/golden/ (made by make_golden.py from the oracle; the reference holds none) against
    the HIP path directly: primary hits of two scenes, radiance of the Cornell scene, and C5's scene content."""
    import os
    gold = os.path.join(os.path.dirname(__file__), "golden")
    for name, kw, iters in [("cornell", dict(width=64, height=64), (0, 7)), ("random", dict(n=2000, seed=1, width=64, height=64), (0, 41))]:
        g = np.load(os.path.join(gold, "%s_64_primary.npz" % name))
        s, bvh, _ = get_scene(name, **kw)
        be = backend_for(s, bvh)
        for it in iters:
            hit, t = be.primary_hits(it)
            assert np.array_equal(hit, g["hit_%d" % it]) and np.array_equal(bits(t), g["tbits_%d" % it])
    for fixture, name, kw, spp in [("cornell_32_radiance_4spp", "cornell", dict(width=32, height=32), 4),
                                   ("glossy_48x27_radiance_3spp", "glossy", dict(width=48, height=27), 3)]:
        g = np.load(os.path.join(gold, fixture + ".npz"))
        s, bvh, _ = get_scene(name, **kw)
        be = backend_for(s, bvh)
        d = dev_image(be)
        be.render(0, spp, d.data_ptr()); be.synchronize()
        assert np.array_equal(bits(d.cpu().numpy()), g["bits"].reshape(-1))
        st = be.stats()
        cnt = dict(zip([str(k) for k in g["counter_names"]], [int(v) for v in g["counters"]]))
        assert (st["rays_primary"], st["rays_bounce"], st["paths"]) == (cnt["rays_primary"], cnt["rays_bounce"], cnt["paths"])
        if "hit_0" in g.files:
            hit, t = be.primary_hits(0)
            assert np.array_equal(hit, g["hit_0"]) and np.array_equal(bits(t), g["tbits_0"])


def test_device_image_helpers_without_the_hip_api():
    """hpt_image_alloc / hpt_read_image / hpt_image_free: the device-resident batched path for a caller that has only
    the C-ABI (no torch, no HIP runtime calls of its own)."""
    s, bvh, o = get_scene("cornell", width=64, height=48)
    be = backend_for(s, bvh)
    d = be.image_alloc()
    assert not be.read_image(d).any()                                    # zero-filled
    be.render(0, 3, d); be.render(3, 5, d)
    assert mismatch(be.read_image(d), o.render(0, 5)[0]) == 0
    be.first_bounce(); be.bounce(); be.reconstruct_paths(d)
    be.image_free(d)
    be.image_free(0)                                                     # NULL is a no-op
    with pytest.raises(api.HptError, match="read_image"):
        be.read_image(0)


def test_glossy_and_refractive_materials_bit_exact():
    """Triangle material with specular + IOR (the C5 material) and a transmissive triangle material."""
    for spec, ior, trans in [((0.7, 0.7, 0.2), 1.9, False), ((0, 0, 0), 1.5, True)]:
        s, bvh, o = get_scene("cornell", width=96, height=96)
        s = dict(s)
        s["materials"] = scenes.make_material((0.6, 0.4, 0.2), specular=spec, ior=ior, transmissive=trans)
        s["cfg"] = dict(s["cfg"], max_depth=15)
        from oracle import binding as orc
        o2 = orc.OracleScene(bvh["triangles"], bvh["bbox"], bvh["info"], s["primitives"], s["materials"], s["camera"], 96, 96, s["cfg"])
        be = backend_for(s, bvh)
        for it in (0, 3):
            assert mismatch(be.kernel_execute(it), o2.kernel_execute(it)[0]) == 0


def _camera(eye, n, u, v):
    cam = np.zeros((), api.CAMERA)
    cam["eye"] = (*eye, 0); cam["n"] = (*n, 0); cam["u"] = (*u, 0); cam["v"] = (*v, 0)
    return cam


@pytest.mark.parametrize("cam", [
    ((30.0, -20.0, -150.0), (-0.19, 0.13, -0.97), (0.98, 0.0, -0.19), (0.02, 0.99, 0.13)),      # off axis, rotated basis
    ((0.0, 0.0, -200.0), (0.0, 0.0, -1.0), (0.3, 0.0, 0.0), (0.0, 0.5, 0.0)),                  # narrow, anisotropic field of view
    ((5.0, 10.0, -20.0), (0.0, 0.0, -1.0), (2.5, 0.5, 0.0), (-0.5, 2.0, 0.0)),                 # inside the scene, wide, skewed
    ((80.0, 0.0, -200.0), (0.0, 0.0, -1.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)),                 # every ray (0,0,1): 1/0 = inf, (lo-o)*inf = NaN on the wall planes
    ((-200.0, -80.0, 80.0), (-1.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)),               # every ray (1,0,0), origin on two box planes
    ((0.0, 200.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1e-30), (1e-30, 0.0, 0.0)),                # (0,-1,0) plus denormal-sized offsets
])
def test_other_cameras_bit_exact(cam):
    """hpt_set_camera with cameras other than the reference's fixed one (pathtracer.cpp:224-263), including rays with
    exactly zero direction components -- reciprocals +-inf, plane distances NaN where the origin lies on a box
    plane -- so the packed slab test's NaN polarity is compared with the oracle's comparison form on real trees."""
    from oracle import binding as orc
    camera = _camera(*cam)
    for name, kw in [("cornell", dict(width=96, height=96)), ("random", dict(n=2000, seed=1, width=64, height=64))]:
        s, bvh, _ = get_scene(name, **kw)
        s = dict(s, camera=camera)
        o = orc.OracleScene(bvh["triangles"], bvh["bbox"], bvh["info"], s["primitives"], s["materials"], camera,
                            s["width"], s["height"], s["cfg"])
        be = backend_for(s, bvh)
        hit, t = be.primary_hits(2)
        oh, ot, _ = o.primary_hits(2)
        assert np.array_equal(hit, oh) and mismatch(t, ot) == 0
        d = dev_image(be)
        be.render(0, 3, d.data_ptr()); be.synchronize()
        assert mismatch(d.cpu().numpy(), o.render(0, 3)[0]) == 0


def test_running_mean_fused_and_stepwise():
    """hpt_render over [0,n) == n x hpt_render_sample == oracle render (pathtracer.cpp:200-217)."""
    s, bvh, o = get_scene("cornell", width=64, height=64)
    be = backend_for(s, bvh)
    oimg, _ = o.render(0, 6)
    d = dev_image(be)
    be.render(0, 6, d.data_ptr()); be.synchronize()
    assert mismatch(d.cpu().numpy(), oimg) == 0
    # split range: [0,2) then [2,6) continues the same running mean
    d2 = dev_image(be)
    be.render(0, 2, d2.data_ptr()); be.render(2, 6, d2.data_ptr()); be.synchronize()
    assert mismatch(d2.cpu().numpy(), oimg) == 0
    # host in/out form, one iteration per call
    img = np.zeros((64 * 64, 3), np.float32)
    for it in range(6):
        be.render_sample(it, img)
    assert mismatch(img, oimg) == 0
    # kernel_execute + hpt_reconstruct on device
    d3 = dev_image(be)
    for it in range(6):
        smp = dev_image(be, be.kernel_execute(it))
        be.reconstruct(d3.data_ptr(), smp.data_ptr(), 64 * 64, it + 1)
    be.synchronize()
    assert mismatch(d3.cpu().numpy(), oimg) == 0


def test_config_diffuse_scale_depth_and_rr():
    s, bvh, o = get_scene("random", n=2000, seed=1, width=64, height=64)
    from oracle import binding as orc
    for cfg in [dict(max_depth=15, rr_start=3, diffuse_scale=1.0, seed=7),     # the iterative kernels' cos factor
                dict(max_depth=2, rr_start=0, diffuse_scale=2.0, seed=0x5EED),
                dict(max_depth=30, rr_start=1, diffuse_scale=2.0, seed=123456789)]:
        s2 = dict(s, cfg=cfg)
        o2 = orc.OracleScene(bvh["triangles"], bvh["bbox"], bvh["info"], s["primitives"], s["materials"], s["camera"], 64, 64, cfg)
        be = backend_for(s2, bvh)
        assert mismatch(be.kernel_execute(2), o2.kernel_execute(2)[0]) == 0


@pytest.mark.parametrize("env", [dict(HPT_STACK_CAP="1"), dict(HPT_STACK_CAP="3", HPT_SHADE_THRESHOLD="1"),
                                 dict(HPT_SHADE_THRESHOLD="64", HPT_TRIS_PER_TRIP="1", HPT_NODE_REPEAT_MIN="65"),
                                 dict(HPT_TRIS_PER_TRIP="64", HPT_NODE_REPEAT_MIN="1", HPT_BLOCKS_PER_CU="1"),
                                 dict(HPT_TRI_WEIGHT="1"), dict(HPT_TRI_WEIGHT="256", HPT_SHADE_THRESHOLD="7"),
                                 dict(HPT_PAIR_FETCH="1", HPT_LEAF_RECORDS="singles"),       # the helper-lane NODE fetch of gigabyte trees
                                 dict(HPT_PAIR_FETCH="1", HPT_LEAF_RECORDS="singles", HPT_NODE_REPEAT_MIN="1", HPT_STACK_CAP="2"),
                                 # the LDS copy of the tree's top: none, the root alone, a partial level, more nodes than fit beside
                                 # the stacks at full occupancy (fewer blocks per CU then), and with the helper-lane fetch
                                 dict(HPT_LDS_TOP_NODES="0"), dict(HPT_LDS_TOP_NODES="1"), dict(HPT_LDS_TOP_NODES="45", HPT_STACK_CAP="10"),
                                 dict(HPT_LDS_TOP_NODES="1024", HPT_STACK_CAP="2"),
                                 dict(HPT_LDS_TOP_NODES="37", HPT_PAIR_FETCH="1", HPT_LEAF_RECORDS="singles")])
def test_scheduling_knobs_do_not_change_results(env, monkeypatch):
    """The wave scheduler's knobs (which step kind runs, how long, how much of the traversal stack sits in
    LDS before it overflows to HBM, how many nodes of the tree's top every block keeps in LDS) only reorder work
    between lanes or change where a node is read from: every image stays bit-exact.
    HPT_STACK_CAP=1 pushes almost every stack entry through the HBM overflow path."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for name, kw in [("sponza", dict(width=96, height=54)), ("random", dict(n=2000, seed=1, width=64, height=64))]:
        s, bvh, o = get_scene(name, **kw)
        be = backend_for(s, bvh)
        d = dev_image(be)
        be.render(0, 3, d.data_ptr()); be.synchronize()
        assert mismatch(d.cpu().numpy(), o.render(0, 3)[0]) == 0


@pytest.mark.parametrize("form", ["singles", "pairs"])
def test_leaf_record_forms_do_not_change_results(form, monkeypatch):
    """The triangle-record form (one or two triangles per record) is a per-tree layout choice of the backend,
    with its own instantiation of the traversal kernels: whole-path render, primary hits and the iterative
    pipeline stay bit-exact in either."""
    monkeypatch.setenv("HPT_LEAF_RECORDS", form)
    for name, kw in [("sponza", dict(width=96, height=54)), ("random", dict(n=2000, seed=1, width=64, height=64))]:
        s, bvh, o = get_scene(name, **kw)
        be = backend_for(s, bvh)
        d = dev_image(be)
        be.render(0, 3, d.data_ptr()); be.synchronize()
        assert mismatch(d.cpu().numpy(), o.render(0, 3)[0]) == 0
        hit, t = be.primary_hits(1)
        oh, ot, _ = o.primary_hits(1)
        assert np.array_equal(hit, oh) and mismatch(t, ot) == 0
        want = np.zeros((s["width"] * s["height"], 3), np.float32)
        d.zero_()
        be.first_bounce(); o.iter_first_bounce()
        for _ in range(4):
            be.bounce(); be.reconstruct_paths(d.data_ptr()); o.iter_bounce(); o.iter_reconstruct(want)
        be.synchronize()
        assert mismatch(d.cpu().numpy(), want) == 0


def test_sample_buffer_batches(monkeypatch):
    """A render whose samples do not fit the sample buffer is split into several launches of render_kernel,
    each folded in iteration order: same image.  1 MB holds 10 iterations of a 96x96 shard."""
    monkeypatch.setenv("HPT_SAMPLE_BUDGET_MB", "1")
    s, bvh, o = get_scene("cornell", width=96, height=96)
    be = backend_for(s, bvh)
    d = dev_image(be)
    be.render(0, 23, d.data_ptr()); be.synchronize()
    assert mismatch(d.cpu().numpy(), o.render(0, 23)[0]) == 0
    st = be.stats()
    assert st["paths"] == 23 * 96 * 96
    assert st["launches"] > 1
    # a render so long that not even one tile's iterations fit: one tile at a time, iterations in batches
    s2, bvh2, o2 = get_scene("cornell", width=16, height=24)
    be2 = backend_for(s2, bvh2)
    d2 = dev_image(be2)
    be2.render(0, 1500, d2.data_ptr()); be2.render(1500, 1503, d2.data_ptr()); be2.synchronize()
    assert mismatch(d2.cpu().numpy(), o2.render(0, 1503)[0]) == 0


@pytest.mark.parametrize("w,h,world", [(64, 64, 2), (100, 52, 3), (72, 40, 8)])
def test_tile_shards_reassemble_bit_exact(w, h, world):
    """Each shard renders only its 8x8 tiles; pack -> (all-gather) -> unpack gives the 1-GPU image."""
    s, bvh, o = get_scene("cornell", width=w, height=h)
    be = backend_for(s, bvh)
    full = dev_image(be)
    be.render(0, 3, full.data_ptr()); be.synchronize()
    full = full.cpu().numpy().reshape(-1, 3)
    assert mismatch(full, o.render(0, 3)[0]) == 0
    n = None
    slabs = []
    for r in range(world):
        be.set_shard(r, world)
        n = be.slab_floats()
        assert n == dist.slab_tiles(w, h, world) * 64 * 3
        d = dev_image(be)
        be.render(0, 3, d.data_ptr())
        slab = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
        be.pack_tiles(d.data_ptr(), slab.data_ptr()); be.synchronize()
        part = d.cpu().numpy().reshape(-1, 3)
        idx = dist.slab_pixel_index(w, h, r, world)
        own = np.zeros(w * h, bool); own[idx[idx >= 0]] = True
        assert mismatch(part[own], full[own]) == 0 and not part[~own].any()        # only its own tiles were touched
        assert np.array_equal(slab.cpu().numpy().reshape(-1, 3), dist.pack_tiles_np(part, w, h, r, world))
        slabs.append(slab)
    out = dev_image(be)
    be.unpack_tiles(torch.cat(slabs).data_ptr(), out.data_ptr()); be.synchronize()
    assert mismatch(out.cpu().numpy(), full) == 0


@pytest.mark.parametrize("world,budget", [(2, None), (3, "1")])
def test_sample_shards(world, budget, monkeypatch):
    """hpt_set_sample_shard: rank r renders the iterations s % world == r and folds them as its own running
    mean.  Each rank's image is bit-exact against the oracle's samples of those iterations folded the same
    way (also when the render is continued, and when it is cut into batches: 1 MB holds 10 iterations here);
    the (n_r/N)-weighted sum is the unsharded frame within 1e-5 relative (fp32 summation order differs)."""
    from pathtracer_amd import dist as pd
    if budget:
        monkeypatch.setenv("HPT_SAMPLE_BUDGET_MB", budget)
    s, bvh, o = get_scene("cornell", width=96, height=96)
    n_total, cut = 37, 5
    samples = [o.kernel_execute(k)[0].reshape(-1) for k in range(n_total)]
    be = backend_for(s, bvh)
    frame = np.zeros(96 * 96 * 3, np.float32)
    for r in range(world):
        be.set_sample_shard(r, world)
        d = dev_image(be)
        be.render(0, cut, d.data_ptr())                   # continued render: [0, cut) then [cut, n_total)
        be.render(cut, n_total, d.data_ptr()); be.synchronize()
        mine = samples[r::world]
        mean = mine[0].copy()
        for k in range(1, len(mine)):
            it = np.float32(k + 1)
            mean = (np.float32(1) / it) * mine[k] + (np.float32(k) / it) * mean
        assert mismatch(d.cpu().numpy(), mean) == 0, r
        assert pd.sample_shard_count(0, n_total, r, world) == len(mine)
        assert be.stats()["paths"] == pd.sample_shard_count(cut, n_total, r, world) * 96 * 96
        frame += np.float32(len(mine) / n_total) * d.cpu().numpy()
    want = o.render(0, n_total)[0].reshape(-1)
    assert np.allclose(frame, want, rtol=1e-5, atol=1e-6)
    be.set_sample_shard(0, 1)
    d = dev_image(be)
    be.render(0, n_total, d.data_ptr()); be.synchronize()
    assert mismatch(d.cpu().numpy(), want) == 0          # back to the whole range: bit-exact again
    assert mismatch(be.kernel_execute(4), samples[4]) == 0
    with pytest.raises(api.HptError, match="sample shard"):
        be.set_sample_shard(2, 2)


def test_tonemap_kernel():
    from oracle import binding as orc
    s, bvh, o = get_scene("cornell", width=64, height=64)
    be = backend_for(s, bvh)
    x = (np.random.default_rng(0).random(64 * 64 * 3, dtype=np.float32) * 1.5 - 0.25).astype(np.float32)
    x[:3] = (np.nan, 1.0, 0.999)
    d = dev_image(be, x)
    out = torch.zeros(64 * 64 * 3, dtype=torch.uint8, device="cuda")
    be.tonemap(d.data_ptr(), out.data_ptr(), 64 * 64); be.synchronize()
    assert np.array_equal(out.cpu().numpy(), orc.tonemap(x))


def test_single_leaf_tree_and_empty_tree():
    """Edge cases of the seam: a BVH that is one leaf (hand-made arrays), and no BVH at all."""
    from oracle import binding as orc
    s, bvh, _ = get_scene("cornell", width=48, height=48)
    tris = bvh["triangles"][:3].copy()
    v = np.concatenate([tris["v1"][:, :3], tris["v1"][:, :3] + tris["e1"][:, :3], tris["v1"][:, :3] + tris["e2"][:, :3]])
    bbox = np.zeros(1, api.BBOX); bbox["min"][0] = v.min(0); bbox["max"][0] = v.max(0)
    info = np.zeros(1, api.INFO); info["nPrimitives"][0] = 3
    o = orc.OracleScene(tris, bbox, info, s["primitives"], s["materials"], s["camera"], 48, 48, s["cfg"])
    be = api.Backend(48, 48)
    be.upload_bvh(bbox, info); be.upload_triangles(tris)
    be.set_scene(s["primitives"], s["materials"]); be.set_camera(s["camera"]); be.set_config(s["cfg"])
    hit, t = be.primary_hits(0)
    oh, ot, _ = o.primary_hits(0)
    assert np.array_equal(hit, oh) and mismatch(t, ot) == 0
    assert mismatch(be.kernel_execute(1), o.kernel_execute(1)[0]) == 0
    # no triangles: primitives only
    o0 = orc.OracleScene(tris[:0], bbox[:0], info[:0], s["primitives"], s["materials"], s["camera"], 48, 48, s["cfg"])
    be0 = api.Backend(48, 48)
    be0.set_scene(s["primitives"], s["materials"]); be0.set_camera(s["camera"]); be0.set_config(s["cfg"])
    assert mismatch(be0.kernel_execute(0), o0.kernel_execute(0)[0]) == 0


@pytest.mark.parametrize("form", ["singles", "pairs", None])
@pytest.mark.parametrize("n", [1, 2, 5, 32, 63, 64])
def test_big_and_tiny_leaves_either_upload_order(n, form, monkeypatch):
    """Leaves of 1..64 triangles (the traversal reads them as records of two, or of one when the tree's leaves
    are small: both forms forced, and the per-tree choice; the upload accepts up to 64 per leaf) as a one-leaf
    tree, triangles uploaded before the tree and the tree replaced afterwards."""
    from oracle import binding as orc
    if form:
        monkeypatch.setenv("HPT_LEAF_RECORDS", form)
    s, bvh, _ = get_scene("random", n=2000, seed=1, width=40, height=40)
    tris = bvh["triangles"][:n].copy()
    v = np.concatenate([tris["v1"][:, :3], tris["v1"][:, :3] + tris["e1"][:, :3], tris["v1"][:, :3] + tris["e2"][:, :3]])
    bbox = np.zeros(1, api.BBOX); bbox["min"][0] = v.min(0); bbox["max"][0] = v.max(0)
    info = np.zeros(1, api.INFO); info["nPrimitives"][0] = n
    o = orc.OracleScene(tris, bbox, info, s["primitives"], s["materials"], s["camera"], 40, 40, s["cfg"])
    be = api.Backend(40, 40)
    be.upload_triangles(tris); be.upload_bvh(bbox, info)            # triangles first
    be.set_scene(s["primitives"], s["materials"]); be.set_camera(s["camera"]); be.set_config(s["cfg"])
    assert mismatch(be.kernel_execute(0), o.kernel_execute(0)[0]) == 0
    be.upload_bvh(bvh["bbox"], bvh["info"]); be.upload_triangles(bvh["triangles"])      # then the full scene replaces it
    _, _, ofull = get_scene("random", n=2000, seed=1, width=40, height=40)
    assert mismatch(be.kernel_execute(0), ofull.kernel_execute(0)[0]) == 0
    hit, t = be.primary_hits(3)
    oh, ot, _ = ofull.primary_hits(3)
    assert np.array_equal(hit, oh) and mismatch(t, ot) == 0
    if n == 64:
        info["nPrimitives"][0] = 65
        with pytest.raises(api.HptError, match="limit"):
            be.upload_bvh(bbox, info)


def test_error_reporting():
    s, bvh, _ = get_scene("cornell", width=32, height=32)
    be = api.Backend(32, 32)
    with pytest.raises(api.HptError, match="set_camera"):
        be.kernel_execute(0)
    bad = bvh["info"].copy()
    bad["secondChildOffset"][0] = 10 ** 6
    with pytest.raises(api.HptError, match="out of range"):
        be.upload_bvh(bvh["bbox"], bad)
    loop = bvh["info"].copy()
    loop["secondChildOffset"][0] = 1                      # both children the same node
    with pytest.raises(api.HptError, match="twice"):
        be.upload_bvh(bvh["bbox"], loop)
    be.upload_bvh(bvh["bbox"], bvh["info"])
    be.upload_triangles(bvh["triangles"][:5])             # leaves point past the triangle array
    be.set_scene(s["primitives"], s["materials"]); be.set_camera(s["camera"])
    with pytest.raises(api.HptError, match="triangles are uploaded"):
        be.kernel_execute(0)
    with pytest.raises(api.HptError) as e:
        be.set_shard(3, 2)
    assert e.value.status == -1                           # HPT_ERR_INVALID
    with pytest.raises(api.HptError, match="max_depth"):
        be.set_config(dict(max_depth=0, rr_start=3, diffuse_scale=2.0, seed=1))
    with pytest.raises(api.HptError) as e:                # 2 x 480 GB of staging images: more than the GPU has
        api.Backend(200000, 200000)
    assert e.value.status == -3                           # HPT_ERR_NOMEM, and the partly built context is gone
    api.Backend(32, 32).close()                           # the device is still usable


def test_pathtracer_facade_matches_oracle():
    """The C++ facade with the reference's method names: set_triangles -> BVH build -> upload -> render()."""
    s, bvh, o = get_scene("cornell", width=64, height=64)
    pt = api.Pathtracer(64, 64)
    assert pt.set_triangles(s["verts"].reshape(-1, 3)) == 1
    assert pt.set_scene(s["primitives"], s["materials"]) == 1
    assert pt.set_camera() == 1                          # the reference's fixed camera == scenes.default_camera()
    assert pt.set_config(s["cfg"]) == 1
    oimg = o.render(0, 3)[0]
    for _ in range(3):
        img = pt.render()
    assert pt.get_iterations() == 3 and mismatch(img, oimg) == 0
    img = pt.render(2)                                   # batched continuation
    assert mismatch(img, o.render(3, 5, image=oimg.copy())[0]) == 0
    pt2 = api.Pathtracer(64, 64)
    pt2.set_triangles(s["verts"].reshape(-1, 3)); pt2.set_scene(s["primitives"], s["materials"]); pt2.set_camera(); pt2.set_config(s["cfg"])
    # live-loop interface: first bounce cached, one bounce per call, finished paths folded per pixel
    o.iter_first_bounce()
    want = np.zeros((64 * 64, 3), np.float32)
    pt2.iterative_render_first_bounce()
    for _ in range(6):
        pt2.iterative_render(); pt2.reconstruct()
        o.iter_bounce(); o.iter_reconstruct(want)
    assert mismatch(pt2.get_image(), want) == 0


@pytest.mark.parametrize("name,kw,cfg", [
    ("cornell", dict(width=64, height=64), None),
    ("random", dict(n=2000, seed=1, width=64, height=64), dict(max_depth=15, rr_start=3, diffuse_scale=1.0, seed=7)),
    ("glossy", dict(width=48, height=27), dict(max_depth=6, rr_start=1, diffuse_scale=1.0, seed=99)),
    ("cornell", dict(width=40, height=24), dict(max_depth=1, rr_start=3, diffuse_scale=2.0, seed=5)),
])
def test_iterative_pipeline_matches_oracle(name, kw, cfg):
    """hpt_first_bounce / hpt_bounce / hpt_reconstruct_paths against the oracle's restatement of the live
    frame loop with its intended semantics: after every call the running-mean image, the per-pixel path
    counters and the ray counters are bit-identical.  max_depth = 1: the cached bounce ends every path, so
    each call completes one path per pixel."""
    from oracle import binding as orc
    s, bvh, o = get_scene(name, **kw)
    if cfg is not None:
        s = dict(s, cfg=cfg)
        o = orc.OracleScene(bvh["triangles"], bvh["bbox"], bvh["info"], s["primitives"], s["materials"], s["camera"],
                            s["width"], s["height"], cfg)
    be = backend_for(s, bvh)
    d = dev_image(be)
    want = np.zeros((s["width"] * s["height"], 3), np.float32)
    with pytest.raises(api.HptError):
        be.bounce()                                       # before first_bounce
    be.first_bounce(); o.iter_first_bounce()
    for k in range(12):
        be.bounce(); o.iter_bounce()
        if k == 3:
            with pytest.raises(api.HptError):
                be.bounce()                               # contributions not folded yet
        be.reconstruct_paths(d.data_ptr()); o.iter_reconstruct(want)
        be.synchronize()
        assert mismatch(d.cpu().numpy(), want) == 0, k
        assert np.array_equal(be.path_counts(), o.iter_px["completed"]), k
    st = be.stats()
    assert st["rays_primary"] == o.iter_st.rays_primary and st["rays_bounce"] == o.iter_st.rays_bounce
    assert o.iter_px["completed"].max() >= 2              # some pixels restarted from the cache
    # restarting: same sequence again
    be.first_bounce(); o.iter_first_bounce()
    d.zero_(); want[:] = 0
    for k in range(3):
        be.bounce(); be.reconstruct_paths(d.data_ptr()); o.iter_bounce(); o.iter_reconstruct(want)
    be.synchronize()
    assert mismatch(d.cpu().numpy(), want) == 0


def test_iterative_pipeline_on_a_shard():
    """With a shard set the iterative calls touch this rank's tiles only; the union over ranks is the
    unsharded result."""
    s, bvh, o = get_scene("cornell", width=64, height=48)
    full = backend_for(s, bvh)
    a = dev_image(full)
    full.first_bounce()
    for _ in range(5):
        full.bounce(); full.reconstruct_paths(a.data_ptr())
    full.synchronize()
    acc = torch.zeros_like(a)
    counts = np.zeros(64 * 48, np.int32)
    for r in range(3):
        be = backend_for(s, bvh)
        be.set_shard(r, 3)
        d = dev_image(be)
        be.first_bounce()
        for _ in range(5):
            be.bounce(); be.reconstruct_paths(d.data_ptr())
        be.synchronize()
        acc += d                                          # shards are disjoint, the rest stays 0
        counts += be.path_counts()
    assert torch.equal(acc.view(torch.int32), a.view(torch.int32))
    assert np.array_equal(counts, full.path_counts())


def test_full_size_properties_c3():
    """BASELINE config C3 at full size (262 267 triangles, 1920x1080), checked through size-independent
    properties plus oracle spot checks: (1) shards reassemble to the unsharded image, (2) a split spp
    range continues the same running mean, (3) re-rendering is idempotent, (4) three pixel rows equal
    the oracle bit for bit, (5) ray counters equal paths + bounces."""
    s, bvh, o = get_scene("sponza")
    be = backend_for(s, bvh)
    w, h = s["width"], s["height"]
    a = dev_image(be)
    be.render(0, 2, a.data_ptr()); be.synchronize()
    st = be.stats()
    assert st["paths"] == 2 * w * h and st["rays_primary"] == st["paths"] and st["rays_bounce"] > st["paths"]
    b = dev_image(be)
    be.render(0, 1, b.data_ptr()); be.render(1, 2, b.data_ptr()); be.synchronize()
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    c = dev_image(be)
    slabs = []
    for r in range(4):
        be.set_shard(r, 4)
        d = dev_image(be)
        be.render(0, 2, d.data_ptr())
        slab = torch.empty(be.slab_floats(), dtype=torch.float32, device="cuda")
        be.pack_tiles(d.data_ptr(), slab.data_ptr()); be.synchronize()
        slabs.append(slab)
    be.unpack_tiles(torch.cat(slabs).data_ptr(), c.data_ptr()); be.synchronize()
    assert torch.equal(a.view(torch.int32), c.view(torch.int32))
    img = a.cpu().numpy().reshape(-1, 3)
    for row in (3, 540, 1077):
        oimg, _ = o.render(0, 2, px_begin=row * w, px_end=(row + 1) * w)
        assert mismatch(img[row * w:(row + 1) * w], oimg[row * w:(row + 1) * w]) == 0


@pytest.mark.parametrize("env", [dict(), dict(HPT_SHADE_THRESHOLD="1"), dict(HPT_SHADE_THRESHOLD="64", HPT_SAMPLE_BUDGET_MB="1")])
def test_c5_glossy_render_bit_exact(env, monkeypatch):
    """Config C5 (glossy sponza-class geometry, specular + IOR triangle material, 64 mirror and 64 glass sphere
    primitives: 129 analytic tests per ray) through hpt_render -- the path the bench times -- at reduced resolution,
    several iterations, continued across calls: Fresnel coin, mirror / refraction / total internal reflection,
    Russian roulette (pathtracer_kernel.cl:804-845; primitives ray_kernel.cl:524-602).  Run with the SHADE
    threshold the backend derives from the primitive count (40), with 1 and 64 forced, and with a sample buffer
    that cuts the render into launches over tile ranges."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    s, bvh, o = get_scene("glossy", width=160, height=90)
    assert len(s["primitives"]) == 129
    be = backend_for(s, bvh)
    d = dev_image(be)
    be.render(0, 3, d.data_ptr()); be.render(3, 10, d.data_ptr()); be.synchronize()
    st = be.stats()
    want, cnt = o.render(0, 10)
    assert mismatch(d.cpu().numpy(), want) == 0
    _, cnt2 = o.render(3, 10, image=o.render(0, 3)[0])
    assert (st["rays_primary"], st["rays_bounce"], st["paths"]) == (cnt2["rays_primary"], cnt2["rays_bounce"], cnt2["paths"])
    if "HPT_SAMPLE_BUDGET_MB" in env:
        assert st["launches"] > 1
    # the glass and mirror spheres are really hit, and paths really run long (RR region b > 3)
    hit, _ = be.primary_hits(0)
    assert (hit <= -3).sum() > 20 and cnt["rays_bounce"] > 2 * cnt["rays_primary"]


def test_c5_glossy_both_shard_modes():
    """C5 under the two multi-GPU partitions of SURVEY.md 8e: tile shards reassemble bit for bit; each sample
    shard equals the oracle's samples of its iterations folded as its own running mean, and the weighted sum is
    the frame within 1e-5."""
    from pathtracer_amd import dist as pd
    s, bvh, o = get_scene("glossy", width=96, height=54)
    w, h, n_it = 96, 54, 6
    want = o.render(0, n_it)[0].reshape(-1)
    be = backend_for(s, bvh)
    out = dev_image(be)
    slabs = []
    for r in range(4):
        be.set_shard(r, 4)
        d = dev_image(be)
        be.render(0, n_it, d.data_ptr())
        slab = torch.empty(be.slab_floats(), dtype=torch.float32, device="cuda")
        be.pack_tiles(d.data_ptr(), slab.data_ptr()); be.synchronize()
        slabs.append(slab)
    be.unpack_tiles(torch.cat(slabs).data_ptr(), out.data_ptr()); be.synchronize()
    assert mismatch(out.cpu().numpy(), want) == 0
    be.set_shard(0, 1)
    samples = [o.kernel_execute(k)[0].reshape(-1) for k in range(n_it)]
    frame = np.zeros(w * h * 3, np.float32)
    for r in range(2):
        be.set_sample_shard(r, 2)
        d = dev_image(be)
        be.render(0, n_it, d.data_ptr()); be.synchronize()
        mine = samples[r::2]
        mean = mine[0].copy()
        for k in range(1, len(mine)):
            it = np.float32(k + 1)
            mean = (np.float32(1) / it) * mine[k] + (np.float32(k) / it) * mean
        assert mismatch(d.cpu().numpy(), mean) == 0, r
        frame += np.float32(pd.sample_shard_count(0, n_it, r, 2) / n_it) * d.cpu().numpy()
    assert np.allclose(frame, want, rtol=1e-5, atol=1e-6)


def test_full_size_properties_c5():
    """BASELINE config C5 at full size (glossy, 129 primitives, 3840x2160), two samples per pixel through hpt_render:
    (1) a split spp range continues the same running mean, (2) eight tile shards reassemble to the unsharded image,
    (3) the two sample shards' weighted sum is the frame within 1e-5, (4) two pixel rows equal the oracle bit for
    bit, (5) counters: one primary ray per path, more bounce rays than paths."""
    s, bvh, o = get_scene("glossy")
    be = backend_for(s, bvh)
    w, h = s["width"], s["height"]
    assert (w, h) == (3840, 2160)
    a = dev_image(be)
    be.render(0, 2, a.data_ptr()); be.synchronize()
    st = be.stats()
    assert st["paths"] == 2 * w * h and st["rays_primary"] == st["paths"] and st["rays_bounce"] > st["paths"]
    b = dev_image(be)
    be.render(0, 1, b.data_ptr()); be.render(1, 2, b.data_ptr()); be.synchronize()
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    c = dev_image(be)
    slabs = []
    for r in range(8):
        be.set_shard(r, 8)
        d = dev_image(be)
        be.render(0, 2, d.data_ptr())
        slab = torch.empty(be.slab_floats(), dtype=torch.float32, device="cuda")
        be.pack_tiles(d.data_ptr(), slab.data_ptr()); be.synchronize()
        slabs.append(slab)
    be.unpack_tiles(torch.cat(slabs).data_ptr(), c.data_ptr()); be.synchronize()
    assert torch.equal(a.view(torch.int32), c.view(torch.int32))
    del slabs, c, b
    be.set_shard(0, 1)
    frame = torch.zeros_like(a)
    for r in range(2):
        be.set_sample_shard(r, 2)
        d = dev_image(be)
        be.render(0, 2, d.data_ptr()); be.synchronize()
        frame += 0.5 * d
    assert torch.allclose(frame, a, rtol=1e-5, atol=1e-6)
    img = a.cpu().numpy().reshape(-1, 3)
    for row in (900, 1300):
        oimg, _ = o.render(0, 2, px_begin=row * w, px_end=(row + 1) * w)
        assert mismatch(img[row * w:(row + 1) * w], oimg[row * w:(row + 1) * w]) == 0


def test_full_size_properties_c2():
    """BASELINE config C2 at full size (69 451 triangles, 1024x1024), three samples per pixel: split range, four tile
    shards, idempotence, and three pixel rows against the oracle."""
    s, bvh, o = get_scene("bunny")
    be = backend_for(s, bvh)
    w, h = s["width"], s["height"]
    assert (w, h) == (1024, 1024)
    a = dev_image(be)
    be.render(0, 3, a.data_ptr()); be.synchronize()
    st = be.stats()
    assert st["paths"] == 3 * w * h and st["rays_primary"] == st["paths"]
    b = dev_image(be)
    be.render(0, 2, b.data_ptr()); be.render(2, 3, b.data_ptr()); be.synchronize()
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    be.render(0, 3, b.data_ptr()); be.synchronize()                      # spp_begin == 0 overwrites: idempotent
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    c = dev_image(be)
    slabs = []
    for r in range(4):
        be.set_shard(r, 4)
        d = dev_image(be)
        be.render(0, 3, d.data_ptr())
        slab = torch.empty(be.slab_floats(), dtype=torch.float32, device="cuda")
        be.pack_tiles(d.data_ptr(), slab.data_ptr()); be.synchronize()
        slabs.append(slab)
    be.unpack_tiles(torch.cat(slabs).data_ptr(), c.data_ptr()); be.synchronize()
    assert torch.equal(a.view(torch.int32), c.view(torch.int32))
    img = a.cpu().numpy().reshape(-1, 3)
    for row in (100, 512, 800):
        oimg, _ = o.render(0, 3, px_begin=row * w, px_end=(row + 1) * w)
        assert mismatch(img[row * w:(row + 1) * w], oimg[row * w:(row + 1) * w]) == 0


def test_stats_per_stage_times_and_lane_loads(monkeypatch):
    """hpt_get_stats (SURVEY.md 8b "ms per stage"): HIP-event times of render_kernel, fold_kernel, the iterative
    trace / shade kernels and the reconstruct kernels, the number of launches a render was cut into, and (contexts
    created under HPT_COUNT_LANE_LOADS=1) the in-kernel count of 16-B lane-loads, which is a function of the work
    alone: the same for any schedule, and the counting kernel renders the same image."""
    s, bvh, o = get_scene("sponza", width=192, height=108)
    be = backend_for(s, bvh)
    d = dev_image(be)
    be.render(0, 6, d.data_ptr()); be.synchronize()
    st = be.stats()
    assert st["launches"] == 1 and st["last_kernel_ms"] > 0 and st["fold_ms"] > 0 and st["lane_loads"] == 0
    assert st["trace_ms"] == 0 and st["shade_ms"] == 0 and st["reconstruct_ms"] == 0
    sample = torch.rand(192 * 108 * 3, device="cuda")
    keep = d.clone()
    be.reconstruct(d.data_ptr(), sample.data_ptr(), 192 * 108, 5); be.synchronize()
    assert be.stats()["reconstruct_ms"] > 0
    be.first_bounce(); be.bounce(); be.reconstruct_paths(d.data_ptr()); be.synchronize()
    st = be.stats()
    assert st["trace_ms"] > 0 and st["shade_ms"] > 0 and st["reconstruct_ms"] > 0
    monkeypatch.setenv("HPT_COUNT_LANE_LOADS", "1")
    loads = []
    for budget in (None, "1"):
        if budget:
            monkeypatch.setenv("HPT_SAMPLE_BUDGET_MB", budget)
        be2 = backend_for(s, bvh)
        d2 = dev_image(be2)
        be2.render(0, 6, d2.data_ptr()); be2.synchronize()
        st2 = be2.stats()
        assert torch.equal(d2.view(torch.int32), keep.view(torch.int32))
        assert (st2["launches"] > 1) == bool(budget) and st2["last_kernel_ms"] > 0 and st2["fold_ms"] > 0
        rays = st2["rays_primary"] + st2["rays_bounce"]
        # global lane-loads: node visits below the LDS-resident top of the tree (4 each) and triangle records; C3-like scenes ~230 per ray
        assert 4 * st2["rays_primary"] <= st2["lane_loads"] <= 2000 * rays
        loads.append(st2["lane_loads"])
    assert loads[0] == loads[1]
    # without the LDS copy of the tree's top every node visit is four global lane-loads: more of them, the same image
    monkeypatch.setenv("HPT_LDS_TOP_NODES", "0")
    be3 = backend_for(s, bvh)
    d3 = dev_image(be3)
    be3.render(0, 6, d3.data_ptr()); be3.synchronize()
    assert torch.equal(d3.view(torch.int32), keep.view(torch.int32))
    assert be3.stats()["lane_loads"] > 1.1 * loads[0]


def test_full_size_properties_c4(monkeypatch):
    """BASELINE config C4 at full size (10 M triangles, 19.6 M reference nodes, 3840x2160), one sample per pixel:
    (1) the two triangle-record forms and an LDS stack of one entry (everything through the HBM overflow) give the
    same image bit for bit, (2) eight tile shards reassemble to it, (3) two pixel rows equal the oracle."""
    s, bvh, o = get_scene("tenm")
    w, h = s["width"], s["height"]
    images = []
    for env in (dict(HPT_LEAF_RECORDS="singles"), dict(HPT_LEAF_RECORDS="pairs", HPT_STACK_CAP="1")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        be = backend_for(s, bvh)
        a = dev_image(be)
        be.render(0, 1, a.data_ptr()); be.synchronize()
        assert be.stats()["paths"] == w * h
        images.append(a)
        monkeypatch.delenv("HPT_STACK_CAP", raising=False)
    assert torch.equal(images[0].view(torch.int32), images[1].view(torch.int32))
    c = dev_image(be)
    slabs = []
    for r in range(8):
        be.set_shard(r, 8)
        d = dev_image(be)
        be.render(0, 1, d.data_ptr())
        slab = torch.empty(be.slab_floats(), dtype=torch.float32, device="cuda")
        be.pack_tiles(d.data_ptr(), slab.data_ptr()); be.synchronize()
        slabs.append(slab)
    be.unpack_tiles(torch.cat(slabs).data_ptr(), c.data_ptr()); be.synchronize()
    assert torch.equal(images[0].view(torch.int32), c.view(torch.int32))
    img = images[0].cpu().numpy().reshape(-1, 3)
    for row in (700, 1500):
        oimg, _ = o.render(0, 1, px_begin=row * w, px_end=(row + 1) * w)
        assert mismatch(img[row * w:(row + 1) * w], oimg[row * w:(row + 1) * w]) == 0


def test_cli_obj_to_png_end_to_end(tmp_path):
    """The headless replacement of main.cpp: OBJ file -> load_obj -> BVH build -> upload -> render -> PNG, all in the
    C++ host side; the PNG must equal the oracle's image through the reference's 8-bit conversion."""
    import os
    import subprocess
    from PIL import Image
    from oracle import binding as orc
    from conftest import ROOT
    s, bvh, _ = get_scene("cornell", width=64, height=64)
    v = s["verts"].reshape(-1, 3)
    obj = tmp_path / "cornell.obj"
    with open(obj, "w") as f:
        f.write("# cornell-like test scene\n")
        for x, y, z in v:
            f.write("v %.9g %.9g %.9g\n" % (x, y, z))
        for i in range(len(v) // 3):
            f.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
    out = tmp_path / "o.png"
    light = (0.0, -52.0, 5.0, 16.0, 12.0)
    for extra in ([], ["--progressive"]):
        r = subprocess.run([os.path.join(ROOT, "pathtracer_amd", "hpt_cli"), "--obj", str(obj), "--w", "64", "--h", "64", "--spp", "3",
                            "--depth", "4", "--out", str(out), "--light"] + [repr(x) for x in light] + extra,
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        prims = scenes.sphere(light[:3], light[3], emissive=(light[4],) * 3)
        cfg = dict(s["cfg"], max_depth=4)
        o = orc.OracleScene(bvh["triangles"], bvh["bbox"], bvh["info"], prims, scenes.make_material((0.6, 0.4, 0.2)),
                            scenes.default_camera(), 64, 64, cfg)
        want = orc.tonemap(o.render(0, 3)[0]).reshape(64, 64, 3)
        got = np.asarray(Image.open(str(out)).convert("RGB"))
        assert np.array_equal(got, want)
    # the viewer's frame loop: first bounce cached, 7 x (one bounce; fold the finished paths)
    r = subprocess.run([os.path.join(ROOT, "pathtracer_amd", "hpt_cli"), "--obj", str(obj), "--w", "64", "--h", "64", "--live", "7",
                        "--depth", "4", "--out", str(out), "--light"] + [repr(x) for x in light],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    want = np.zeros((64 * 64, 3), np.float32)
    o.iter_first_bounce()
    for _ in range(7):
        o.iter_bounce(); o.iter_reconstruct(want)
    assert np.array_equal(np.asarray(Image.open(str(out)).convert("RGB")), orc.tonemap(want).reshape(64, 64, 3))


def test_render_entry_point_one_and_two_ranks(tmp_path):
    """python -m pathtracer_amd.render: OBJ -> PNG on one process, and on two processes started by
    torch.distributed.run that share this GPU (gloo rehearsal of the N>1 path).  Tile shards give the oracle's
    image byte for byte; sample shards give it within one 8-bit step (their sum is within 1e-5 relative)."""
    import os
    import subprocess
    import sys
    from PIL import Image
    from oracle import binding as orc
    from conftest import ROOT
    s, bvh, _ = get_scene("cornell", width=64, height=64)
    v = s["verts"].reshape(-1, 3)
    obj = tmp_path / "cornell.obj"
    with open(obj, "w") as f:
        for x, y, z in v:
            f.write("v %.9g %.9g %.9g\n" % (x, y, z))
        for i in range(len(v) // 3):
            f.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
    light = (0.0, -52.0, 5.0, 16.0, 12.0)
    prims = scenes.sphere(light[:3], light[3], emissive=(light[4],) * 3)
    o = orc.OracleScene(bvh["triangles"], bvh["bbox"], bvh["info"], prims, scenes.make_material((0.6, 0.4, 0.2)),
                        scenes.default_camera(), 64, 64, dict(s["cfg"], max_depth=4))
    want = orc.tonemap(o.render(0, 5)[0]).reshape(64, 64, 3).astype(np.int32)
    args = ["-m", "pathtracer_amd.render", "--obj", str(obj), "--w", "64", "--h", "64", "--spp", "5", "--depth", "4",
            "--light"] + [repr(x) for x in light]
    env = dict(os.environ, PYTHONPATH=ROOT, HPT_DIST_BACKEND="gloo")
    port = 30500 + (os.getpid() % 2000)
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1"]
    for k, (cmd, tol) in enumerate([([sys.executable] + args, 0),
                                    (launch + ["--master-port", str(port)] + args, 0),
                                    (launch + ["--master-port", str(port + 1)] + args + ["--shard", "samples"], 1)]):
        out = tmp_path / ("o%d.png" % k)
        r = subprocess.run(cmd + ["--out", str(out)], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        got = np.asarray(Image.open(str(out)).convert("RGB")).astype(np.int32)
        assert np.abs(got - want).max() <= tol, k


def test_plain_c_example_matches_oracle(tmp_path):
    """examples/minimal.c, built with gcc against libhpt.so: a procedural mesh through hpt_host_build_bvh, the
    uploads and hpt_render_sample (host buffers, the reference-shaped progressive call) to a PNG that equals the
    oracle's image of the same scene."""
    import math
    import os
    import subprocess
    from PIL import Image
    from oracle import binding as orc
    from conftest import ROOT
    exe = tmp_path / "minimal"
    lib_dir = os.path.join(ROOT, "pathtracer_amd")
    r = subprocess.run(["gcc", "-std=c11", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "minimal.c"),
                        "-L" + lib_dir, "-lhpt", "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib", "-lm", "-o", str(exe)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = tmp_path / "m.png"
    r = subprocess.run([str(exe), str(out), "96", "64", "5"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    # the same mesh, in fp32 exactly as the C program computes it (sinf/cosf of glibc on both sides)
    import ctypes
    m = ctypes.CDLL("libm.so.6")
    m.sinf.restype = m.cosf.restype = ctypes.c_float
    m.sinf.argtypes = m.cosf.argtypes = [ctypes.c_float]
    f32 = np.float32
    pi = f32(3.14159265358979)
    verts = []
    for ring in range(8):
        for seg in range(16):
            p = []
            for k in range(4):
                th = f32(f32(pi * f32(ring + (k >> 1))) / f32(8)); ph = f32(f32(f32(f32(2.0) * pi) * f32(seg + (k & 1))) / f32(16))
                st, ct, sp, cp = f32(m.sinf(th)), f32(m.cosf(th)), f32(m.sinf(ph)), f32(m.cosf(ph))
                p.append((f32(f32(f32(60.0) * st) * cp), f32(f32(60.0) * ct), f32(f32(f32(60.0) * st) * sp)))
            for t in ((0, 1, 2), (2, 1, 3)):
                verts.append([p[i] for i in t])
    verts = np.array(verts, np.float32)
    bvh = api.build_bvh(verts)
    light = scenes.sphere((40.0, -160.0, -60.0), 50.0, emissive=(12.0, 12.0, 12.0))
    o = orc.OracleScene(bvh["triangles"], bvh["bbox"], bvh["info"], light, scenes.make_material((0.6, 0.4, 0.2)),
                        scenes.default_camera(), 96, 64, dict(scenes.DEFAULT_CFG))
    want = orc.tonemap(o.render(0, 5)[0]).reshape(64, 96, 3)
    assert np.array_equal(np.asarray(Image.open(str(out)).convert("RGB")), want)


def _random_case(case):
    """Seeded random configuration: a triangle soup of 12..3000 triangles of varying extent and size, random analytic
    primitives (spheres and planes: diffuse, glossy, glass, emissive), random triangle material, camera, depth
    limit, Russian-roulette start, diffuse factor and RNG seed.  -> (scene dict, bvh, oracle scene)"""
    from oracle import binding as orc
    rng = np.random.default_rng(1000 + case)
    n = int(rng.choice([12, 13, 40, 300, 1500, 3000]))
    s = scenes.scene("random", n=n, seed=int(rng.integers(1, 1 << 20)), extent=float(rng.uniform(20, 120)),
                     size=float(rng.uniform(2, 60)), width=int(rng.integers(17, 80)), height=int(rng.integers(9, 70)))
    prims = [scenes.sphere(tuple(rng.uniform(-150, 150, 3)), float(rng.uniform(10, 60)), emissive=tuple(rng.uniform(4, 16, 3)))]
    # cases from 200 on: LONG primitive lists (the two-phase loop of the closest-hit query, render_kernels.h): 12 (the
    # switch-over length), 31..33 and 63..65 (the candidate masks' word and group boundaries), 129 (C5's), 140
    n_extra = int(rng.integers(0, 6)) if case < 200 else int(rng.choice([11, 12, 30, 31, 32, 62, 63, 64, 128, 139]))
    for _ in range(n_extra):
        kind = int(rng.integers(0, 4))
        c = tuple(rng.uniform(-90, 90, 3))
        if kind == 0:
            prims.append(scenes.sphere(c, float(rng.uniform(3, 30)), diffuse=tuple(rng.uniform(0, 1, 3))))
        elif kind == 1:
            prims.append(scenes.sphere(c, float(rng.uniform(3, 30)), diffuse=tuple(rng.uniform(0, 1, 3)),
                                       specular=tuple(rng.uniform(0, 1, 3)), ior=float(rng.uniform(1.1, 2.5))))
        elif kind == 2:
            prims.append(scenes.sphere(c, float(rng.uniform(3, 30)), ior=float(rng.uniform(1.1, 2.5)), transmissive=True))
        else:
            nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
            prims.append(scenes.plane(c, tuple(nrm), tuple(rng.uniform(0, 1, 3)), ptype=int(rng.choice([1, 1, 4]))))
    if case >= 200:
        # ties and degenerate discriminants: an exact duplicate of a sphere (the lower index must win), a sphere around the
        # camera (rays start inside: the t1 branch), a grazing-size sphere (discriminant near 0), a dead type in between
        k = int(rng.integers(1, len(prims)))
        prims.insert(int(rng.integers(1, len(prims))), prims[k].copy())
        prims[int(rng.integers(1, len(prims)))] = scenes.sphere((0.0, 0.0, -200.0), 400.0, diffuse=(0.5, 0.5, 0.5))
        prims[int(rng.integers(1, len(prims)))] = scenes.sphere(tuple(rng.uniform(-30, 30, 3)), 1e-3, diffuse=(1, 0, 0))
        prims[int(rng.integers(1, len(prims)))] = scenes.plane((0, 0, 0), (0, 1, 0), (0.3, 0.3, 0.3), ptype=7)
    mat_kind = int(rng.integers(0, 3))
    mats = scenes.make_material(tuple(rng.uniform(0, 1, 3)),
                                specular=tuple(rng.uniform(0, 1, 3)) if mat_kind == 1 else (0, 0, 0),
                                ior=float(rng.uniform(1.0, 2.2)), transmissive=mat_kind == 2)
    eye = rng.uniform(-60, 60, 3) + np.array([0, 0, -200.0])
    nvec = -(np.zeros(3) - eye) / np.linalg.norm(eye)          # n points from the target to the eye
    up = np.array([0.0, 1.0, 0.0])
    u = np.cross(up, -nvec); u /= np.linalg.norm(u)
    v = np.cross(-nvec, u)
    camera = _camera(tuple(eye), tuple(nvec), tuple(u * rng.uniform(0.4, 1.5)), tuple(v * rng.uniform(0.4, 1.5)))
    cfg = dict(max_depth=int(rng.integers(1, 21)), rr_start=int(rng.integers(0, 6)), diffuse_scale=float(rng.choice([1.0, 2.0, 0.7])),
               seed=int(rng.integers(0, 1 << 32)))
    s = dict(s, primitives=np.concatenate(prims), materials=mats, camera=camera, cfg=cfg)
    bvh = api.build_bvh(s["verts"])
    o = orc.OracleScene(bvh["triangles"], bvh["bbox"], bvh["info"], s["primitives"], s["materials"], camera,
                        s["width"], s["height"], cfg)
    return s, bvh, o


@pytest.mark.parametrize("case", range(24))
def test_randomized_scenes_bit_exact(case):
    """Twenty-four random configurations (_random_case): primary hits, two radiance samples and a 4-sample running
    mean against the oracle, bit for bit."""
    s, bvh, o = _random_case(case)
    be = backend_for(s, bvh)
    hit, t = be.primary_hits(1)
    oh, ot, _ = o.primary_hits(1)
    assert np.array_equal(hit, oh) and mismatch(t, ot) == 0
    for it in (0, 37):
        assert mismatch(be.kernel_execute(it), o.kernel_execute(it)[0]) == 0
    d = dev_image(be)
    be.render(0, 4, d.data_ptr()); be.synchronize()
    assert mismatch(d.cpu().numpy(), o.render(0, 4)[0]) == 0


@pytest.mark.parametrize("case", range(100, 108))
def test_randomized_iterative_pipeline_bit_exact(case):
    """Eight random configurations through the live frame loop (hpt_first_bounce, then bounce + reconstruct calls):
    image, per-pixel path counters and ray counters equal the oracle's after every call."""
    s, bvh, o = _random_case(case)
    be = backend_for(s, bvh)
    d = dev_image(be)
    want = np.zeros((s["width"] * s["height"], 3), np.float32)
    be.first_bounce(); o.iter_first_bounce()
    for k in range(9):
        be.bounce(); o.iter_bounce()
        be.reconstruct_paths(d.data_ptr()); o.iter_reconstruct(want)
        be.synchronize()
        assert mismatch(d.cpu().numpy(), want) == 0, k
        assert np.array_equal(be.path_counts(), o.iter_px["completed"]), k
    st = be.stats()
    assert st["rays_primary"] == o.iter_st.rays_primary and st["rays_bounce"] == o.iter_st.rays_bounce


@pytest.mark.parametrize("case", range(200, 212))
def test_randomized_long_primitive_lists_bit_exact(case):
    """Scenes with 13..141 analytic primitives (spheres of every material, planes of type 1, dead types, an exact
    duplicate, a sphere around the camera, a near-zero-radius sphere): the closest-hit query walks long lists in two
    phases (candidate bits, then each lane its candidates in index order) and must give the reference's sequential
    result -- hit ids, t bits, radiance, ray counters -- through hpt_render, hpt_kernel_execute and the iterative calls."""
    s, bvh, o = _random_case(case)
    assert len(s["primitives"]) >= 12
    be = backend_for(s, bvh)
    hit, t = be.primary_hits(1)
    oh, ot, _ = o.primary_hits(1)
    assert np.array_equal(hit, oh) and mismatch(t, ot) == 0
    img = be.kernel_execute(2)
    oimg, cnt = o.kernel_execute(2)
    assert mismatch(img, oimg) == 0
    st = be.stats()
    assert (st["rays_primary"], st["rays_bounce"], st["paths"]) == (cnt["rays_primary"], cnt["rays_bounce"], cnt["paths"])
    d = dev_image(be)
    be.render(0, 3, d.data_ptr()); be.synchronize()
    assert mismatch(d.cpu().numpy(), o.render(0, 3)[0]) == 0
    if case % 3 == 0:
        want = np.zeros((s["width"] * s["height"], 3), np.float32)
        d = dev_image(be)
        be.first_bounce(); o.iter_first_bounce()
        for _ in range(3):
            be.bounce(); be.reconstruct_paths(d.data_ptr()); o.iter_bounce(); o.iter_reconstruct(want)
        be.synchronize()
        assert mismatch(d.cpu().numpy(), want) == 0
