"""CPU tests of the oracle (oracle/pt_oracle.c): hand-computed known answers for the
reference arithmetic it restates, plus the committed golden vectors.

PARITY STATUS: the reference has no tests/goldens and its kernels cannot run here
(SURVEY.md §8c), so the oracle is "parity unpinned" against a real reference run; the
KATs below pin each function to the reference's written formulas (file:line cited).
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import bits, get_scene
from oracle import binding as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def f3(*v):
    return np.array(v, np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def box(o, d, lo, hi, t=1000.0):
    return orc.lib().orc_intersect_box(_p(f3(*o)), _p(f3(*d)), _p(f3(*lo)), _p(f3(*hi)), t)


def tri(o, d, v1, v2, v3, t=1000.0):
    tr = np.zeros(1, orc.TRIANGLE)
    tr["v1"][0, :3] = v1
    tr["e1"][0, :3] = np.float32(v2) - np.float32(v1)
    tr["e2"][0, :3] = np.float32(v3) - np.float32(v1)
    tt = C.c_float(t)
    hit = orc.lib().orc_intersect_tri(_p(f3(*o)), _p(f3(*d)), _p(tr), C.byref(tt))
    return hit, tt.value


# ---- slab test, ray_kernel.cl:321-377 ----
def test_box_axis_hit_and_miss():
    assert box((0, 0, -10), (0, 0, 1), (-1, -1, -1), (1, 1, 1)) == 1
    assert box((0, 0, -10), (0, 0, -1), (-1, -1, -1), (1, 1, 1)) == 0       # behind: tmax < EPSILON
    assert box((2, 0, -10), (0, 0, 1), (-1, -1, -1), (1, 1, 1)) == 0       # x slab misses (inf comparisons)
    assert box((0, 0, -10), (0, 0, 1), (-1, -1, -1), (1, 1, 1), t=8.0) == 0  # tmin (9) >= t prunes
    assert box((0, 0, -10), (0, 0, 1), (-1, -1, -1), (1, 1, 1), t=9.5) == 1


def test_box_nan_on_slab_plane():
    # origin exactly on a face plane with dir.x = 0: (1-1)*inf = NaN.  Traced through
    # ray_kernel.cl:329-376 by hand: the early rejects compare false, the NaN is never replaced
    # (`tymax < NaN`, `tymin > NaN` are false), and the final (tmin < t) && (tmax > EPSILON) fails.
    assert box((1, 0, -10), (0, 0, 1), (-1, -1, -1), (1, 1, 1)) == 0      # NaN lands in tmax
    assert box((-1, 0, -10), (0, 0, 1), (-1, -1, -1), (1, 1, 1)) == 0     # NaN lands in tmin
    assert box((0.5, 0, -10), (0, 0, 1), (-1, -1, -1), (1, 1, 1)) == 1    # strictly inside the slab: -inf/+inf, kept
    # negative zero direction: 1/-0 = -inf -> dirIsNeg true (ray_kernel.cl:400)
    assert box((0, 0, -10), (-0.0, 0, 1), (-1, -1, -1), (1, 1, 1)) == 1


def test_box_inside_origin():
    assert box((0, 0, 0), (0.6, 0.0, 0.8), (-1, -1, -1), (1, 1, 1)) == 1      # tmin negative, tmax > EPSILON


# ---- Moller-Trumbore, ray_kernel.cl:467-520 ----
def test_tri_hit_t_value():
    hit, t = tri((0.25, 0.25, -5), (0, 0, 1), (0, 0, 0), (1, 0, 0), (0, 1, 0))
    assert hit == 1 and t == 5.0


def test_tri_edges_and_culls():
    assert tri((0, 0, -5), (0, 0, 1), (0, 0, 0), (1, 0, 0), (0, 1, 0))[0] == 1        # u = v = 0 is inside (u < 0 fails only)
    assert tri((1, 0, -5), (0, 0, 1), (0, 0, 0), (1, 0, 0), (0, 1, 0))[0] == 1        # u = 1 inside (u > 1 fails only)
    assert tri((0.6, 0.6, -5), (0, 0, 1), (0, 0, 0), (1, 0, 0), (0, 1, 0))[0] == 0    # u + v > 1
    assert tri((0.25, 0.25, 5), (0, 0, 1), (0, 0, 0), (1, 0, 0), (0, 1, 0))[0] == 0   # behind: tt < 0
    assert tri((0.25, 0.25, -5), (0, 0, 1), (0, 0, 0), (1, 0, 0), (0, 1, 0), t=5.0)[0] == 0   # strict tt < t
    assert tri((0.25, 0.25, -5), (1, 0, 0), (0, 0, 0), (1, 0, 0), (0, 1, 0))[0] == 0  # parallel: |det| < EPSILON
    # tiny triangle: |det| = 0.02*0.02 = 4e-4 < 1e-3 is culled
    assert tri((0.005, 0.005, -5), (0, 0, 1), (0, 0, 0), (0.02, 0, 0), (0, 0.02, 0))[0] == 0


# ---- sphere / plane, ray_kernel.cl:524-602 ----
def prim(kind, center, radius=1.0, normal=(0, 1, 0)):
    p = np.zeros(1, orc.PRIMITIVE)
    p["type"][0, 0] = kind
    p["center"][0, :3] = center
    p["plane_normal"][0, :3] = normal
    p["radius"][0] = radius
    return p


def iprim(o, d, p, t=1000.0):
    tt = C.c_float(t)
    return orc.lib().orc_intersect_prim(_p(f3(*o)), _p(f3(*d)), _p(p), C.byref(tt)), tt.value


def test_sphere():
    assert iprim((0, 0, -5), (0, 0, 1), prim(0, (0, 0, 0), 1.0)) == (1, 4.0)
    assert iprim((0, 0, 0), (0, 0, 1), prim(0, (0, 0, 0), 1.0)) == (1, 1.0)       # inside: t0 < EPSILON -> far root
    assert iprim((0, 2, -5), (0, 0, 1), prim(0, (0, 0, 0), 1.0))[0] == 0
    assert iprim((0, 0, -5), (0, 0, 1), prim(0, (0, 0, 0), 1.0), t=4.0)[0] == 0      # strict tt < t
    assert iprim((0, 0, -5), (0, 0, 1), prim(4, (0, 0, 0), 1.0))[0] == 0             # type 4 is never hit (pathtracer.cpp:340)


def test_plane():
    assert iprim((0, 5, 0), (0, -1, 0), prim(1, (0, 0, 0), normal=(0, 1, 0))) == (1, 5.0)
    assert iprim((0, 5, 0), (1, 0, 0), prim(1, (0, 0, 0), normal=(0, 1, 0)))[0] == 0   # |denom| <= EPSILON
    assert iprim((0, 5, 0), (0, 1, 0), prim(1, (0, 0, 0), normal=(0, 1, 0)))[0] == 0   # tt < 0
    assert iprim((0, 0, 0), (0, -1, 0), prim(1, (0, 0, 0), normal=(0, 1, 0))) == (1, 0.0)   # tt >= 0 accepted


# ---- RNG / sincos ----
def test_rand_range_and_determinism():
    L = orc.lib()
    v = np.array([L.orc_rand(0x5EED, p, s, c) for p in range(40) for s in range(5) for c in range(5)], np.float32)
    assert v.min() >= 0.0 and v.max() < 1.0
    assert len(np.unique(v)) > 0.99 * len(v)
    assert 0.4 < v.mean() < 0.6
    assert L.orc_rand(1, 2, 3, 4) == L.orc_rand(1, 2, 3, 4)
    assert L.orc_rand(1, 2, 3, 4) != L.orc_rand(1, 2, 3, 5)


def test_sincos_accuracy():
    L = orc.lib()
    s, c = C.c_float(), C.c_float()
    worst = 0.0
    for phi in np.linspace(0, 2 * np.pi, 20001, dtype=np.float64):
        ph = np.float32(phi)
        L.orc_sincos(float(ph), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - np.sin(np.float64(ph))), abs(c.value - np.cos(np.float64(ph))))
    assert worst < 2.5e-7     # <= ~2 ulp at |x| ~ 1 (OpenCL allows 4 ulp)


# ---- camera ray, initial_ray_kernel.cl:859-883 ----
def test_camera_ray_matches_formula():
    s, bvh, o = get_scene("cornell", width=64, height=64)
    w = h = 64
    for gid, it in [(0, 0), (63, 0), (64 * 32 + 32, 5), (64 * 63 + 1, 35), (100, 36)]:
        org, d = o.camera_ray(gid, it)
        assert tuple(org) == (0.0, 0.0, -200.0)
        r0 = np.float32(orc.lib().orc_rand(s["cfg"]["seed"], gid, it, 0))
        r1 = np.float32(orc.lib().orc_rand(s["cfg"]["seed"], gid, it, 1))
        f = np.float32
        x = f(gid % w) / f(w) - f(0.5)
        y = f(f(gid) / f(w)) / f(h) - f(0.5)          # y = gid/(float)width is NOT floored
        left = x - f(1) / f(2 * w)
        top = y - f(1) / f(2 * h)
        cell = it % 36
        x = left + (f(cell % 6) + r0) * (f(1) / f(6 * w))
        y = top + (f(cell // 6) + r1) * (f(1) / f(6 * h))
        v = np.array([x, y, 1.0], np.float32)              # x*u + y*v - n with the default camera
        inv = f(1) / np.sqrt(f(f(x * x + y * y) + f(1)))
        assert np.array_equal(bits(d), bits(v * inv))
        assert abs(np.linalg.norm(d.astype(np.float64)) - 1) < 1e-6


# ---- accumulation (pathtracer.cpp:200-217) and tonemap (png_writer.cpp:7-17) ----
def test_accumulate_running_mean():
    rng = np.random.default_rng(0)
    img = np.zeros(300, np.float32)
    samples = rng.random((5, 300), dtype=np.float32) * 3
    ref = np.zeros(300, np.float32)
    for it in range(1, 6):
        orc.accumulate(img, samples[it - 1], it)
        if it == 1:
            ref = samples[0].copy()
        else:
            i1 = np.float32(1) / np.float32(it)
            i2 = np.float32(it - 1) / np.float32(it)
            ref = i1 * samples[it - 1] + i2 * ref
        assert np.array_equal(bits(img), bits(ref))
    assert np.allclose(img, samples.mean(0), rtol=1e-5)


def test_tonemap():
    x = np.array([-1.0, 0.0, 0.5, 0.999, 1.0, 7.0, np.nan, 1.0 / 255, 0.00392], np.float32)
    assert list(orc.tonemap(x)) == [0, 0, 127, 254, 255, 255, 0, 1, 0]


# ---- BVH restatement (bvh.cpp) : structural invariants ----
@pytest.mark.parametrize("name,kw", [("cornell", {}), ("random", dict(n=2000, seed=1)),
                                     ("degenerate", dict(n_same=200, n_other=30))])
def test_bvh_invariants(name, kw):
    from pathtracer_amd import scenes
    s = scenes.scene(name, **kw)
    v = s["verts"]
    bbox, info, order = orc.build_bvh(v)
    n = len(v)
    assert sorted(order) == list(range(n))
    leaves = info[info["nPrimitives"] > 0]
    assert leaves["nPrimitives"].sum() == n and leaves["nPrimitives"].max() <= 24      # MAX_TRIANGLES_LEAF
    assert list(leaves["offset"]) == list(np.concatenate([[0], np.cumsum(leaves["nPrimitives"])[:-1]]))
    vo = v[order]
    for k in range(len(info)):
        if info["nPrimitives"][k] > 0:
            tv = vo[info["offset"][k]: info["offset"][k] + info["nPrimitives"][k]].reshape(-1, 3)
            assert np.array_equal(bbox["min"][k], tv.min(0)) and np.array_equal(bbox["max"][k], tv.max(0))
        else:
            c0, c1 = k + 1, info["secondChildOffset"][k]
            assert k < c0 < c1 < len(info) and 0 <= info["axis"][k] <= 2
            assert np.array_equal(bbox["min"][k], np.minimum(bbox["min"][c0], bbox["min"][c1]))
            assert np.array_equal(bbox["max"][k], np.maximum(bbox["max"][c0], bbox["max"][c1]))
            # "still split at mid" (bvh.cpp:325-326): subtree triangle counts differ by at most one
    if name == "random":
        assert len(info) == 2 * n - 1            # 2000 random tris -> 3999 nodes, single-triangle leaves


def test_bvh_needs_12_triangles():
    v = np.zeros((5, 3, 3), np.float32)
    with pytest.raises(ValueError):
        orc.build_bvh(v)


# ---- OBJ restatement (obj_reader.cpp) ----
def test_obj_reader(tmp_path):
    p = tmp_path / "a.obj"
    p.write_text("# c\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1.5\nvn 0 0 1\nf 1 2 3\nf 1 3 4\n")
    n, v = orc.load_obj(str(p))
    assert n == 6
    assert np.array_equal(v, np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 0], [0, 1, 0], [0, 0, 1.5]], np.float32))
    q = tmp_path / "b.obj"
    q.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/1/1 2/2/2 3/3/3\n")
    assert orc.load_obj(str(q))[0] == -2          # "File can't be read by our simple parser" (obj_reader.cpp:47-50)
    assert orc.load_obj(str(tmp_path / "missing.obj"))[0] == -1


# ---- golden vectors (generated by tests/golden/make_golden.py from this oracle) ----
def test_golden_scene_hashes():
    want = json.load(open(os.path.join(GOLD, "scene_hashes.json")))
    for name, kw in [("cornell", {}), ("random", dict(n=2000, seed=1)), ("degenerate", dict(n_same=200, n_other=30)),
                     ("bunny", {})]:
        s, bvh, o = get_scene(name, **kw)
        ob, oi, oo = orc.build_bvh(s["verts"])
        sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()   # noqa: E731
        assert sha(s["verts"]) == want[name]["verts"], "scene generator drifted"
        assert sha(ob) == want[name]["bbox"] and sha(oo) == want[name]["order"]
        assert sha(oi["nPrimitives"]) == want[name]["info_nprims"]
        assert sha(orc.make_triangles(s["verts"], oo)) == want[name]["triangles"]


@pytest.mark.parametrize("name,kw,iters", [("cornell", dict(width=64, height=64), (0, 7)),
                                           ("random", dict(n=2000, seed=1, width=64, height=64), (0, 41))])
def test_golden_primary_hits(name, kw, iters):
    g = np.load(os.path.join(GOLD, "%s_64_primary.npz" % name))
    s, bvh, o = get_scene(name, **kw)
    for it in iters:
        hit, t, _ = o.primary_hits(it)
        assert np.array_equal(hit, g["hit_%d" % it]) and np.array_equal(bits(t), g["tbits_%d" % it])
    assert (g["hit_0"] >= 0).sum() > 500            # the fixture does exercise triangles


def test_golden_radiance():
    g = np.load(os.path.join(GOLD, "cornell_32_radiance_4spp.npz"))
    s, bvh, o = get_scene("cornell", width=32, height=32)
    img, cnt = o.render(0, 4, n_threads=3)          # thread count must not matter
    assert np.array_equal(bits(img), g["bits"])
    assert [cnt[k] for k in sorted(cnt)] == list(g["counters"])


def test_golden_radiance_glossy():
    """C5's scene content (glossy triangle material, 64 mirror + 64 glass sphere primitives) at 48x27."""
    g = np.load(os.path.join(GOLD, "glossy_48x27_radiance_3spp.npz"))
    s, bvh, o = get_scene("glossy", width=48, height=27)
    img, cnt = o.render(0, 3, n_threads=2)
    assert np.array_equal(bits(img), g["bits"])
    assert [cnt[k] for k in sorted(cnt)] == list(g["counters"])
    hit, t, _ = o.primary_hits(0)
    assert np.array_equal(hit, g["hit_0"]) and np.array_equal(bits(t), g["tbits_0"])
    assert (g["hit_0"] <= -3).sum() > 20 and (g["hit_0"] >= 0).sum() > 200      # spheres and triangles are both hit


def test_oracle_brute_force_closest_hit():
    """The traversal order never changes WHICH hit is closest unless two hits tie: compare with an
    O(n) loop over all triangles where the minimum is unique."""
    s, bvh, o = get_scene("random", n=2000, seed=1, width=64, height=64)
    hit, t, _ = o.primary_hits(0)
    tris = bvh["triangles"]
    L = orc.lib()
    rng = np.random.default_rng(1)
    for gid in rng.choice(64 * 64, 60, replace=False):
        org, d = o.camera_ray(int(gid), 0)
        best, best_id = np.float32(1000.0), -1
        for i in range(len(s["primitives"])):
            tt = C.c_float(best)
            if L.orc_intersect_prim(_p(org), _p(d), _p(o.primitives[i:i + 1]), C.byref(tt)):
                best, best_id = np.float32(tt.value), -(2 + i)
        for i in range(len(tris)):
            tt = C.c_float(best)
            if L.orc_intersect_tri(_p(org), _p(d), _p(tris[i:i + 1]), C.byref(tt)):
                best, best_id = np.float32(tt.value), i
        assert best == t[gid]
        assert best_id == hit[gid]
