"""Procedural stand-in scenes for the five BASELINE.json configs.

The reference has no scene assets; geometry comes from scenes/libhpt_scenes.so
(deterministic C generator).  Primitives/materials/camera follow the shapes of
Pathtracer::set_scene()/set_camera() (reference pathtracer.cpp:224-263, 322-382):
one material for all triangles, a few analytic primitives, one of which is the
emissive sphere that lights the scene (triangles cannot emit).
"""
import ctypes as C
import os

import numpy as np

from .types import CAMERA, MATERIAL, PRIMITIVE

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(_HERE), "scenes", "libhpt_scenes.so")
        if not os.path.exists(path):
            raise RuntimeError("scenes/libhpt_scenes.so missing: run `make scenes` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.scn_generate.restype = C.c_int64
        L.scn_generate.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_int64]
        _LIB = L
    return _LIB


def generate(scene_id, p0=0.0, p1=0.0, p2=0.0, p3=0.0):
    """-> (n,3,3) float32 triangle vertices."""
    L = _lib()
    n = L.scn_generate(scene_id, p0, p1, p2, p3, None, 0)
    if n < 0:
        raise ValueError("unknown scene id %r" % scene_id)
    v = np.zeros((n, 3, 3), np.float32)
    m = L.scn_generate(scene_id, p0, p1, p2, p3, v.ctypes.data_as(C.c_void_p), n)
    assert m == n
    return v


def default_camera():
    """set_camera(): eye (0,0,-200) looking at the origin; n=(0,0,-1), u=(1,0,0), v=(0,1,0)
    after the sign flips (pathtracer.cpp:228-260)."""
    cam = np.zeros((), CAMERA)
    cam["eye"] = (0, 0, -200, 0)
    cam["n"] = (0, 0, -1, 0)
    cam["u"] = (1, 0, 0, 0)
    cam["v"] = (0, 1, 0, 0)
    return cam


def make_material(diffuse=(0.6, 0.4, 0.2), specular=(0, 0, 0), ior=1.0, transmissive=False):
    """create_material() + overrides (pathtracer.cpp:28-35, 327-328)."""
    m = np.zeros(1, MATERIAL)
    m["refractive"][0] = (ior, 0, 0, 1.0 if transmissive else 0.0)
    m["specular"][0] = (*specular, 0)
    m["diffuse"][0] = (*diffuse, 0)
    return m


def sphere(center, radius, diffuse=(0, 0, 0), specular=(0, 0, 0), emissive=None, ior=1.0, transmissive=False):
    """create_primitive() + a type-0 sphere (pathtracer.cpp:10-25, 352-371)."""
    p = np.zeros(1, PRIMITIVE)
    p["refractive"][0] = (ior, 0, 0, 1.0 if transmissive else 0.0)
    p["specular"][0] = (*specular, 0)
    p["diffuse"][0] = (*diffuse, 0)
    if emissive is not None:
        p["emissive"][0] = (*emissive, emissive[0])
    p["center"][0] = (*center, 0)
    p["type"][0] = (0, 0, 0, 0)
    p["radius"][0] = radius
    return p


def plane(center, normal, diffuse, ptype=1):
    p = np.zeros(1, PRIMITIVE)
    p["refractive"][0] = (1, 0, 0, 0)
    p["diffuse"][0] = (*diffuse, 0)
    p["center"][0] = (*center, 0)
    p["plane_normal"][0] = (*normal, 0)
    p["type"][0] = (ptype, 0, 0, 0)
    p["radius"][0] = 1.0
    return p


DEFAULT_CFG = {"max_depth": 15, "rr_start": 3, "diffuse_scale": 2.0, "seed": 0x5EED}


def _cfg(**kw):
    c = dict(DEFAULT_CFG)
    c.update(kw)
    return c


def _lcg(state):
    state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
    return state, ((state >> 8) & 0xFFFFFF) / 16777216.0


def scene(name, **kw):
    """-> dict(verts, primitives, materials, camera, width, height, spp, cfg, label)."""
    if name == "cornell":       # C1: ~32 tris, 256x256, 16 spp, 4 bounces
        prims = np.concatenate([
            sphere((0, -52, 5), 16.0, emissive=(12, 12, 12)),
            sphere((38, 62, -45), 18.0, diffuse=(0.7, 0.7, 0.2), specular=(0.7, 0.7, 0.2), ior=1.9),
            sphere((-45, 64, -50), 16.0, ior=1.5, transmissive=True),
            plane((0, 8, 0), (0, -1, 0), (0.6, 0.6, 0.6), ptype=4),     # dead plane, as in set_scene()
        ])
        return dict(verts=generate(1), primitives=prims, materials=make_material((0.7, 0.65, 0.6)),
                    camera=default_camera(), width=256, height=256, spp=16, cfg=_cfg(max_depth=4),
                    label="C1 cornell-like 32 tris 256x256 16spp depth4")
    if name == "bunny":         # C2: ~69k tris, 1024x1024, 64 spp diffuse
        prims = np.concatenate([sphere((-160, -260, -120), 90.0, emissive=(14, 13, 11))])
        return dict(verts=generate(2), primitives=prims, materials=make_material((0.6, 0.4, 0.2)),
                    camera=default_camera(), width=1024, height=1024, spp=64, cfg=_cfg(),
                    label="C2 bunny-class 69451 tris 1024x1024 64spp diffuse")
    if name in ("sponza", "glossy"):   # C3: ~262k tris 1920x1080 256 spp 8 bounces; C5: glossy variant at 4K
        prims = [sphere((0, -420, 0), 150.0, emissive=(16, 15, 13))]
        if name == "sponza":
            return dict(verts=generate(3), primitives=np.concatenate(prims), materials=make_material((0.6, 0.4, 0.2)),
                        camera=default_camera(), width=1920, height=1080, spp=256, cfg=_cfg(max_depth=8),
                        label="C3 sponza-class 262267 tris 1920x1080 256spp depth8")
        st = 4
        for k in range(128):
            st, a = _lcg(st); st, b = _lcg(st); st, c = _lcg(st); st, r = _lcg(st)
            centre = (-130 + 260 * a, 50 - 100 * b, -70 + 140 * c)
            if k < 64:
                prims.append(sphere(centre, 3 + 5 * r, diffuse=(0.7, 0.7, 0.7), specular=(0.9, 0.9, 0.9), ior=1.9))
            else:
                prims.append(sphere(centre, 3 + 5 * r, ior=1.5, transmissive=True))
        return dict(verts=generate(3), primitives=np.concatenate(prims),
                    materials=make_material((0.6, 0.4, 0.2), specular=(0.7, 0.7, 0.2), ior=1.9),
                    camera=default_camera(), width=3840, height=2160, spp=1024, cfg=_cfg(),
                    label="C5 glossy sponza-class + 128 mirror/glass spheres 3840x2160 1024spp")
    if name == "tenm":          # C4: ~10M tris, 3840x2160, 64 spp
        copies = int(kw.get("copies", 10000))
        prims = np.concatenate([sphere((0, -700, -100), 200.0, emissive=(14, 14, 14))])
        return dict(verts=generate(4, copies), primitives=prims, materials=make_material((0.6, 0.4, 0.2)),
                    camera=default_camera(), width=3840, height=2160, spp=64, cfg=_cfg(),
                    label="C4 %d x 1000-tri noise spheres 3840x2160 64spp" % copies)
    if name == "random":        # parity-test scene: n random triangles
        n, seed = int(kw.get("n", 2000)), int(kw.get("seed", 1))
        prims = np.concatenate([sphere((0, -150, -40), 40.0, emissive=(10, 10, 10)),
                                sphere((30, 20, -90), 14.0, diffuse=(0.7, 0.7, 0.2), specular=(0.7, 0.7, 0.2), ior=1.9),
                                sphere((-35, 25, -95), 12.0, ior=1.5, transmissive=True),
                                plane((0, 95, 0), (0, -1, 0), (0.6, 0.6, 0.6), ptype=1)])
        return dict(verts=generate(10, n, seed, kw.get("extent", 70.0), kw.get("size", 25.0)), primitives=prims,
                    materials=make_material((0.6, 0.4, 0.2)), camera=default_camera(),
                    width=int(kw.get("width", 128)), height=int(kw.get("height", 128)), spp=int(kw.get("spp", 4)),
                    cfg=_cfg(), label="random %d tris" % n)
    if name == "degenerate":    # identical centroids -> the n<=24 leaf branch of bvh.cpp:174-193
        prims = np.concatenate([sphere((0, -150, -40), 40.0, emissive=(10, 10, 10))])
        return dict(verts=generate(11, kw.get("n_same", 40), kw.get("n_other", 30)), primitives=prims,
                    materials=make_material((0.6, 0.4, 0.2)), camera=default_camera(),
                    width=int(kw.get("width", 96)), height=int(kw.get("height", 96)), spp=2, cfg=_cfg(),
                    label="degenerate centroids")
    raise ValueError(name)
