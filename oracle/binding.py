"""ctypes binding of oracle/libpt_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; nothing under pathtracer_amd/ does.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# numpy views of the wire-format PODs of include/hpt/types.h
TRIANGLE = np.dtype([("v1", "<f4", 4), ("e1", "<f4", 4), ("e2", "<f4", 4)])
BBOX = np.dtype([("min", "<f4", 3), ("max", "<f4", 3)])
INFO = np.dtype([("offset", "<i4"), ("secondChildOffset", "<i4"), ("nPrimitives", "<i4"), ("axis", "<i4")])
MATERIAL = np.dtype([("refractive", "<f4", 4), ("specular", "<f4", 4), ("diffuse", "<f4", 4), ("emissive", "<f4", 4)])
PRIMITIVE = np.dtype([("refractive", "<f4", 4), ("specular", "<f4", 4), ("diffuse", "<f4", 4), ("emissive", "<f4", 4),
                      ("center", "<f4", 4), ("type", "<f4", 4), ("plane_normal", "<f4", 4),
                      ("radius", "<f4"), ("pad", "<f4", 3)])
CAMERA = np.dtype([("eye", "<f4", 4), ("n", "<f4", 4), ("u", "<f4", 4), ("v", "<f4", 4)])
assert TRIANGLE.itemsize == 48 and BBOX.itemsize == 24 and INFO.itemsize == 16
assert MATERIAL.itemsize == 64 and PRIMITIVE.itemsize == 128 and CAMERA.itemsize == 64


class Config(C.Structure):
    _fields_ = [("max_depth", C.c_int32), ("rr_start", C.c_int32), ("diffuse_scale", C.c_float), ("seed", C.c_uint32)]


class Scene(C.Structure):
    _fields_ = [("triangles", C.c_void_p), ("n_triangles", C.c_int32),
                ("bbox", C.c_void_p), ("info", C.c_void_p), ("n_nodes", C.c_int32),
                ("primitives", C.c_void_p), ("n_primitives", C.c_int32),
                ("materials", C.c_void_p), ("n_materials", C.c_int32),
                ("_pad", C.c_int32 * 3),     # hpt_camera is 16-B aligned in C
                ("camera", C.c_float * 16),
                ("width", C.c_int32), ("height", C.c_int32),
                ("cfg", Config)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays_primary", "rays_bounce", "paths", "nodes_primary", "nodes_bounce",
                                          "tris_primary", "tris_bounce")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libpt_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libpt_oracle.so missing: run `make oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        lay = (C.c_int * 4)()
        L.orc_scene_layout(lay)
        assert (lay[1], lay[2], lay[3]) == (Scene.camera.offset, Scene.width.offset, Scene.cfg.offset), \
            "orc_scene layout mismatch between pt_oracle.h and binding.py"
        L.orc_build_bvh.restype = C.c_int
        L.orc_load_obj.restype = C.c_int64
        L.orc_load_obj.argtypes = [C.c_char_p, C.c_void_p, C.c_int64]
        L.orc_rand.restype = C.c_float
        L.orc_rand.argtypes = [C.c_uint32] * 4
        L.orc_intersect_box.argtypes = [C.c_void_p] * 4 + [C.c_float]
        L.orc_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.orc_tonemap.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_sincos.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleScene:
    """Holds numpy arrays alive and the orc_scene struct that points at them."""

    def __init__(self, triangles, bbox, info, primitives, materials, camera, width, height, cfg):
        self.triangles = np.ascontiguousarray(triangles, dtype=TRIANGLE)
        self.bbox = np.ascontiguousarray(bbox, dtype=BBOX)
        self.info = np.ascontiguousarray(info, dtype=INFO)
        self.primitives = np.ascontiguousarray(primitives, dtype=PRIMITIVE)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL)
        self.camera = np.ascontiguousarray(camera, dtype=CAMERA).reshape(())
        self.width, self.height = int(width), int(height)
        s = Scene()
        s.triangles = self.triangles.ctypes.data; s.n_triangles = len(self.triangles)
        s.bbox = self.bbox.ctypes.data; s.info = self.info.ctypes.data; s.n_nodes = len(self.bbox)
        s.primitives = self.primitives.ctypes.data; s.n_primitives = len(self.primitives)
        s.materials = self.materials.ctypes.data; s.n_materials = len(self.materials)
        cam = np.frombuffer(self.camera.tobytes(), dtype="<f4")
        for i in range(16):
            s.camera[i] = float(cam[i])
        s.width, s.height = self.width, self.height
        s.cfg = Config(int(cfg["max_depth"]), int(cfg["rr_start"]), float(cfg["diffuse_scale"]), int(cfg["seed"]))
        self.c = s

    @property
    def npx(self):
        return self.width * self.height

    def primary_hits(self, iteration, px_begin=0, px_end=None):
        px_end = self.npx if px_end is None else px_end
        n = px_end - px_begin
        hit = np.empty(n, np.int32); t = np.empty(n, np.float32)
        cnt = Counters()
        lib().orc_primary_hits(C.byref(self.c), C.c_int(iteration), C.c_int(px_begin), C.c_int(px_end), _p(hit), _p(t), C.byref(cnt))
        return hit, t, cnt.as_dict()

    def kernel_execute(self, iteration, n_threads=0, px_begin=0, px_end=None):
        px_end = self.npx if px_end is None else px_end
        rgb = np.zeros((self.npx, 3), np.float32)
        cnt = Counters()
        lib().orc_kernel_execute(C.byref(self.c), C.c_int(iteration), C.c_int(px_begin), C.c_int(px_end), _p(rgb), C.byref(cnt), C.c_int(n_threads))
        return rgb, cnt.as_dict()

    def render(self, spp_begin, spp_end, image=None, n_threads=0, px_begin=0, px_end=None):
        px_end = self.npx if px_end is None else px_end
        if image is None:
            image = np.zeros((self.npx, 3), np.float32)
        cnt = Counters()
        lib().orc_render(C.byref(self.c), C.c_int(spp_begin), C.c_int(spp_end), C.c_int(px_begin), C.c_int(px_end), _p(image), C.byref(cnt), C.c_int(n_threads))
        return image, cnt.as_dict()

    # ---- iterative pipeline (orc_iter_*): state is a numpy structured array mirroring orc_iter_pixel
    ITER_PIXEL = np.dtype([("o", "<f4", 3), ("d", "<f4", 3), ("weight", "<f4", 3), ("color", "<f4", 3),
                           ("ctr", "<u4"), ("bounce", "<i4"),
                           ("c_o", "<f4", 3), ("c_d", "<f4", 3), ("c_weight", "<f4", 3), ("c_color", "<f4", 3),
                           ("c_ctr", "<u4"), ("c_finished", "<i4"), ("completed", "<i4"), ("fresh", "<i4"),
                           ("contribution", "<f4", 3)])

    class _IterState(C.Structure):
        _fields_ = [("px", C.c_void_p), ("rays_primary", C.c_uint64), ("rays_bounce", C.c_uint64)]

    def iter_first_bounce(self):
        assert self.ITER_PIXEL.itemsize == 132
        self.iter_px = np.zeros(self.npx, self.ITER_PIXEL)
        self.iter_st = self._IterState(self.iter_px.ctypes.data, 0, 0)
        lib().orc_iter_first_bounce(C.byref(self.c), C.byref(self.iter_st))

    def iter_bounce(self):
        lib().orc_iter_bounce(C.byref(self.c), C.byref(self.iter_st))

    def iter_reconstruct(self, image):
        lib().orc_iter_reconstruct(C.byref(self.c), C.byref(self.iter_st), _p(image))

    def camera_ray(self, gid, iteration):
        o = np.empty(3, np.float32); d = np.empty(3, np.float32)
        lib().orc_camera_ray(C.byref(self.c), C.c_int(gid), C.c_int(iteration), _p(o), _p(d))
        return o, d


def build_bvh(verts):
    """verts: (n,3,3) float32 -> (bbox, info, order) by the bvh.cpp restatement."""
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 9)
    n = len(verts)
    bbox = np.zeros(2 * n, BBOX); info = np.zeros(2 * n, INFO); order = np.zeros(n, np.int32)
    nn = lib().orc_build_bvh(_p(verts), C.c_int(n), _p(bbox), _p(info), _p(order))
    if nn < 0:
        raise ValueError("orc_build_bvh: needs >= 12 triangles (reference reads vtriangles[0..11])")
    return bbox[:nn].copy(), info[:nn].copy(), order


def make_triangles(verts, order):
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 9)
    out = np.zeros(len(verts), TRIANGLE)
    lib().orc_make_triangles(_p(verts), _p(np.ascontiguousarray(order, np.int32)), C.c_int(len(verts)), _p(out))
    return out


def load_obj(path):
    n = lib().orc_load_obj(path.encode(), None, 0)
    if n < 0:
        return n, None
    out = np.zeros((n, 3), np.float32)
    lib().orc_load_obj(path.encode(), _p(out), n)
    return n, out


def accumulate(image, sample, iterations):
    lib().orc_accumulate(_p(image), _p(sample), image.size, iterations)
    return image


def tonemap(image):
    out = np.empty(image.size, np.uint8)
    lib().orc_tonemap(_p(np.ascontiguousarray(image, np.float32)), _p(out), image.size)
    return out.reshape(image.shape)
