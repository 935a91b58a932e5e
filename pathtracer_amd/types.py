"""numpy views of the wire-format PODs of include/hpt/types.h
(reference pathtracer/common.hpp:116-186)."""
import ctypes as C

import numpy as np

TRIANGLE = np.dtype([("v1", "<f4", 4), ("e1", "<f4", 4), ("e2", "<f4", 4)])
BBOX = np.dtype([("min", "<f4", 3), ("max", "<f4", 3)])
INFO = np.dtype([("offset", "<i4"), ("secondChildOffset", "<i4"), ("nPrimitives", "<i4"), ("axis", "<i4")])
MATERIAL = np.dtype([("refractive", "<f4", 4), ("specular", "<f4", 4), ("diffuse", "<f4", 4), ("emissive", "<f4", 4)])
PRIMITIVE = np.dtype([("refractive", "<f4", 4), ("specular", "<f4", 4), ("diffuse", "<f4", 4), ("emissive", "<f4", 4),
                      ("center", "<f4", 4), ("type", "<f4", 4), ("plane_normal", "<f4", 4),
                      ("radius", "<f4"), ("pad", "<f4", 3)])
CAMERA = np.dtype([("eye", "<f4", 4), ("n", "<f4", 4), ("u", "<f4", 4), ("v", "<f4", 4)])
assert (TRIANGLE.itemsize, BBOX.itemsize, INFO.itemsize, MATERIAL.itemsize, PRIMITIVE.itemsize, CAMERA.itemsize) == \
    (48, 24, 16, 64, 128, 64)


class Config(C.Structure):
    """hpt_config"""
    _fields_ = [("max_depth", C.c_int32), ("rr_start", C.c_int32), ("diffuse_scale", C.c_float), ("seed", C.c_uint32)]


class Stats(C.Structure):
    """hpt_stats"""
    _fields_ = [("rays_primary", C.c_uint64), ("rays_bounce", C.c_uint64), ("paths", C.c_uint64),
                ("last_kernel_ms", C.c_double), ("wide_nodes", C.c_int32), ("tree_depth", C.c_int32),
                ("grid_waves", C.c_int32), ("launches", C.c_int32),
                ("fold_ms", C.c_double), ("trace_ms", C.c_double), ("shade_ms", C.c_double), ("reconstruct_ms", C.c_double),
                ("lane_loads", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


HIT_MISS = -1
