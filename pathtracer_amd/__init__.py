"""pathtracer_amd — MI355X (gfx950) HIP render path for brianrli/pathtracer behind a C-ABI.

csrc/            HIP kernels + C-ABI (include/hpt/hpt.h) + C++ host side (facade, BVH, OBJ, image out)
api.py           ctypes binding (Backend = the C-ABI, Pathtracer = the C++ facade)
scenes.py        procedural stand-in scenes for the BASELINE.json configs
"""
from . import api, scenes, types  # noqa: F401
from .api import Backend, HptError, Pathtracer, build_bvh, load_obj, save_png  # noqa: F401
