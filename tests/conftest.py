import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def bits(a):
    """float32 array -> uint32 bit patterns with every NaN mapped to one value
    (x86 and gfx950 differ in NaN sign/payload, nothing else)."""
    a = np.ascontiguousarray(a, np.float32)
    b = a.view(np.uint32).copy()
    b[np.isnan(a)] = 0x7FC00000
    return b


_SCENE_CACHE = {}


def get_scene(name, **kw):
    """scene dict + host-built BVH + OracleScene, cached per (name, kwargs)."""
    from oracle import binding as orc
    from pathtracer_amd import api, scenes
    key = (name, tuple(sorted(kw.items())))
    if key not in _SCENE_CACHE:
        s = scenes.scene(name, **{k: v for k, v in kw.items() if k not in ("width", "height") or name in ("random", "degenerate")})
        if "width" in kw:
            s["width"], s["height"] = kw["width"], kw["height"]
        bvh = api.build_bvh(s["verts"])
        o = orc.OracleScene(bvh["triangles"], bvh["bbox"], bvh["info"], s["primitives"], s["materials"], s["camera"],
                            s["width"], s["height"], s["cfg"])
        _SCENE_CACHE[key] = (s, bvh, o)
    return _SCENE_CACHE[key]
