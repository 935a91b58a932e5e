"""Development tool: what the analytic primitive list costs at C5 -- the glossy scene rendered (8 spp) with its 129 primitives, with 13 (still the
two-phase kernel), 11 (one-pass loop) and 1.  Different primitives give different paths; the figures are per-ray rates.   python bench/c5_split.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pathtracer_amd import api, scenes
scn = scenes.scene("glossy")
bvh = api.build_bvh(scn["verts"])
for label, prims in (("129 primitives", scn["primitives"]), ("13 primitives (short-list kernel boundary)", scn["primitives"][:13]), ("11 primitives (one-pass loop)", scn["primitives"][:11]), ("1 primitive", scn["primitives"][:1])):
    be = api.Backend(scn["width"], scn["height"])
    be.load(dict(scn, primitives=prims), bvh)
    img = torch.zeros(scn["width"] * scn["height"] * 3, dtype=torch.float32, device="cuda")
    for i in range(2):
        be.render(0, 8, img.data_ptr()); st = be.stats()
    rays = st["rays_primary"] + st["rays_bounce"]
    print("%-45s %8.2f ms  %6.1f Mrays/s  %.2f rays/path" % (label, st["last_kernel_ms"], rays / st["last_kernel_ms"] / 1e3, rays / st["paths"]))
    be.close()
