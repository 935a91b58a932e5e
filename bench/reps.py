"""Development tool: kernel time of each of N successive hpt_render calls of one workload (HPT_LIB=...).
    python bench/reps.py [scene] [spp] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pathtracer_amd import api, scenes
scn = scenes.scene(sys.argv[1] if len(sys.argv) > 1 else "sponza")
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
bvh = api.build_bvh(scn["verts"])
be = api.Backend(scn["width"], scn["height"])
be.load(scn, bvh)
img = torch.zeros(scn["width"] * scn["height"] * 3, dtype=torch.float32, device="cuda")
ms = []
for i in range(reps):
    be.render(0, spp, img.data_ptr())
    ms.append(round(be.stats()["last_kernel_ms"], 2))
print(os.environ.get("HPT_LIB", "default"), ms, flush=True)
