"""Development tool: runs one launch of a -DHPT_PROFILE_STEPS build (HPT_LIB=...) and lets the library print
its per-step-kind trip and lane-activity counters.   python bench/step_profile.py [scene] [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pathtracer_amd import api, scenes
scn = scenes.scene(sys.argv[1] if len(sys.argv) > 1 else "sponza")
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bvh = api.build_bvh(scn["verts"])
be = api.Backend(scn["width"], scn["height"])
be.load(scn, bvh)
img = torch.zeros(scn["width"] * scn["height"] * 3, dtype=torch.float32, device="cuda")
be.render(0, spp, img.data_ptr())
st = be.stats()
print(st)
