"""Development tool: render_kernel time of each of N tile shards of one frame on ONE GPU (how even the round-robin
tile assignment is: the slowest shard sets an N-GPU frame's time).
    python bench/shard_balance.py [scene] [world] [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pathtracer_amd import api, scenes
scn = scenes.scene(sys.argv[1] if len(sys.argv) > 1 else "sponza")
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
spp = int(sys.argv[3]) if len(sys.argv) > 3 else scn["spp"]
bvh = api.build_bvh(scn["verts"])
be = api.Backend(scn["width"], scn["height"])
be.load(scn, bvh)
img = torch.zeros(scn["width"] * scn["height"] * 3, dtype=torch.float32, device="cuda")
be.render(0, spp, img.data_ptr()); whole = be.stats()
ms, rays = [], []
for r in range(world):
    be.set_shard(r, world)
    be.render(0, spp, img.data_ptr()); be.render(0, spp, img.data_ptr())
    st = be.stats()
    ms.append(st["last_kernel_ms"] + st["fold_ms"]); rays.append(st["rays_primary"] + st["rays_bounce"])
print("%s %d spp: whole frame %.2f ms (%d launches); %d shards: %s ms; max %.2f = %.3f of whole/%d; rays max/mean %.4f"
      % (scn["label"].split(" ")[0], spp, whole["last_kernel_ms"] + whole["fold_ms"], whole["launches"], world,
         " ".join("%.2f" % m for m in ms), max(ms), max(ms) / ((whole["last_kernel_ms"] + whole["fold_ms"]) / world), world,
         max(rays) / (sum(rays) / world)))
