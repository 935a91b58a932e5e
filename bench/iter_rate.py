"""Development tool: throughput of the iterative pipeline (hpt_first_bounce / hpt_bounce / hpt_reconstruct_paths)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pathtracer_amd import api, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "sponza"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
scn = scenes.scene(name)
bvh = api.build_bvh(scn["verts"])
be = api.Backend(scn["width"], scn["height"]); be.load(scn, bvh)
img = torch.zeros(scn["width"] * scn["height"] * 3, dtype=torch.float32, device="cuda")
for rep in range(2):
    be.first_bounce(); be.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        be.bounce(); be.reconstruct_paths(img.data_ptr())
    be.synchronize()
    dt = time.perf_counter() - t0
    st = be.stats()
    print("%s: %d bounce calls %.1f ms each, %.0f Mrays/s (bounce rays %d)" % (name, n, dt / n * 1e3, st["rays_bounce"] / dt / 1e6, st["rays_bounce"]), flush=True)
