"""Development tool: rate of the iterative pipeline (hpt_bounce + hpt_reconstruct_paths per call) on one workload,
with the per-stage device times of hpt_get_stats.
    python bench/iter_rate.py [scene] [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pathtracer_amd import api, scenes
scn = scenes.scene(sys.argv[1] if len(sys.argv) > 1 else "sponza")
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 64
bvh = api.build_bvh(scn["verts"])
be = api.Backend(scn["width"], scn["height"])
be.load(scn, bvh)
img = torch.zeros(scn["width"] * scn["height"] * 3, dtype=torch.float32, device="cuda")
be.first_bounce(); be.synchronize()
st = be.stats()
print("first_bounce: trace %.3f ms shade %.3f ms" % (st["trace_ms"], st["shade_ms"]))
r0 = st["rays_primary"] + st["rays_bounce"]
tr = sh = rc = 0.0
t0 = time.perf_counter()
for _ in range(calls):
    be.bounce(); be.reconstruct_paths(img.data_ptr())
    st = be.stats()                         # (synchronises: the times below are device times, the wall clock includes this)
    tr += st["trace_ms"]; sh += st["shade_ms"]; rc += st["reconstruct_ms"]
be.synchronize()
dt = time.perf_counter() - t0
rays = st["rays_primary"] + st["rays_bounce"] - r0
print("%s: %d calls, %.0f rays/call; per call: trace %.3f ms, shade %.3f ms, reconstruct %.3f ms; device %.0f Mrays/s, wall %.0f Mrays/s"
      % (scn["label"].split(" ")[0], calls, rays / calls, tr / calls, sh / calls, rc / calls, rays / (tr + sh + rc) / 1e3, rays / dt / 1e6))
