"""Development tool: time from host arrays to the first finished launch (upload, node/record layout, first render).
    python bench/load_time.py [scene]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pathtracer_amd import api, scenes
scn = scenes.scene(sys.argv[1] if len(sys.argv) > 1 else "tenm")
t0 = time.perf_counter(); bvh = api.build_bvh(scn["verts"]); t_build = time.perf_counter() - t0
be = api.Backend(scn["width"], scn["height"])
img = torch.zeros(scn["width"] * scn["height"] * 3, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter(); be.upload_triangles(bvh["triangles"]); t_tris = time.perf_counter() - t0
t0 = time.perf_counter(); be.upload_bvh(bvh["bbox"], bvh["info"]); t_bvh = time.perf_counter() - t0
be.set_scene(scn["primitives"], scn["materials"]); be.set_camera(scn["camera"]); be.set_config(scn["cfg"])
t0 = time.perf_counter(); be.render(0, 1, img.data_ptr()); be.synchronize(); t_first = time.perf_counter() - t0
t0 = time.perf_counter(); be.render(0, 1, img.data_ptr()); be.synchronize(); t_second = time.perf_counter() - t0
print("%s: %d triangles, %d nodes: build %.3f s, upload_triangles %.3f s, upload_bvh %.3f s, first 1-spp render %.3f s (second %.3f s)"
      % (sys.argv[1] if len(sys.argv) > 1 else "tenm", len(bvh["triangles"]), len(bvh["bbox"]), t_build, t_tris, t_bvh, t_first, t_second))
