"""Development tool: wall time of the host BVH build (api.build_bvh) for a scene at several thread counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import api, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "tenm"
s = scenes.scene(name)
for th in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,4,16").split(",")]:
    os.environ["HPT_BUILD_THREADS"] = str(th)
    t = time.perf_counter(); b = api.build_bvh(s["verts"]); dt = time.perf_counter() - t
    print("%s: %d triangles, %d nodes, %2d threads: %.3f s" % (name, len(b["triangles"]), len(b["bbox"]), th, dt), flush=True)
