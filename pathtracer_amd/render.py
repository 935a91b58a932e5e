"""Headless render on one or several GPUs: the multi-GPU counterpart of hpt_cli (host/hpt_cli.cpp), which takes
the place of the reference's GLUT viewer (main.cpp:92-151; its 'p' key, main.cpp:194-203, becomes --out).

    python -m pathtracer_amd.render --obj model.obj --w 512 --h 512 --spp 64 --out out.png
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m pathtracer_amd.render --obj ... --out out.png

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the launcher); scene and BVH are built by every rank
(replicated); --shard tiles (default: 8x8 tiles round-robin, one all-gather, image bit-identical
for any number of GPUs) or samples (iterations s % N == rank, one all-reduce of weighted running means, within
1e-5 relative).  Rank 0 writes the image.  Scene set-up follows hpt_cli: reference default scene and camera, or
one emissive sphere with --light; --scene NAME renders one of the bench scenes (pathtracer_amd/scenes.py).
"""
import argparse
import json
import os
import sys
import time

import numpy as np


def default_scene():
    """Pathtracer::set_scene() (reference pathtracer.cpp:322-382): materials[0] diffuse (0.6,0.4,0.2), five
    primitives of which the two planes are typed 4 (dead) and sphere 1 is the light."""
    from . import scenes
    prims = np.concatenate([
        scenes.plane((0, 8, 0), (0, -1, 0), (0.6, 0.6, 0.6), ptype=4),
        scenes.sphere((-0.5, 0.0, 0), 0.4, emissive=(10, 10, 10)),
        scenes.sphere((-0.5, 0.6, 0), 0.4, diffuse=(0.7, 0.7, 0.2), specular=(0.7, 0.7, 0.2), ior=1.9),
        scenes.sphere((0.5, 0.6, 0.0), 0.4, diffuse=(0.4, 0.4, 0.4)),
        scenes.plane((0, 0, 1.5), (0, 0, 1), (0.8, 0.8, 0.95), ptype=4),
    ])
    return prims, scenes.make_material((0.6, 0.4, 0.2))


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m pathtracer_amd.render")
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--obj", help="Wavefront OBJ file (what the reference's load_obj reads, plus a/b/c, negative indices, polygons)")
    src.add_argument("--scene", help="a bench scene: cornell | bunny | sponza | glossy | tenm")
    ap.add_argument("--w", type=int, default=0)
    ap.add_argument("--h", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--depth", type=int, default=0, help="max path depth (reference MAXDEPTH 15)")
    ap.add_argument("--light", type=float, nargs=5, metavar=("CX", "CY", "CZ", "RADIUS", "INTENSITY"),
                    help="replace the default primitives by one emissive sphere (triangles cannot emit)")
    ap.add_argument("--shard", default="tiles", choices=["tiles", "samples"])
    ap.add_argument("--out", default="out.png", help=".png or .ppm")
    a = ap.parse_args(argv)

    import torch
    import torch.distributed as dist
    from . import api, scenes
    from . import dist as pdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("pathtracer_amd.render needs a GPU: the render path has no CPU fallback")
    # HPT_DIST_BACKEND=gloo rehearses N>1 on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("HPT_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if a.scene:
        scn = scenes.scene(a.scene)
    else:
        prims, mats = default_scene()
        scn = dict(verts=api.load_obj(a.obj).reshape(-1, 3, 3), primitives=prims, materials=mats, camera=scenes.default_camera(),
                   width=100, height=100, spp=16, cfg=dict(scenes.DEFAULT_CFG), label=a.obj)
    if a.light:
        cx, cy, cz, radius, e = a.light
        scn["primitives"] = scenes.sphere((cx, cy, cz), radius, emissive=(e, e, e))
    w, h, spp = a.w or scn["width"], a.h or scn["height"], a.spp or scn["spp"]
    scn = dict(scn, width=w, height=h, cfg=dict(scn["cfg"], max_depth=a.depth or scn["cfg"]["max_depth"]))

    t0 = time.perf_counter()
    bvh = api.build_bvh(scn["verts"])
    t_bvh = time.perf_counter() - t0
    be = api.Backend(w, h, device=local_rank)
    be.load(scn, bvh)
    be.set_stream(torch.cuda.current_stream().cuda_stream)
    if a.shard == "tiles":
        be.set_shard(rank, world)
    else:
        be.set_sample_shard(rank, world)
    image = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
    t0 = time.perf_counter()
    be.render(0, spp, image.data_ptr())
    if world > 1:
        exchange = pdist.Exchange(be, image, world)
        if a.shard == "tiles":
            exchange.gather()
        else:
            exchange.reduce(pdist.sample_shard_count(0, spp, rank, world), spp)
    be.synchronize()
    torch.cuda.synchronize()
    t_render = time.perf_counter() - t0
    st = be.stats()
    rays = torch.tensor([float(st["rays_primary"] + st["rays_bounce"])], dtype=torch.float64,
                        device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(rays)
    if rank == 0:
        rgb = image.cpu().numpy().reshape(h, w, 3)
        if a.out.endswith(".ppm"):
            if api.lib().hpt_host_save_ppm(api._p(np.ascontiguousarray(rgb)), w, h, os.fsencode(a.out)) != 0:
                raise SystemExit("cannot write %s" % a.out)
        else:
            api.save_png(rgb, a.out)
        print(json.dumps({"triangles": int(len(bvh["triangles"])), "width": w, "height": h, "spp": spp, "gpus": world,
                          "shard": a.shard, "bvh_s": round(t_bvh, 4), "render_s": round(t_render, 4),
                          "rays": float(rays[0]), "out": a.out}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
