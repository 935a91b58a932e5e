"""Headless render on one or several GPUs: the multi-GPU counterpart of hpt_cli (host/hpt_cli.cpp), which takes
the place of the reference's GLUT viewer (main.cpp:92-151; its 'p' key, main.cpp:194-203, becomes --out).

    python -m pathtracer_amd.render --obj model.obj --w 512 --h 512 --spp 64 --out out.png
    python -m pathtracer_amd.render --gpus 8 --obj ... --out out.png          (starts the 8 ranks itself)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m pathtracer_amd.render --gpus 8 --obj ... --out out.png

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the launcher); scene and BVH are built by every rank
(replicated); --shard tiles (default: 8x8 tiles round-robin, one all-gather, image bit-identical
for any number of GPUs) or samples (iterations s % N == rank, one all-reduce of weighted running means, within
1e-5 relative).  Rank 0 writes the image.  --reduce-every K renders progressively: K-iteration slices with the exchange after each
(every rank then holds the frame so far; --preview writes it after every slice).  --checkpoint FILE saves the
accumulation buffer and its sample count after every slice (per rank: FILE.rankR for N > 1), --resume continues from it:
in tile mode bit-identically to a render that was never stopped.  Scene set-up follows hpt_cli: reference default scene and camera, or
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
    ap.add_argument("--gpus", type=int, default=0, help="number of GPUs (ranks); without a launcher the ranks are started here")
    ap.add_argument("--reduce-every", type=int, default=0, metavar="K", help="progressive: exchange after every K iterations")
    ap.add_argument("--exchange", default="torch", choices=["torch", "native"],
                    help="torch.distributed collectives, or the library's own communicator (RCCL called from C++)")
    ap.add_argument("--checkpoint", help="file for the accumulation buffer + sample count, written after every slice")
    ap.add_argument("--resume", action="store_true", help="continue from --checkpoint if it exists")
    ap.add_argument("--preview", action="store_true", help="with --reduce-every: write --out after every slice")
    ap.add_argument("--stop-after", type=int, default=0, metavar="S", help="stop once S iterations are done (testing --resume)")
    a = ap.parse_args(argv)

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # start the ranks before anything here touches the GPU; rank 0 prints the line, the exit status is the launcher's
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        args = list(sys.argv[1:] if argv is None else argv)
        return subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
                                "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "pathtracer_amd.render"] + args)

    import torch
    import torch.distributed as dist
    from . import api, scenes
    from . import dist as pdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus and a.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
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
    # sample shards rendered progressively: the rank's own running mean must survive each reduction
    frame = torch.zeros_like(image) if (a.shard == "samples" and world > 1 and a.reduce_every) else image
    exchange = None
    if world > 1:
        exchange = pdist.NativeExchange(be, image, world, rank) if a.exchange == "native" else pdist.Exchange(be, image, world)
    ckpt = None
    if a.checkpoint:
        ckpt = a.checkpoint if world == 1 else "%s.rank%d" % (a.checkpoint, rank)
    done = 0
    if a.resume and ckpt and os.path.exists(ckpt):
        done = be.checkpoint_load(image.data_ptr(), ckpt)      # refuses another size, config or shard
    if a.resume and ckpt and world > 1:
        # the slices are collective: every rank must continue at the same iteration
        lohi = torch.tensor([done, -done], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(lohi, op=dist.ReduceOp.MAX)
        if int(lohi[0]) != -int(lohi[1]):
            raise SystemExit("the ranks' checkpoints disagree (%d..%d iterations done): remove them or restore a matching set"
                             % (-int(lohi[1]), int(lohi[0])))

    def write(path):
        rgb = frame.cpu().numpy().reshape(h, w, 3)
        if path.endswith(".ppm"):
            if api.lib().hpt_host_save_ppm(api._p(np.ascontiguousarray(rgb)), w, h, os.fsencode(path)) != 0:
                raise SystemExit("cannot write %s" % path)
        else:
            api.save_png(rgb, path)

    t0 = time.perf_counter()
    K = a.reduce_every if a.reduce_every > 0 else spp
    target = min(spp, a.stop_after) if a.stop_after else spp
    n_rays = 0
    while done < target:
        nxt = min(done + K, target)
        be.render(done, nxt, image.data_ptr())
        if ckpt:
            # this rank's own accumulation (before the exchange overwrites the other ranks' tiles: they save theirs)
            be.checkpoint_save(image.data_ptr(), nxt, ckpt)
        if exchange is not None:
            if a.shard == "tiles":
                exchange.gather()
            else:
                exchange.reduce(pdist.sample_shard_count(0, nxt, rank, world), nxt, None if frame is image else frame)
        st = be.stats()                                       # (waits for the slice)
        n_rays += st["rays_primary"] + st["rays_bounce"]
        done = nxt
        if a.preview and a.reduce_every and rank == 0:
            write(a.out)
    if exchange is not None and n_rays == 0 and done > 0:      # resumed from a finished checkpoint: it holds this rank's part only
        if a.shard == "tiles":
            exchange.gather()
        else:
            exchange.reduce(pdist.sample_shard_count(0, done, rank, world), done, None if frame is image else frame)
    be.synchronize()
    torch.cuda.synchronize()
    t_render = time.perf_counter() - t0
    rays = torch.tensor([float(n_rays)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(rays)
    if rank == 0:
        write(a.out)
        print(json.dumps({"triangles": int(len(bvh["triangles"])), "width": w, "height": h, "spp": done, "gpus": world,
                          "shard": a.shard, "bvh_s": round(t_bvh, 4), "render_s": round(t_render, 4),
                          "rays": float(rays[0]), "out": a.out}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
