#!/usr/bin/env python3
"""bench.py — headline benchmark of the HIP render path.

    python bench.py --gpus N --steps K --warmup W

One STEP = one whole frame of the BASELINE.json headline config C3 (Sponza-class stand-in scene,
262 267 triangles, 1920x1080, 256 spp, max depth 8) through the hot path: camera rays, BVH
traversal + triangle tests, bounce/shading/Russian roulette, running-mean accumulation, and for
N > 1 the all-gather of the tile shards (RCCL).  Scene, BVH and image stay resident in HBM; scene
generation, BVH build and upload are outside the timed region.  N > 1: one process per GPU (torchrun
env), the frame's 8x8 tiles are sharded round-robin over the ranks (strong scaling: the frame is fixed).

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` and, at N=1,
`cpu_baseline` (the CPU oracle timed on this box's host cores on a bounded sample of the same frame).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
ENGINE_CLOCK_HZ = 2.4e9         # MI355X peak engine clock (same guide)
# vector-L1 ceiling for 16-B-per-lane gathers, measured with bench/micro/gather_exec.hip (node-shaped fetches
# from an L2-resident table: 1.35; L1-resident: 1.45-1.58 lane-loads per clock per CU)
L1_GATHER_CEILING = 1.45


def algorithmic_bytes(costs, rays_primary, rays_bounce, n_pixels, spp):
    """bytes(ray) = 40*N_nodes + 48*N_tris + 144, N_* = reference-order counts measured by the
    CPU oracle (bench/frozen_costs.json); + 28*W*H per accumulated sample."""
    bp = 40.0 * costs["nodes_per_primary_ray"] + 48.0 * costs["tris_per_primary_ray"] + 144.0
    bb = 40.0 * costs["nodes_per_bounce_ray"] + 48.0 * costs["tris_per_bounce_ray"] + 144.0
    return rays_primary * bp + rays_bounce * bb + 28.0 * n_pixels * spp


def cpu_baseline(scn, bvh, seconds_target=15.0):
    """Times the CPU oracle (kind "port": scalar C restatement, OpenMP over pixels) on a bounded sample:
    the same frame at a reduced sample count, grown until ~seconds_target of CPU work."""
    from oracle import binding as orc
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    o = orc.OracleScene(bvh["triangles"], bvh["bbox"], bvh["info"], scn["primitives"], scn["materials"], scn["camera"],
                        scn["width"], scn["height"], scn["cfg"])
    t0 = time.perf_counter()
    _, c = o.render(0, 1, n_threads=threads)
    dt = time.perf_counter() - t0
    rays = c["rays_primary"] + c["rays_bounce"]
    spp = 1
    extra = int(max(0, min(16, (seconds_target - dt) / max(dt, 1e-3))))
    if extra >= 1:
        t0 = time.perf_counter()
        _, c = o.render(1, 1 + extra, n_threads=threads)
        dt = time.perf_counter() - t0
        rays = c["rays_primary"] + c["rays_bounce"]
        spp = extra
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "%s, full %dx%d frame at %d spp (%d rays, %.1f s)" % (scn["label"].split(" ")[0], scn["width"], scn["height"], spp, rays, dt),
            "nodes_per_ray": (c["nodes_primary"] + c["nodes_bounce"]) / rays,
            "tris_per_ray": (c["tris_primary"] + c["tris_bounce"]) / rays}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="sponza", help="sponza (C3, headline) | bunny (C2) | cornell (C1) | glossy (C5) | tenm (C4)")
    ap.add_argument("--spp", type=int, default=0, help="override the config's spp (debugging only: makes the number non-headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard", default="tiles", choices=["tiles", "samples"],
                    help="N>1: tiles = 8x8 tiles round-robin + one all-gather (bit-identical image, the default); "
                         "samples = iterations s %% N == rank + one all-reduce (SURVEY.md 8e secondary mode)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pathtracer_amd import api, scenes
    from pathtracer_amd import dist as pdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    # HPT_DIST_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks (ranks then share devices)
    dist_backend = os.environ.get("HPT_DIST_BACKEND", "nccl")
    if dist_backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(dist_backend, rank=rank, world_size=world)

    scn = scenes.scene(args.scene)
    spp = args.spp or scn["spp"]
    w, h = scn["width"], scn["height"]
    t_build = time.perf_counter()
    bvh = api.build_bvh(scn["verts"])
    t_build = time.perf_counter() - t_build
    be = api.Backend(w, h, device=local_rank)
    be.load(scn, bvh)
    if args.shard == "tiles":
        be.set_shard(rank, world)
    else:
        be.set_sample_shard(rank, world)
    be.set_stream(torch.cuda.current_stream().cuda_stream)
    image = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")

    def step():
        be.render(0, spp, image.data_ptr())
        if world > 1 and args.shard == "tiles":
            pdist.gather_image(be, image, world)
        elif world > 1:
            be.synchronize()
            if dist_backend == "nccl":
                pdist.reduce_samples(image, pdist.sample_shard_count(0, spp, rank, world), spp)
            else:                                         # gloo rehearsal: stage through the host
                host = image.cpu()
                pdist.reduce_samples(host, pdist.sample_shard_count(0, spp, rank, world), spp)
                image.copy_(host)

    def fence():
        be.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    kernel_ms = []
    rays_p = rays_b = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # hpt_get_stats waits for the launch stream (the step is already synchronous for N>1) and reads the
        # HIP-event time of the render kernel recorded on that stream
        st = be.stats()
        kernel_ms.append(st["last_kernel_ms"])
        rays_p, rays_b = st["rays_primary"], st["rays_bounce"]
    fence()
    elapsed = time.perf_counter() - t0

    tot = torch.tensor([elapsed, float(rays_p), float(rays_b), sum(kernel_ms) / len(kernel_ms)], dtype=torch.float64,
                       device="cuda" if dist_backend == "nccl" else "cpu")
    if world > 1:
        mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, kern_ms = float(mx[0]), float(mx[3])
        rays_p, rays_b = float(sm[1]), float(sm[2])
    else:
        kern_ms = float(tot[3])
    rays_frame = rays_p + rays_b

    if rank == 0:
        costs_all = json.load(open(os.path.join(ROOT, "bench", "frozen_costs.json")))
        costs = costs_all.get(args.scene)
        roofline = None
        if costs:
            # per launch = per rank per frame; with N ranks each launch handles 1/N of the frame's rays
            alg = algorithmic_bytes(costs, rays_p / world, rays_b / world, w * h / world, spp)
            achieved = alg / (kern_ms * 1e-3) / 1e9
            traffic = None
            gather = None
            scene_mb = (len(bvh["bbox"]) * 40 + len(bvh["triangles"]) * 48) / 1e6
            pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc_path) and world == 1 and not args.spp:
                pmc = json.load(open(pmc_path)).get(args.scene, {})
                traffic = pmc.get("hbm_bytes_per_launch")
                if pmc.get("tcp_lane_accesses_per_launch"):
                    # what actually bounds the kernel: lane-loads served by the vector L1 per clock per CU
                    n_cu = torch.cuda.get_device_properties(local_rank).multi_processor_count
                    rate = pmc["tcp_lane_accesses_per_launch"] / (kern_ms * 1e-3) / n_cu / ENGINE_CLOCK_HZ
                    gather = {"lane_loads_per_clk_per_cu": rate, "ceiling": L1_GATHER_CEILING, "frac": rate / L1_GATHER_CEILING,
                              "lane_loads_per_ray": pmc["tcp_lane_accesses_per_launch"] / rays_frame,
                              "source": "TCP_TOTAL_CACHE_ACCESSES of profiles/pmc_traffic.json over this run's kernel time"}
            roofline = {"bound": "hbm", "kernel": "render_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                        "algorithmic_bytes_per_launch": alg, "kernel_ms": kern_ms,
                        "bytes_per_primary_ray": 40.0 * costs["nodes_per_primary_ray"] + 48.0 * costs["tris_per_primary_ray"] + 144.0,
                        "bytes_per_bounce_ray": 40.0 * costs["nodes_per_bounce_ray"] + 48.0 * costs["tris_per_bounce_ray"] + 144.0,
                        "l1_gather": gather,
                        # SURVEY.md 8d asks for these beside the HBM fraction (same PMC run as `traffic`)
                        "pmc": ({k: pmc[k] for k in ("l2_hit_rate", "l1_miss_to_l2_frac", "valu_busy_frac_of_cu_time", "lane_utilisation") if k in pmc}
                                if traffic is not None else None),
                        "note": ("BVH+triangles (%.0f MB) are cache-resident, so algorithmic bytes/s can exceed the HBM peak; "
                                 "the measured limiter is the vector-L1 gather rate (l1_gather); see DESIGN.md §Roofline"
                                 if scene_mb <= 256 else
                                 "BVH+triangles (%.0f MB) exceed the 256 MB Infinity Cache: the top of the tree still hits in "
                                 "cache, the rest streams from HBM (PMC: profiles/r01_final/other_configs); see DESIGN.md §Roofline")
                                % scene_mb}
        out = {
            "metric": "Mrays/s (primary+bounce); ms_per_step = ms/frame %dx%d %dspp" % (w, h, spp),
            "value": rays_frame * args.steps / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": scn["label"] + ("" if not args.spp else " [spp overridden to %d]" % spp),
                       "width": w, "height": h, "spp": spp, "max_depth": scn["cfg"]["max_depth"],
                       "triangles": int(len(bvh["triangles"])), "bvh_nodes": int(len(bvh["bbox"])),
                       "rays_per_frame": rays_frame, "primary_rays": rays_p, "bounce_rays": rays_b,
                       "sharding": ("8x8 tiles round-robin over %d rank(s), all-gather of slabs" if args.shard == "tiles" else
                                    "iterations s %% %d == rank, all-reduce of weighted running means") % world,
                       "bvh_build_s": t_build},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scn, bvh)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
