#!/usr/bin/env python3
"""bench.py — headline benchmark of the HIP render path.

    python bench.py --gpus N --steps K --warmup W

One STEP = one whole frame of the BASELINE.json headline config C3 (Sponza-class stand-in scene,
262 267 triangles, 1920x1080, 256 spp, max depth 8) through the hot path: camera rays, BVH
traversal + triangle tests, bounce/shading/Russian roulette, running-mean accumulation, and for
N > 1 the all-gather of the tile shards (RCCL).  Scene, BVH and image stay resident in HBM; scene
generation, BVH build and upload are outside the timed region.  N > 1: one process per GPU (torchrun
env), the frame's 8x8 tiles are sharded round-robin over the ranks (strong scaling: the frame is fixed).

`--gpus N` with N > 1 and no launcher environment (no WORLD_SIZE): this script starts the N ranks itself, as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`, before anything touches
the GPU, passes rank 0's line through and exits with the launcher's status.  Under a launcher, WORLD_SIZE must equal
--gpus: the line never reports an `n_gpus` other than the one asked for.

`--reduce-every K` renders the frame progressively (BASELINE config 5: "1024 spp progressive, reduce every 16 spp"):
K-iteration slices, the exchange after every slice; the line then also carries the single-exchange figure.
`--exchange native` runs the exchange through the library's own communicator (RCCL called from C++, hpt_gather_tiles /
hpt_reduce_samples) instead of torch.distributed's collectives; `--force-exchange` runs the exchange step at N = 1 too
(a one-rank RCCL communicator: what a one-GPU box can execute of that path).

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
VALU_PEAK_TFLOPS = 157.3        # fp32 vector peak (same guide)
# Ceiling of the vector L1 for 16-B-per-lane gathers, in lane-loads per clock per CU: the rate of
# bench/micro/gather_exec.hip for node-shaped fetches (the four quarters of one 64-B line per lane, what the NODE
# step issues) and for single 4-/8-byte gathers from an L1-resident table with all 64 lanes active (1.48 / 1.49: the
# best case of any gather; 1.36 when the table is L2-resident, 1.08-1.34 with 32 of 64 lanes active); its output and
# the rocprofv3 trace of that run are profiles/r02/gather_exec.txt and profiles/r02/gather_exec_kernel_stats.csv.
L1_GATHER_CEILING = 1.49
LANE_LOAD_BYTES = 16


def kernel_source_hash():
    """sha1 of the sources the render kernel is built from: PMC-derived figures in profiles/pmc_traffic.json carry
    the hash they were measured at and are reported only while it still matches (they go stale with the kernel)."""
    import hashlib
    import re
    h = hashlib.sha1()
    for f in ("pathtracer_amd/csrc/render_kernels.h", "pathtracer_amd/csrc/device_math.h", "pathtracer_amd/csrc/hpt_backend.hip"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    # of the Makefile only what reaches the kernel: the compiler flags and the two sizes they carry (round 3: a new test
    # target in the Makefile must not make the PMC passes look stale)
    mk = open(os.path.join(ROOT, "Makefile")).read()
    for pat in (r"^HIPFLAGS\s*=(?:[^\n]*\\\n)*[^\n]*$", r"^SAMPLE_BUDGET_MB\s*\?=[^\n]*$", r"^NOALLOC_SCRATCH_MB\s*\?=[^\n]*$"):
        m = re.search(pat, mk, re.M)
        h.update((m.group(0) if m else "").encode())
    return h.hexdigest()[:12]


def reference_costs(costs, rays_primary, rays_bounce, n_pixels, spp):
    """bytes(ray) = 40*N_nodes + 48*N_tris + 144, flops(ray) = 12*N_nodes + 45*N_tris, N_* =
    reference-order counts measured by the CPU oracle (bench/frozen_costs.json); + 28*W*H bytes per accumulated sample."""
    bp = 40.0 * costs["nodes_per_primary_ray"] + 48.0 * costs["tris_per_primary_ray"] + 144.0
    bb = 40.0 * costs["nodes_per_bounce_ray"] + 48.0 * costs["tris_per_bounce_ray"] + 144.0
    fp = 12.0 * costs["nodes_per_primary_ray"] + 45.0 * costs["tris_per_primary_ray"]
    fb = 12.0 * costs["nodes_per_bounce_ray"] + 45.0 * costs["tris_per_bounce_ray"]
    return (rays_primary * bp + rays_bounce * bb + 28.0 * n_pixels * spp, rays_primary * fp + rays_bounce * fb, bp, bb)


def host_cpu():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count() or 1


def cpu_baseline(scn, bvh, seconds_target=14.0):
    """Times the CPU oracle (kind "port": scalar C restatement of the reference kernels, OpenMP over 16x16 pixel
    tiles) on a bounded sample of the same frame: the whole frame at a reduced sample count on every core this job
    may use (OMP_NUM_THREADS -- the GPU box gives a 1-GPU job 16 of the host's cores -- else all of them), grown
    until ~seconds_target of work, and one thread on a band of rows for the single-thread figure."""
    from oracle import binding as orc
    model, n_logical = host_cpu()
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or n_logical
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
    # one thread: a band of rows in the middle of the frame, sized from the multi-thread rate to ~4 s
    w, h = scn["width"], scn["height"]
    rows = int(max(1, min(h, 4.0 * (rays / dt / threads) / (rays / spp / h))))
    r0 = (h - rows) // 2
    t1 = time.perf_counter()
    _, c1 = o.render(0, 1, n_threads=1, px_begin=r0 * w, px_end=(r0 + rows) * w)
    dt1 = time.perf_counter() - t1
    rays1 = c1["rays_primary"] + c1["rays_bounce"]
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "%s, full %dx%d frame at %d spp (%d rays, %.1f s)" % (scn["label"].split(" ")[0], w, h, spp, rays, dt),
            "one_thread_value": rays1 / dt1 / 1e6,
            "one_thread_sample": "rows %d..%d of the frame at 1 spp (%d rays, %.1f s)" % (r0, r0 + rows - 1, rays1, dt1),
            "host_cpu_model": model, "host_logical_cpus": n_logical,
            "threads_note": "OMP_NUM_THREADS=%s" % os.environ.get("OMP_NUM_THREADS", "unset (all logical CPUs)"),
            "nodes_per_ray": (c["nodes_primary"] + c["nodes_bounce"]) / rays,
            "tris_per_ray": (c["tris_primary"] + c["tris_bounce"]) / rays}


def count_lane_loads(api, scn, bvh, w, h, spp, device, rank, world, shard):
    """One more frame, outside the timed region, through the counting build of render_kernel (a context created
    under HPT_COUNT_LANE_LOADS=1): the 16-B lane-loads of exactly this rank's share of the frame, counted in the kernel."""
    import torch
    os.environ["HPT_COUNT_LANE_LOADS"] = "1"
    try:
        be = api.Backend(w, h, device=device)
    finally:
        del os.environ["HPT_COUNT_LANE_LOADS"]
    be.load(scn, bvh)
    if shard == "tiles":
        be.set_shard(rank, world)
    else:
        be.set_sample_shard(rank, world)
    img = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
    be.render(0, spp, img.data_ptr())
    st = be.stats()
    be.close()
    return st["lane_loads"], st["rays_primary"] + st["rays_bounce"], st["last_kernel_ms"]


def iterative_rate(be, image, calls=32):
    """The reference's live frame loop (main.cpp:130-131) through hpt_first_bounce + calls x (hpt_bounce + hpt_reconstruct_paths)
    on the same scene, untimed as far as `value` goes: Mrays/s of the calls after the first bounce, by the wall clock."""
    be.first_bounce()
    be.synchronize()
    st = be.stats()
    r0 = st["rays_primary"] + st["rays_bounce"]
    t0 = time.perf_counter()
    for _ in range(calls):
        be.bounce()
        be.reconstruct_paths(image.data_ptr())
    be.synchronize()
    dt = time.perf_counter() - t0
    st = be.stats()
    return (st["rays_primary"] + st["rays_bounce"] - r0) / dt / 1e6


def launch_ranks(n):
    """--gpus N without a launcher: N ranks of this script under torch.distributed.run, started before this process has
    touched the GPU (it never does).  Rank 0 prints the line; the exit status is the launcher's (non-zero if any rank failed)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="sponza", help="sponza (C3, headline) | bunny (C2) | cornell (C1) | glossy (C5) | tenm (C4)")
    ap.add_argument("--spp", type=int, default=0, help="override the config's spp (debugging only: makes the number non-headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-count", action="store_true", help="skip the extra, untimed frame through the counting kernel (PMC passes: "
                                                            "only the timed kernel may run); the l1_gather fields are then null")
    ap.add_argument("--no-iter", action="store_true", help="skip the untimed rate of the iterative pipeline (config.iter_mrays_per_s)")
    ap.add_argument("--shard", default="tiles", choices=["tiles", "samples"],
                    help="N>1: tiles = 8x8 tiles round-robin + one all-gather (bit-identical image, the default); "
                         "samples = iterations s %% N == rank + one all-reduce (SURVEY.md 8e secondary mode)")
    ap.add_argument("--reduce-every", type=int, default=0, metavar="K",
                    help="progressive: render K-iteration slices and exchange after each (C5: 16); 0 = one exchange per frame")
    ap.add_argument("--exchange", default="torch", choices=["torch", "native"],
                    help="torch = torch.distributed collectives (RCCL); native = the library's own communicator (RCCL called from C++)")
    ap.add_argument("--force-exchange", action="store_true", help="run the exchange step at N=1 too (one-rank communicator)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")

    # HPT_DIST_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks (ranks then share devices)
    dist_backend = os.environ.get("HPT_DIST_BACKEND", "nccl")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import torch                                   # (importing torch and counting devices does not initialise the GPU)
        have = torch.cuda.device_count()
        if dist_backend == "nccl" and have < args.gpus:
            raise SystemExit("--gpus %d but this node has %d GPU(s) (HPT_DIST_BACKEND=gloo rehearses with shared devices)" % (args.gpus, have))
        sys.exit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    from pathtracer_amd import api, scenes
    from pathtracer_amd import dist as pdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the line would report another n_gpus than the one asked for" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if dist_backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    exchanging = world > 1 or args.force_exchange
    group_up = world > 1 or (args.force_exchange and args.exchange == "torch")
    if group_up:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
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
    t_upload = time.perf_counter()
    be.load(scn, bvh)                                   # H2D copies + the traversal's record layout, outside the timed region
    be.synchronize()
    t_upload = time.perf_counter() - t_upload
    if args.shard == "tiles":
        be.set_shard(rank, world)
    else:
        be.set_sample_shard(rank, world)
    be.set_stream(torch.cuda.current_stream().cuda_stream)
    image = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
    # sample shards, progressive: the rank's own running mean must survive the reduction -> the frame is a buffer of its own
    frame = torch.zeros_like(image) if (args.shard == "samples" and args.reduce_every and exchanging) else None
    exchange = None
    if exchanging:
        exchange = (pdist.NativeExchange(be, image, world, rank) if args.exchange == "native" else pdist.Exchange(be, image, world))

    def do_exchange(spp_end):
        if args.shard == "tiles":
            exchange.gather()                             # pack -> all-gather -> unpack, in stream order under RCCL
        else:
            exchange.reduce(pdist.sample_shard_count(0, spp_end, rank, world), spp_end, frame)

    def step():                                           # one frame, ONE exchange
        be.render(0, spp, image.data_ptr())
        if exchanging:
            do_exchange(spp)

    K = args.reduce_every if 0 < args.reduce_every < spp else 0
    slices = [(k0, min(k0 + K, spp)) for k0 in range(0, spp, K)] if K else []
    slice_events = []

    def progressive_step(record=False):                   # the same frame in slices of K iterations, an exchange after every slice
        for k0, k1 in slices:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if record else None
            if record: ev[0].record()
            be.render(k0, k1, image.data_ptr())
            if record: ev[1].record()
            if exchanging:
                do_exchange(k1)
            if record:
                ev[2].record()
                slice_events.append(ev)

    def fence():
        be.synchronize()
        torch.cuda.synchronize()
        if group_up:
            dist.barrier()
            torch.cuda.synchronize()

    timed_step = (lambda: progressive_step(True)) if K else step
    for _ in range(args.warmup):
        (progressive_step if K else step)()
    fence()
    kernel_ms = []
    rays_p = rays_b = launches = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        timed_step()
        if not K:
            # hpt_get_stats waits for the launch stream and reads the HIP-event time of the render kernel recorded on it
            st = be.stats()
            kernel_ms.append(st["last_kernel_ms"])
            rays_p, rays_b, launches = st["rays_primary"], st["rays_bounce"], st["launches"]
    fence()
    elapsed = time.perf_counter() - t0

    progressive = None
    if K:
        # device times per slice from the stream's events; rays and the kernel time of the roofline from one untimed,
        # unsliced frame (the same paths: the image does not depend on how the frame is cut)
        render_ms = sum(e[0].elapsed_time(e[1]) for e in slice_events) / args.steps
        exch_ms = sum(e[1].elapsed_time(e[2]) for e in slice_events) / args.steps
        step(); fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
            st = be.stats()
            kernel_ms.append(st["last_kernel_ms"])
            rays_p, rays_b, launches = st["rays_primary"], st["rays_bounce"], st["launches"]
        fence()
        single = (time.perf_counter() - t1) / args.steps
        progressive = {"reduce_every": K, "slices": len(slices), "render_ms_per_slice": render_ms / len(slices),
                       "exchange_ms_per_slice": exch_ms / len(slices), "exchange_ms_per_frame": exch_ms,
                       "exchange_bytes_per_slice": w * h * 12,
                       "single_exchange_ms_per_step": single * 1e3,
                       "note": "device times from events on the launch stream; ms_per_step is the progressive frame (%d slices, an "
                               "exchange after each), single_exchange_ms_per_step the same frame with one exchange" % len(slices)}

    # this rank's lane-loads of its share of the frame (counting kernel, one untimed frame); summed over the ranks below
    lane_loads = counted_rays = 0
    if not args.no_count:
        lane_loads, counted_rays, _ = count_lane_loads(api, scn, bvh, w, h, spp, local_rank, rank, world, args.shard)

    tot = torch.tensor([elapsed, float(rays_p), float(rays_b), sum(kernel_ms) / len(kernel_ms), float(lane_loads), float(counted_rays)],
                       dtype=torch.float64, device="cuda" if dist_backend == "nccl" else "cpu")
    if world > 1:
        mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, kern_ms = float(mx[0]), float(mx[3])
        rays_p, rays_b = float(sm[1]), float(sm[2])
        lane_loads, counted_rays = float(sm[4]), float(sm[5])
    else:
        kern_ms = float(tot[3])
    rays_frame = rays_p + rays_b
    if args.no_count:
        lane_loads = counted_rays = None

    iter_rate = None
    if rank == 0 and world == 1 and not args.no_iter:
        iter_rate = iterative_rate(be, torch.zeros_like(image))

    if rank == 0:
        costs_all = json.load(open(os.path.join(ROOT, "bench", "frozen_costs.json")))
        costs = costs_all.get(args.scene)
        n_cu = torch.cuda.get_device_properties(local_rank).multi_processor_count
        scene_mb = (len(bvh["bbox"]) * 40 + len(bvh["triangles"]) * 48) / 1e6
        cache_resident = scene_mb <= 256
        t_kernel = kern_ms * 1e-3
        # Per launch = per rank per frame.  kernel_ms is the slowest rank's; lane-loads are the ranks' sum / world.
        loads_launch = lane_loads / world if lane_loads else None
        # the vector-L1 gather roofline: what the kernel asks of the L1 over the kernel's time
        l1_peak_gbps = L1_GATHER_CEILING * LANE_LOAD_BYTES * n_cu * ENGINE_CLOCK_HZ / 1e9
        counted_gbps = loads_launch * LANE_LOAD_BYTES / t_kernel / 1e9 if loads_launch else None
        # PMC-derived figures (separate rocprofv3 --pmc passes, bench/pmc.sh + bench/pmc_stall.sh): measured at one kernel-source
        # hash; at another hash only the RATIOS are carried over (marked), absolute counts are dropped
        pmc_all = {}
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        khash = kernel_source_hash()
        if os.path.exists(pmc_path):
            pmc_all = json.load(open(pmc_path)).get(args.scene, {})
        pmc_current = bool(pmc_all) and pmc_all.get("kernel_source_hash") == khash
        pmc = pmc_all if (pmc_current and world == 1 and not args.spp) else {}
        traffic = pmc.get("hbm_bytes_per_launch")
        hbm_traffic_gbps = traffic / t_kernel / 1e9 if traffic else None
        # What the L1 really looks up: TCP_TOTAL_CACHE_ACCESSES.  Lanes of a quad that read the same 64-B line are merged into
        # one lookup, so this is BELOW the lane-loads the kernel issues, and it is the figure that falls when the unit is relieved.
        tcp_launch = pmc.get("tcp_lane_accesses_per_launch")
        merge_ratio = pmc_all.get("tcp_accesses_per_issued_lane_load")        # PMC lookups / counted lane-loads, same frame
        if tcp_launch:
            l1_lookups, basis = tcp_launch, "TCP_TOTAL_CACHE_ACCESSES of the PMC pass at this kernel hash (profiles/pmc_traffic.json)"
        elif loads_launch and merge_ratio:
            l1_lookups = loads_launch * merge_ratio
            basis = ("lane-loads counted in this run x %.3f lookups per issued lane-load (TCP_TOTAL_CACHE_ACCESSES / counted, PMC pass at "
                     "kernel hash %s%s)" % (merge_ratio, pmc_all.get("kernel_source_hash"), "" if pmc_current else ", NOT this kernel"))
        else:
            l1_lookups, basis = None, None
        l1_gbps = l1_lookups * LANE_LOAD_BYTES / t_kernel / 1e9 if l1_lookups else None
        roofline = {}
        if cache_resident:
            roofline.update({"bound": "l1_gather", "achieved": l1_gbps, "peak": l1_peak_gbps,
                             "frac": l1_gbps / l1_peak_gbps if l1_gbps else None, "unit": "GB/s", "traffic": traffic,
                             "frac_basis": basis,
                             "note": "BVH+triangles (%.0f MB) are cache-resident: HBM does not bind (hbm_traffic_frac); the vector L1's gather "
                                     "rate is the nearest roof.  achieved = L1 tag lookups (quad-merged lane accesses, the hardware's count) x "
                                     "16 B / kernel time, peak = %.2f lane-loads/clk/CU (profiles/r02/gather_exec.txt: every lane its own line, "
                                     "nothing merges) x 16 B x %d CUs x 2.4 GHz.  Not the only limiter: ~29 %% of the TA's busy cycles are "
                                     "stalls on the cache behind it (ta_stalled_by_tc_frac), L1 miss latency co-limits (DESIGN.md 4)"
                                     % (scene_mb, L1_GATHER_CEILING, n_cu)})
        else:
            roofline.update({"bound": "hbm", "achieved": hbm_traffic_gbps, "peak": HBM_PEAK_GBPS,
                             "frac": hbm_traffic_gbps / HBM_PEAK_GBPS if hbm_traffic_gbps else None, "unit": "GB/s", "traffic": traffic,
                             "frac_basis": "fabric-side bytes of the PMC passes at this kernel hash" if traffic else None,
                             "note": "BVH+triangles (%.0f MB) exceed the 256 MB Infinity Cache and stream from HBM; achieved = fabric-side "
                                     "bytes of the PMC passes (FETCH_SIZE x2 per the guide + WRITE_SIZE, profiles/pmc_traffic.json) over this "
                                     "run's kernel time, null when those passes were not measured at this kernel" % scene_mb})
        roofline.update({"kernel": "render_kernel", "kernel_ms": kern_ms, "dispatches_per_launch": launches,
                         "kernel_source_hash": khash, "pmc_measured_at_this_hash": pmc_current,
                         # vector L1: lookups (hardware) and issued lane-loads (counted in the kernel)
                         "l1_lookups_per_launch": l1_lookups,
                         "l1_lookups_per_clk_per_cu": l1_lookups / t_kernel / n_cu / ENGINE_CLOCK_HZ if l1_lookups else None,
                         "l1_gather_ceiling_lane_loads_per_clk_per_cu": L1_GATHER_CEILING,
                         "useful_lane_loads_per_launch": loads_launch,
                         "useful_lane_loads_per_ray": lane_loads / counted_rays if lane_loads else None,
                         "useful_lane_loads_gbps": counted_gbps,
                         "useful_lane_loads_over_ceiling": counted_gbps / l1_peak_gbps if counted_gbps else None,
                         "l1_lookups_per_issued_lane_load": (tcp_launch / loads_launch) if (tcp_launch and loads_launch) else merge_ratio,
                         "ta_busy_frac": pmc_all.get("ta_busy_frac"), "ta_stalled_by_tc_frac": pmc_all.get("ta_stalled_by_tc_frac"),
                         "td_busy_frac": pmc_all.get("td_busy_frac"), "td_stalled_frac": pmc_all.get("td_stalled_frac"),
                         # HBM side, measured (fabric-side bytes of the PMC passes over this run's kernel time)
                         "hbm_traffic_gbps": hbm_traffic_gbps,
                         "hbm_traffic_frac": hbm_traffic_gbps / HBM_PEAK_GBPS if hbm_traffic_gbps else None,
                         "l2_hit_rate": pmc.get("l2_hit_rate"), "l1_miss_to_l2_frac": pmc.get("l1_miss_to_l2_frac"),
                         "valu_busy_frac_of_cu_time": pmc.get("valu_busy_frac_of_cu_time"), "lane_utilisation": pmc.get("lane_utilisation")})
        if costs:
            # per launch = per rank per frame; with N ranks each launch handles 1/N of the frame's rays
            alg_bytes, alg_flops, bp, bb = reference_costs(costs, rays_p / world, rays_b / world, w * h / world, spp)
            roofline.update({
                # SURVEY.md 8d's reference-order algorithmic bytes: NOT a bound of this kernel on a cache-resident scene
                # (path state stays in registers, nodes come from L1/L2), kept for comparison with the reference's traffic
                "hbm_algorithmic_bytes_per_launch": alg_bytes, "hbm_algorithmic_gbps": alg_bytes / t_kernel / 1e9,
                "bytes_per_primary_ray": bp, "bytes_per_bounce_ray": bb,
                # SURVEY.md 8d's flop count of the reference-order work against the fp32 vector peak
                "valu_algorithmic_tflops": alg_flops / t_kernel / 1e12,
                "valu_algorithmic_frac": alg_flops / t_kernel / 1e12 / VALU_PEAK_TFLOPS})
        sharding = ("8x8 tiles round-robin over %d rank(s), all-gather of slabs" if args.shard == "tiles" else
                    "iterations s %% %d == rank, all-reduce of weighted running means") % world
        if exchanging:
            sharding += "; exchange: " + ("library communicator (RCCL from C++)" if args.exchange == "native" else "torch.distributed (%s)" % dist_backend)
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
                       "sharding": sharding, "bvh_build_s": t_build, "upload_s": t_upload,
                       # the reference's live frame loop (hpt_bounce + hpt_reconstruct_paths per call), untimed as far as `value` goes
                       "iter_mrays_per_s": iter_rate},
            "roofline": roofline,
        }
        if progressive:
            out["progressive"] = progressive
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scn, bvh)
        print(json.dumps(out), flush=True)
    if group_up:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
