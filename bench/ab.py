"""Kernel A/B harness (development tool): times hpt_render of one workload for several builds /
knob settings, each in its own child process on the same GPU, interleaved over rounds.

    python bench/ab.py --spp 16 --rounds 3 name1=path1.so[,ENV=VAL...] name2=...
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %(root)r)
import torch
from pathtracer_amd import api, scenes
scn = scenes.scene(%(scene)r, **%(kw)r)
bvh = api.build_bvh(scn["verts"])
be = api.Backend(scn["width"], scn["height"])
be.load(scn, bvh)
img = torch.zeros(scn["width"] * scn["height"] * 3, dtype=torch.float32, device="cuda")
ms = []
for i in range(%(reps)d + 1):
    be.render(0, %(spp)d, img.data_ptr()); st = be.stats()
    if i: ms.append(st["last_kernel_ms"])
rays = st["rays_primary"] + st["rays_bounce"]
print(json.dumps({"ms": min(ms), "mrays": rays / min(ms) / 1e3, "sum": float(img.double().sum()), "waves": st["grid_waves"]}))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--kw", default="", help="scene keywords, e.g. copies=1000")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    res = {}
    for rnd in range(a.rounds):
        for v in a.variants:
            name, spec = v.split("=", 1)
            parts = spec.split(",")
            env = dict(os.environ)
            if parts[0]:
                env["HPT_LIB"] = os.path.join(ROOT, parts[0])
            for kv in parts[1:]:
                k, val = kv.split("=")
                env[k] = val
            code = CHILD % dict(root=ROOT, scene=a.scene, reps=a.reps, spp=a.spp,
                                kw={k: int(v) for k, v in (kv.split("=") for kv in a.kw.split(",") if kv)})
            out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
            if out.returncode != 0:
                print(name, "FAILED", out.stderr[-500:], flush=True)
                continue
            r = json.loads(out.stdout.strip().splitlines()[-1])
            res.setdefault(name, []).append(r)
            print("round %d %-28s %8.2f ms %8.1f Mrays/s waves %d sum %.6e" % (rnd, name, r["ms"], r["mrays"], r["waves"], r["sum"]), flush=True)
    print(json.dumps({k: {"best_ms": min(x["ms"] for x in v), "best_mrays": max(x["mrays"] for x in v)} for k, v in res.items()}))


if __name__ == "__main__":
    main()
