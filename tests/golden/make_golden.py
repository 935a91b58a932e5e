"""Regenerates tests/golden/*.npz|json from the CPU oracle (oracle/pt_oracle.c) and the scene
generator.  The reference holds no golden vectors and cannot run here, so these
pin oracle + generator + host BVH builder against drift; they are NOT outputs of the reference.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from conftest import bits, get_scene  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    hashes = {}
    for name, kw in [("cornell", {}), ("random", dict(n=2000, seed=1)), ("degenerate", dict(n_same=200, n_other=30)),
                     ("bunny", {}), ("sponza", {})]:
        s, bvh, o = get_scene(name, **kw)
        hashes[name] = dict(n_triangles=int(len(s["verts"])), n_nodes=int(len(bvh["bbox"])), depth=int(bvh["depth"]),
                            max_leaf=int(bvh["info"]["nPrimitives"].max()), verts=sha(s["verts"]),
                            bbox=sha(bvh["bbox"]), order=sha(bvh["order"]),
                            info_nprims=sha(bvh["info"]["nPrimitives"]), triangles=sha(bvh["triangles"]))
    json.dump(hashes, open(os.path.join(HERE, "scene_hashes.json"), "w"), indent=1, sort_keys=True)

    for name, kw, iters in [("cornell", dict(width=64, height=64), (0, 7)),
                            ("random", dict(n=2000, seed=1, width=64, height=64), (0, 41))]:
        s, bvh, o = get_scene(name, **kw)
        out = {}
        for it in iters:
            hit, t, _ = o.primary_hits(it)
            out["hit_%d" % it] = hit
            out["tbits_%d" % it] = bits(t)
        np.savez_compressed(os.path.join(HERE, "%s_64_primary.npz" % name), **out)

    s, bvh, o = get_scene("cornell", width=32, height=32)
    img, cnt = o.render(0, 4, n_threads=1)
    np.savez_compressed(os.path.join(HERE, "cornell_32_radiance_4spp.npz"), bits=bits(img),
                        counters=np.array([cnt[k] for k in sorted(cnt)], np.int64), counter_names=np.array(sorted(cnt)))
    # C5's material and primitive mix (glossy triangles, 64 mirror and 64 glass spheres) at a reduced frame: radiance of
    # 3 samples through every shading branch, plus the closest hits of iteration 0 (primitive ids included)
    s, bvh, o = get_scene("glossy", width=48, height=27)
    img, cnt = o.render(0, 3, n_threads=1)
    hit, t, _ = o.primary_hits(0)
    np.savez_compressed(os.path.join(HERE, "glossy_48x27_radiance_3spp.npz"), bits=bits(img), hit_0=hit, tbits_0=bits(t),
                        counters=np.array([cnt[k] for k in sorted(cnt)], np.int64), counter_names=np.array(sorted(cnt)))
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
