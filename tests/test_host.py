"""CPU tests of the product's host side (C++: BVH builder, OBJ reader, image writer) against the
oracle's restatements, and of the C-ABI surface (load + exported symbols, no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, get_scene
from oracle import binding as orc
from pathtracer_amd import api, scenes


@pytest.mark.parametrize("name,kw", [("cornell", {}), ("random", dict(n=2000, seed=1)), ("random", dict(n=12, seed=3)),
                                     ("random", dict(n=777, seed=9, extent=5.0, size=30.0)),
                                     ("degenerate", dict(n_same=200, n_other=30)), ("bunny", {}), ("sponza", {})])
def test_host_bvh_is_order_exact(name, kw):
    """host/bvh.cpp must reproduce the bvh.cpp restatement byte for byte (node boxes, leaf sizes and
    offsets, child links, axes, triangle order): hit order on ties depends on it."""
    s = scenes.scene(name, **kw)
    b = api.build_bvh(s["verts"])
    ob, oi, oo = orc.build_bvh(s["verts"])
    assert len(b["bbox"]) == len(ob)
    assert b["bbox"].tobytes() == ob.tobytes()
    assert b["info"].tobytes() == oi.tobytes()
    assert np.array_equal(b["order"], oo)
    assert b["triangles"].tobytes() == orc.make_triangles(s["verts"], oo).tobytes()
    # depth reported by the builder = interior nodes on the longest root-to-leaf path
    depth = np.zeros(len(oi), np.int32)
    for k in range(len(oi)):
        if oi["nPrimitives"][k] == 0:
            depth[k + 1] = depth[oi["secondChildOffset"][k]] = depth[k] + 1
    assert b["depth"] == depth.max()


@pytest.mark.parametrize("case", range(16))
def test_host_bvh_randomized_inputs_are_order_exact(case, monkeypatch):
    """Seeded adversarial inputs for the order-exact builder, 1 to 7 host threads: coordinates snapped to a coarse
    grid (ties in the longest-axis choice and in the bucket index), duplicated triangles (identical centroids: the
    n <= 24 leaf branch, bvh.cpp:174-193), all triangles in one plane or on one line (zero extents), huge and tiny
    coordinates."""
    rng = np.random.default_rng(500 + case)
    n = int(rng.choice([12, 13, 24, 25, 26, 100, 999, 5000]))
    kind = case % 8
    v = rng.uniform(-100, 100, (n, 3, 3)).astype(np.float32)
    if kind == 1:
        v = np.round(v / 25.0) * 25.0                                   # coarse grid: many equal centroid coordinates
    elif kind == 2:
        v[n // 3:] = v[0]                                               # two thirds identical triangles
    elif kind == 3:
        v[:, :, 1] = 7.0                                                # all in the plane y = 7
    elif kind == 4:
        v[:, :, 1] = 0.0; v[:, :, 2] = -3.0                             # all on one line
    elif kind == 5:
        v *= np.float32(1e18)
    elif kind == 6:
        v *= np.float32(1e-30)
    elif kind == 7:
        v = np.round(v / 50.0) * 50.0; v[n // 2:] = v[: n - n // 2]     # grid + duplicates
    v = np.ascontiguousarray(v, np.float32)
    monkeypatch.setenv("HPT_BUILD_THREADS", str(1 + case % 7))
    monkeypatch.setenv("HPT_BUILD_TASK_SIZE", str(int(rng.choice([0, 16, 200]))))
    b = api.build_bvh(v)
    ob, oi, oo = orc.build_bvh(v)
    assert len(b["bbox"]) == len(ob) and b["bbox"].tobytes() == ob.tobytes() and b["info"].tobytes() == oi.tobytes()
    assert np.array_equal(b["order"], oo)
    assert b["triangles"].tobytes() == orc.make_triangles(v, oo).tobytes()


@pytest.mark.parametrize("threads,task_size", [(1, 0), (5, 25), (3, 100), (16, 4000)])
@pytest.mark.parametrize("name,kw", [("random", dict(n=2000, seed=1)), ("degenerate", dict(n_same=200, n_other=30)), ("bunny", {})])
def test_host_bvh_parallel_build_is_thread_invariant(monkeypatch, name, kw, threads, task_size):
    """The parallel builder (top levels shared by threads, subtrees as tasks, the subtree holding global
    positions 0..11 first) gives the sequential recursion's arrays whatever the thread count and task cut."""
    s = scenes.scene(name, **kw)
    ob, oi, oo = orc.build_bvh(s["verts"])
    monkeypatch.setenv("HPT_BUILD_THREADS", str(threads))
    if task_size:
        monkeypatch.setenv("HPT_BUILD_TASK_SIZE", str(task_size))
    b = api.build_bvh(s["verts"])
    assert b["bbox"].tobytes() == ob.tobytes()
    assert b["info"].tobytes() == oi.tobytes()
    assert np.array_equal(b["order"], oo)


def test_obj_reader_parallel_chunks(tmp_path, monkeypatch):
    """Files over 1 MB are parsed in line-aligned chunks by several threads: interleaved `v`/`f` lines, negative
    (relative) indices, polygons, `a/b/c` tokens, comments and CRLF line ends must give what a sequential
    parse gives, whatever the thread count.  A face may name a vertex that FOLLOWS it (the reference resolves indices
    after the whole file is read, obj_reader.cpp:77-81): such a file gives the oracle reader's list; an index beyond the
    file's vertices, or a negative one reaching before the first vertex, is rejected."""
    rng = np.random.default_rng(5)
    lines, verts, want = [], [], []
    for blk in range(4000):
        for _ in range(int(rng.integers(3, 9))):
            v = rng.uniform(-50, 50, 3).astype(np.float32)
            verts.append(v)
            lines.append("v %.9g %.9g %.9g%s" % (v[0], v[1], v[2], "\r" if blk % 7 == 0 else ""))
        if blk % 5 == 0:
            lines.append("# comment %d" % blk)
        for _ in range(int(rng.integers(1, 6))):
            k = int(rng.integers(3, 6))
            idx = rng.integers(1, len(verts) + 1, k)
            if rng.random() < 0.5:
                toks = ["%d" % (i - len(verts) - 1) for i in idx]              # negative = relative to the vertices so far
            else:
                toks = ["%d/%d/%d" % (i, i, i) for i in idx]
            lines.append("f " + " ".join(toks))
            for j in range(2, k):
                want += [verts[idx[0] - 1], verts[idx[j - 1] - 1], verts[idx[j] - 1]]
    text = "\n".join(lines) + "\n"
    text = text * 1 + "# padding\n" * max(0, (1200000 - len(text)) // 10)
    p = tmp_path / "big.obj"
    p.write_text(text)
    assert p.stat().st_size > (1 << 20)
    want = np.array(want, np.float32)
    for threads in (1, 7, 16):
        monkeypatch.setenv("HPT_BUILD_THREADS", str(threads))
        got = api.load_obj(str(p))
        assert got.shape == want.shape and np.array_equal(got, want), threads
    for pad in (0, 120000):                                   # the sequential and the chunked path
        q = tmp_path / ("fwd%d.obj" % pad)
        q.write_text("f 3 1 2\nv 0 0 0\nv 1 0 0\nf 1 2 3\nv 0 1 0\nf 2 3 1\n" + "# padding\n" * pad)
        got = api.load_obj(str(q))
        assert np.array_equal(got, orc.load_obj(str(q))[1]) and got.shape == (9, 3)
        q.write_text("v 0 0 0\nv 1 0 0\nf 1 2 4\nv 0 1 0\n" + "# padding\n" * pad)
        with pytest.raises(api.HptError):
            api.load_obj(str(q))
        q.write_text("v 0 0 0\nf -1 -2 -1\nv 1 0 0\nv 0 1 0\n" + "# padding\n" * pad)
        with pytest.raises(api.HptError):
            api.load_obj(str(q))


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_host_code_under_sanitizers(tmp_path, san):
    """The two multi-threaded pieces of the host side (BVH builder: shared top levels + subtree tasks; OBJ
    reader: three passes over chunks) run clean under ThreadSanitizer and under ASan + UBSan."""
    import subprocess
    host = os.path.join(ROOT, "pathtracer_amd", "csrc", "host")
    exe = tmp_path / "san"
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + san, "-fno-sanitize-recover=all", "-I" + os.path.join(ROOT, "include"),
                        "-I" + host, os.path.join(ROOT, "tests", "native", "host_sanitize.cpp"), os.path.join(host, "bvh.cpp"),
                        os.path.join(host, "obj_reader.cpp"), "-o", str(exe), "-pthread"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, HPT_BUILD_THREADS="6", HPT_BUILD_TASK_SIZE="500")
    r = subprocess.run([str(exe), "40000", str(tmp_path / "s.obj")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.stdout + r.stderr)[-3000:]


def test_host_bvh_rejects_small_input():
    with pytest.raises(api.HptError):
        api.build_bvh(np.zeros((11, 3, 3), np.float32))


def test_obj_reader_matches_oracle_and_extends(tmp_path):
    p = tmp_path / "a.obj"
    p.write_text("# c\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1.5\nvn 0 0 1\nf 1 2 3\nf 1 3 4\n")
    n, want = orc.load_obj(str(p))
    got = api.load_obj(str(p))
    assert np.array_equal(got, want)
    q = tmp_path / "b.obj"                      # beyond the reference: v/vt/vn, negative indices, quads
    q.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1/1/1 2/2/2 3/3/3 4/4/4\nf -4 -3 -2\n")
    got = api.load_obj(str(q))
    assert got.shape == (9, 3)
    assert np.array_equal(got[:6], np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32))
    with pytest.raises(api.HptError):
        api.load_obj(str(tmp_path / "missing.obj"))
    r = tmp_path / "c.obj"
    r.write_text("v 0 0 0\nf 1 2 3\n")        # index out of range: the reference reads out of bounds, this one fails
    with pytest.raises(api.HptError):
        api.load_obj(str(r))


@pytest.mark.parametrize("case", range(12))
def test_obj_reader_randomized_against_oracle(tmp_path, case):
    """Seeded files of the kind the reference's load_obj reads (`v x y z`, `f a b c`), with the number spellings
    fscanf accepts (exponents, signs, bare dots), tabs and repeated blanks, blank lines, comments and unknown
    records, with and without a final newline, and padded to exact multiples of the page size (the reader maps
    the file): same vertex list as the oracle's restatement of obj_reader.cpp."""
    rng = np.random.default_rng(900 + case)
    def num(x):
        forms = ["%.9g" % x, "%.8e" % x, "%+.9g" % x, "%.9gE0" % x]
        if x == int(x):
            forms += ["%d" % int(x), "%d." % int(x)]
        if 0 < abs(x) < 1:
            forms.append(("%.9g" % x).replace("0.", ".", 1))
        return forms[int(rng.integers(len(forms)))]
    def gap():
        return [" ", "  ", "\t", " \t "][int(rng.integers(4))]
    nv = int(rng.integers(3, 400)) if case % 3 else int(rng.integers(4000, 6000))     # the big ones are above the mapping threshold (64 KB)
    lines = []
    for i in range(nv):
        x = rng.choice([rng.uniform(-1e3, 1e3), float(rng.integers(-50, 50)), rng.uniform(-1, 1) * 1e-5, rng.uniform(-1, 1) * 1e20], 3)
        lines.append("v" + gap() + num(float(np.float32(x[0]))) + gap() + num(float(np.float32(x[1]))) + gap() + num(float(np.float32(x[2]))))
        if rng.random() < 0.1:
            lines.append(["", "# a comment", "vn 0 0 1", "vt 0.5 0.5", "g group", "s off"][int(rng.integers(6))])
    for _ in range(int(rng.integers(1, 300))):
        a, b, c = rng.integers(1, nv + 1, 3)
        lines.append("f" + gap() + "%d" % a + gap() + "%d" % b + gap() + "%d" % c)
    text = "\n".join(lines) + ("\n" if case % 2 == 0 else "")
    if case % 3 == 0:                                   # exactly k pages: no zero byte behind the last number in a mapping
        page = os.sysconf("SC_PAGESIZE")
        pad = (-len(text)) % page
        text = "#" + "x" * (pad - 2) + "\n" + text if pad >= 2 else text
    p = tmp_path / "r.obj"
    p.write_text(text)
    n, want = orc.load_obj(str(p))
    got = api.load_obj(str(p))
    assert got.shape == want.shape and got.tobytes() == want.tobytes()


def test_png_writer_round_trip(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    img = (rng.random((37, 53, 3), dtype=np.float32) * 1.4 - 0.2).astype(np.float32)
    img[0, 0] = (np.nan, 2.0, -1.0)
    path = tmp_path / "o.png"
    api.save_png(img, str(path))
    back = np.asarray(Image.open(str(path)).convert("RGB"))
    assert np.array_equal(back, orc.tonemap(img))           # (uint8)(255*clip(c,0,1)), png_writer.cpp:11-17


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "hpt", "hpt.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(hpt_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == sorted(api.C_ABI_SYMBOLS)
    L = C.CDLL(api.LIB_PATH)
    for name in declared:
        assert getattr(L, name) is not None


def test_host_header_symbols_and_plain_c_example(tmp_path):
    """include/hpt/host.h: every declared entry point is exported; both headers are plain C, and the C example
    (examples/minimal.c) compiles with -Wall -Wextra -Werror and links against libhpt.so with gcc."""
    import subprocess
    hdr = open(os.path.join(ROOT, "include", "hpt", "host.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(hpt_host_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == ["hpt_host_build_bvh", "hpt_host_load_obj", "hpt_host_save_png", "hpt_host_save_png_rgb8", "hpt_host_save_ppm"]
    L = C.CDLL(api.LIB_PATH)
    for name in declared:
        assert getattr(L, name) is not None
    exe = tmp_path / "minimal"
    r = subprocess.run(["gcc", "-std=c11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "minimal.c"), "-L" + os.path.dirname(api.LIB_PATH), "-lhpt",
                        "-Wl,-rpath," + os.path.dirname(api.LIB_PATH), "-Wl,-rpath-link,/opt/rocm/lib", "-lm", "-o", str(exe)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr


def test_node_loads_form_one_uninterrupted_group():
    """The 16-B loads of one record -- four for a 64-B node, five for a pair of triangles, three for a single one --
    must be consecutive instructions in every traversal kernel of the built library: ANY instruction between them
    (hipcc puts an s_nop there when it lets the last load overwrite its own address register) costs up to 24 % per
    visit, 23 % of the frame on a tree that streams from HBM (profiles/r01_tuning.md; HPT_KEEP_ADDRESS in
    render_kernels.h keeps the compiler from doing it).  Checked on the gfx950 code object inside libhpt.so."""
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump")
    data = open(api.LIB_PATH, "rb").read()
    i, co = 0, None
    while True:                                          # the embedded AMDGPU ELF (e_machine 224)
        i = data.find(b"\x7fELF", i)
        assert i >= 0, "no gfx950 code object in libhpt.so"
        if int.from_bytes(data[i + 18:i + 20], "little") == 224:
            shoff = int.from_bytes(data[i + 40:i + 48], "little")
            size = shoff + int.from_bytes(data[i + 58:i + 60], "little") * int.from_bytes(data[i + 60:i + 62], "little")
            co = data[i:i + size]
            break
        i += 4
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co); f.flush()
        dis = subprocess.run([objdump, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, timeout=120).stdout
    kernels, name = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            name = m.group(1)
            kernels[name] = []
        elif name and line.strip():
            kernels[name].append(line.split("//")[0].strip())
    checked = 0
    for kname, ins in kernels.items():
        if "render_kernel" not in kname and "iter_trace_kernel" not in kname:
            continue
        # loads through an SGPR base and a VGPR offset (nodes, triangle records), by address operands
        by_addr = {}
        for k, text in enumerate(ins):
            m = re.match(r"global_load_dword(?:x[234])? \S+ (v\d+), (s\[\d+:\d+\])", text)
            if m:
                by_addr.setdefault((m.group(1), m.group(2)), []).append(k)
        sizes = []
        for addr, idx in by_addr.items():
            cluster = [idx[0]]
            for k in idx[1:] + [None]:
                if k is not None and k - cluster[-1] <= 8:
                    cluster.append(k)
                    continue
                if len(cluster) > 1:                      # loads of one record: consecutive instructions
                    assert cluster[-1] - cluster[0] == len(cluster) - 1, \
                        "%s: loads of one record interrupted:\n%s" % (kname, "\n".join(ins[cluster[0]:cluster[-1] + 1]))
                    sizes.append(len(cluster))
                cluster = [k]
        # the node fetch (4 loads) and the triangle-record fetch (5: a pair record, 3: a single record)
        assert 4 in sizes and (5 in sizes or 3 in sizes), "%s: load groups %s" % (kname, sizes)
        checked += 1
    assert checked == 14                                  # render_kernel (plain / counting, short / long primitive lists; helper-lane fetch) and iter_trace_kernel, both record forms


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.HptError, match="no CPU fallback|no HIP device"):
        api.Backend(16, 16)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "pathtracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pt_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_tile_exchange_index_math():
    from pathtracer_amd import dist
    for (w, h, world) in [(64, 64, 1), (100, 52, 2), (1920, 1080, 8), (13, 9, 3)]:
        seen = np.zeros(w * h, np.int32)
        for r in range(world):
            idx = dist.slab_pixel_index(w, h, r, world)
            assert len(idx) == dist.slab_tiles(w, h, world) * 64
            np.add.at(seen, idx[idx >= 0], 1)
        assert (seen == 1).all()                # every pixel belongs to exactly one shard
        img = np.random.default_rng(1).random((w * h, 3), dtype=np.float32)
        slabs = np.stack([dist.pack_tiles_np(img, w, h, r, world) for r in range(world)])
        assert np.array_equal(dist.unpack_tiles_np(slabs, w, h, world), img)


def test_integration_forwarding_body_compiles(tmp_path):
    """The replacement body of OpenCL_Manager printed in INTEGRATION.md (what a reference maintainer would add)
    compiles against a declaration of the reference's constructor and nine public methods (opencl_manager.hpp:11-50, restated in
    tests/native/ocl_decl/) and the C-ABI headers.  Boundary check only; it says nothing about parity."""
    import re
    import subprocess
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```cpp\n(// opencl_manager\.cpp, HIP build.*?)```", text, re.S)
    assert m, "forwarding body not found in INTEGRATION.md"
    body = m.group(1)
    for method in ("initialize", "kernel_load", "new_kernel_load", "triangles_to_texture", "bvh_to_texture", "kernel_execute",
                   "initial_ray_kernel_execute", "ray_kernel_execute", "reconstruct_kernel_execute"):
        assert "OpenCL_Manager::%s(" % method in body, method
    src = tmp_path / "opencl_manager_hip.cpp"
    src.write_text(body)
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-c", str(src), "-o", str(tmp_path / "fwd.o"),
                        "-I" + os.path.join(ROOT, "tests", "native", "ocl_decl"), "-I" + os.path.join(ROOT, "include"),
                        "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    # ... and LINKS with a caller that does what pathtracer.cpp:69-70 does (new OpenCL_Manager(); initialize(w, h)):
    # a body without the constructor of opencl_manager.hpp:11 compiles but leaves this caller with an undefined symbol
    assert "OpenCL_Manager::OpenCL_Manager()" in body
    caller = tmp_path / "caller.cpp"
    caller.write_text('#include "opencl_manager.hpp"\nint main() { OpenCL_Manager *m = new OpenCL_Manager(); return m->initialize(4, 4) ? 0 : 1; }\n')
    r = subprocess.run(["g++", "-std=c++17", str(caller), str(tmp_path / "fwd.o"), "-o", str(tmp_path / "caller"),
                        "-I" + os.path.join(ROOT, "tests", "native", "ocl_decl"), "-L" + os.path.join(ROOT, "pathtracer_amd"), "-lhpt",
                        "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "pathtracer_amd"), "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def test_kernel_source_hash_follows_the_kernel_not_the_whole_makefile(tmp_path, monkeypatch):
    """bench.kernel_source_hash() stamps the PMC figures of profiles/pmc_traffic.json: it must change with the kernel
    sources and the compiler flags, and with nothing else (round 3: a new test target in the Makefile made every PMC
    pass look stale)."""
    import shutil
    import bench
    for f in ("pathtracer_amd/csrc/render_kernels.h", "pathtracer_amd/csrc/device_math.h", "pathtracer_amd/csrc/hpt_backend.hip", "Makefile"):
        os.makedirs(os.path.dirname(tmp_path / f), exist_ok=True)
        shutil.copy(os.path.join(ROOT, f), tmp_path / f)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    h0 = bench.kernel_source_hash()
    assert h0 == __import__("importlib").import_module("bench").kernel_source_hash()
    mk = (tmp_path / "Makefile").read_text()
    (tmp_path / "Makefile").write_text(mk + "\nsome_new_target:\n\techo nothing about the kernel\n")
    assert bench.kernel_source_hash() == h0
    (tmp_path / "Makefile").write_text(mk.replace("-fno-slp-vectorize", "-fno-slp-vectorize -DSOMETHING=1"))
    assert bench.kernel_source_hash() != h0
    (tmp_path / "Makefile").write_text(mk.replace("SAMPLE_BUDGET_MB ?= 2048", "SAMPLE_BUDGET_MB ?= 4096"))
    assert bench.kernel_source_hash() != h0
    (tmp_path / "Makefile").write_text(mk)
    with open(tmp_path / "pathtracer_amd/csrc/device_math.h", "a") as f:
        f.write("// x\n")
    assert bench.kernel_source_hash() != h0
    # the stored PMC figures belong to the kernel in the tree
    monkeypatch.setattr(bench, "ROOT", ROOT)
    import json
    stored = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert {v["kernel_source_hash"] for v in stored.values()} == {bench.kernel_source_hash()}
