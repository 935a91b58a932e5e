"""hpt_render_noalloc (include/hpt/hpt.h): the render entry point that allocates nothing.  Called through the
C-ABI and compared bit for bit with hpt_render, the entry point it is a variant of, and with the oracle.

Collected last on purpose (file name): the round-end run is `pytest -x`.
"""
import numpy as np
import pytest
import torch

from conftest import bits, get_scene
from pathtracer_amd import api

pytestmark = pytest.mark.gpu


def _render_both(s, bvh, spp, budget_check=None):
    be = api.Backend(s["width"], s["height"])
    be.load(s, bvh)
    n = s["width"] * s["height"] * 3
    want = torch.zeros(n, dtype=torch.float32, device="cuda")
    be.render(0, spp, want.data_ptr()); be.synchronize()
    ref_stats = be.stats()
    got = torch.full((n,), -7.0, dtype=torch.float32, device="cuda")      # every pixel must be written (spp_begin == 0)
    be.render_noalloc(0, spp, got.data_ptr()); be.synchronize()
    st = be.stats()
    assert (st["rays_primary"], st["rays_bounce"], st["paths"]) == (ref_stats["rays_primary"], ref_stats["rays_bounce"], ref_stats["paths"])
    assert st["last_kernel_ms"] == 0.0                                    # no events are recorded
    return be, want, got, st


def test_noalloc_matches_render_two_calls_in_a_row():
    """Two calls in a row on one GPU, the second with a smaller input (another context, smaller frame, smaller
    tree): both equal hpt_render bit for bit, i.e. nothing of the first call's scratch contents leaks into the
    second.  The first is large enough (512x512 x 48 spp = 12.6 M samples > the 11.2 M the 128-MiB array holds)
    to be cut into several launches over tile ranges."""
    assert api.lib().hpt_noalloc_scratch_bytes() == 128 << 20
    s, bvh, _ = get_scene("sponza", width=512, height=512)
    be1, want, got, st = _render_both(s, bvh, 48)
    assert st["launches"] >= 2
    assert torch.equal(want.view(torch.int32), got.view(torch.int32))
    s2, bvh2, o2 = get_scene("cornell", width=64, height=64)
    be2, want2, got2, st2 = _render_both(s2, bvh2, 5)
    assert st2["launches"] == 1
    assert torch.equal(want2.view(torch.int32), got2.view(torch.int32))
    assert np.array_equal(bits(got2.cpu().numpy()), bits(o2.render(0, 5)[0].reshape(-1)))
    # the first context again, continuing its image: [0,48) then [48,50) == hpt_render over the same ranges
    be1.render(48, 50, want.data_ptr())
    be1.render_noalloc(48, 50, got.data_ptr()); be1.synchronize()
    assert torch.equal(want.view(torch.int32), got.view(torch.int32))


def test_noalloc_deep_stack_and_shards(monkeypatch):
    """The traversal-stack overflow lives in the scratch array too: with an LDS stack of one entry every deeper
    entry of every lane goes through it.  Tile shards and sample shards use the same launch logic."""
    monkeypatch.setenv("HPT_STACK_CAP", "1")
    s, bvh, o = get_scene("random", n=2000, seed=1, width=96, height=96)
    be = api.Backend(96, 96)
    be.load(s, bvh)
    d = torch.zeros(96 * 96 * 3, dtype=torch.float32, device="cuda")
    be.render_noalloc(0, 4, d.data_ptr()); be.synchronize()
    want = o.render(0, 4)[0].reshape(-1)
    assert np.array_equal(bits(d.cpu().numpy()), bits(want))
    acc = torch.zeros_like(d)
    for r in range(3):
        be.set_shard(r, 3)
        part = torch.zeros_like(d)
        be.render_noalloc(0, 4, part.data_ptr()); be.synchronize()
        acc += part
    assert np.array_equal(bits(acc.cpu().numpy()), bits(want))
    be.set_shard(0, 1)
    be.set_sample_shard(1, 2)
    a = torch.zeros_like(d); b = torch.zeros_like(d)
    be.render(0, 7, a.data_ptr()); be.render_noalloc(0, 7, b.data_ptr()); be.synchronize()
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_noalloc_argument_errors():
    s, bvh, _ = get_scene("cornell", width=32, height=32)
    be = api.Backend(32, 32)
    d = torch.zeros(32 * 32 * 3, dtype=torch.float32, device="cuda")
    with pytest.raises(api.HptError, match="set_camera"):
        be.render_noalloc(0, 1, d.data_ptr())
    be.load(s, bvh)
    with pytest.raises(api.HptError, match="bad render range"):
        be.render_noalloc(3, 2, d.data_ptr())
    with pytest.raises(api.HptError, match="bad render range"):
        be.render_noalloc(0, 1, 0)
    be.upload_triangles(bvh["triangles"][:5])             # leaves now point past the triangle array
    with pytest.raises(api.HptError, match="triangles are uploaded"):
        be.render_noalloc(0, 1, d.data_ptr())


def test_noalloc_captures_into_a_hip_graph():
    """What the entry point is for: under HIP graph capture nothing may allocate or synchronise.  hpt_render_noalloc on
    a capturing stream records only memsets and kernel launches, so the capture succeeds (hpt_render's first use, which
    allocates and records events, invalidates one), and every replay of the graph renders the same image bit for bit."""
    s, bvh, _ = get_scene("cornell", width=128, height=128)
    be = api.Backend(128, 128)
    be.load(s, bvh)
    n = 128 * 128 * 3
    want = torch.zeros(n, dtype=torch.float32, device="cuda")
    be.render(0, 6, want.data_ptr()); be.synchronize()
    d = torch.zeros(n, dtype=torch.float32, device="cuda")
    stream = torch.cuda.Stream()
    be.set_stream(stream.cuda_stream)
    with torch.cuda.stream(stream):
        be.render_noalloc(0, 6, d.data_ptr())             # outside capture once (not required: see be2 below)
    stream.synchronize()
    assert torch.equal(d.view(torch.int32), want.view(torch.int32))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        be.render_noalloc(0, 6, d.data_ptr())
    for _ in range(2):
        d.fill_(-3.0)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(d.view(torch.int32), want.view(torch.int32))
    be.set_stream(0)
    # A context whose FIRST render call is the captured one: the uploads have already asked the runtime for the kernel's
    # attribute and occupancy (choose_render() in ensure_pairs / hpt_set_scene), so the capture needs no warm-up call.
    be2 = api.Backend(128, 128)
    be2.load(s, bvh)
    be2.set_stream(stream.cuda_stream)
    d2 = torch.zeros(n, dtype=torch.float32, device="cuda")
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2, stream=stream):
        be2.render_noalloc(0, 6, d2.data_ptr())
    graph2.replay()
    torch.cuda.synchronize()
    assert torch.equal(d2.view(torch.int32), want.view(torch.int32))
    be2.set_stream(0)
