"""CPU: host-side logic around the boundary (synthetic generators, camera maths, gradient arena)."""
import numpy as np
import pytest
import torch

from dreamscene_amd import multiview, synth
from dreamscene_amd.camera import look_at_camera, orbit_camera


def test_synth_is_seeded_and_shaped():
    a, b = synth.g_object(1000, seed=3, K=16), synth.g_object(1000, seed=3, K=16)
    for k in a:
        assert np.array_equal(a[k], b[k]) and a[k].dtype == np.float32
    assert a["shs"].shape == (1000, 16, 3) and a["opacities"].shape == (1000, 1)
    np.testing.assert_allclose(np.linalg.norm(a["rotations"], axis=1), 1.0, atol=1e-5)
    assert (a["scales"] > 0).all() and (a["opacities"] > 0).all() and (a["opacities"] < 1).all()
    c = synth.g_object(1000, seed=4, K=16)
    assert not np.array_equal(a["means3D"], c["means3D"])
    ind = synth.g_indoor(seed=0, per_wall=200, K=4)
    assert ind["means3D"].shape == (1000, 3) and ind["shs"].shape == (1000, 4, 3)
    init = synth.g_object(100, seed=1, init_opacity=True)
    assert np.allclose(init["opacities"], 0.1)


def test_camera_conventions():
    cam = orbit_camera(5.0, 60.0, 30.0, 0.5, 256, 256)
    # camera centre = inverse(world_view_transform)[3,:3] and is at the orbit radius
    np.testing.assert_allclose(np.linalg.norm(cam.camera_center), 5.0, rtol=1e-5)
    # the origin projects to the image centre and lies at depth = radius (row-vector convention)
    o = np.array([0, 0, 0, 1], np.float32) @ cam.world_view_transform
    np.testing.assert_allclose(o[:3], [0, 0, 5.0], atol=1e-5)
    h = np.array([0, 0, 0, 1], np.float32) @ cam.full_proj_transform
    np.testing.assert_allclose(h[:2] / h[3], [0, 0], atol=1e-6)
    # look_at: the target is on the optical axis, in front
    c2 = look_at_camera([1, 2, 3], [4, 2, 3], 0.9, 128, 128)
    t = np.array([4, 2, 3, 1], np.float32) @ c2.world_view_transform
    np.testing.assert_allclose(t[:3], [0, 0, 3.0], atol=1e-5)
    # +y of the image is "down": a point below the target (smaller world z) lands at larger pixel y
    p = np.array([4, 2, 2.5, 1], np.float32) @ c2.full_proj_transform
    assert p[1] / p[3] > 0


def test_grad_arena_layout_and_alignment():
    P, K = 1001, 16
    a = multiview.GradArena(P, K, "cpu")
    assert a.views["means3D"].shape == (P, 3) and a.views["shs"].shape == (P, K, 3)
    assert a.views["rotations"].shape == (P, 4) and a.views["opacities"].shape == (P, 1)
    base = a.flat.data_ptr()
    spans = []
    for n, v in a.views.items():
        assert v.is_contiguous() and (v.data_ptr() - base) % 16 == 0, n
        spans.append((v.data_ptr() - base, v.numel() * 4))
    spans.sort()
    for (o1, s1), (o2, _) in zip(spans, spans[1:]):
        assert o1 + s1 <= o2                      # regions do not overlap
    a.views["shs"].fill_(2.0)
    assert float(a.flat.sum()) == 2.0 * P * K * 3
    assert multiview.shard_views(8, 1, 4) == [1, 5] and multiview.shard_views(3, 2, 4) == [2]
    assert multiview.allreduce_grads(a) is None   # no process group: no-op


def test_view_stats_single_process():
    g = torch.tensor([[3.0, 4.0, 0.0], [0.0, 0.0, 0.0]])
    r = torch.tensor([5, 0], dtype=torch.int32)
    norm, vis, maxr = multiview.reduce_view_stats(g, r)
    assert norm.tolist() == [5.0, 0.0] and vis.tolist() == [1.0, 0.0] and maxr.tolist() == [5, 0]


def test_bench_self_launch_command():
    """`python bench.py --gpus N` without a launcher environment re-execs itself under torch.distributed.run with one rank
    per GPU on 127.0.0.1 (the flag used to be parsed and ignored)."""
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.self_launch_cmd(8, ["--gpus", "8", "--steps", "20"], 29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29555" and "--nnodes=1" in cmd
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20"]


def test_accumulate_flag_styles_and_bitmap_invalidation():
    """the per-view callback gets `accumulate` by keyword, as a fifth positional (any name, or *args), or not at
    all (four-parameter callbacks: host-side sum, and the arena's reached-row bitmap -- which the callback's K8 overwrote
    with the LAST view's rows -- must be marked invalid so that a rows-format exchange does not drop the other views' rows)."""
    assert multiview._accumulate_style(lambda p, c, g, u, accumulate=False: None) == "keyword"
    assert multiview._accumulate_style(lambda p, c, g, u, **kw: None) == "keyword"
    assert multiview._accumulate_style(lambda p, c, g, u, acc: None) == "positional"
    assert multiview._accumulate_style(lambda p, c, g, u, acc=False: None) == "positional"
    assert multiview._accumulate_style(lambda *a: None) == "positional"
    assert multiview._accumulate_style(lambda p, c, g, u: None) == "none"
    P, K = 130, 16
    arena = multiview.GradArena(P, K, "cpu")
    seen = []

    def legacy(params, cam, grad_out, upstream):            # overwrites grad_out, and (like K8 without accumulate) the bitmap
        for t in grad_out.values():
            t.zero_()
        grad_out["means3D"][cam] = float(cam + 1)
        arena.reached.zero_()
        arena.reached[cam // 64] = 1 << (cam % 64)
        arena.reached_valid = True

    def positional(params, cam, grad_out, upstream, acc):
        seen.append(bool(acc))
        if not acc:
            for t in grad_out.values():
                t.zero_()
        grad_out["means3D"][cam] += 1.0

    multiview.render_views_data_parallel(legacy, {}, [3, 70, 129], [None] * 3, arena)
    assert not arena.reached_valid, "stale bitmap (last view only) left valid over a three-view sum"
    ex = multiview.GradExchange(arena, sh_degree=3, mode="rows")
    assert ex.nonzero_rows().tolist() == [3, 70, 129]
    assert arena.views["means3D"][70, 0] == 71.0 and arena.views["means3D"][3, 0] == 4.0
    multiview.render_views_data_parallel(positional, {}, [1, 2], [None] * 2, arena)
    assert seen == [False, True] and arena.views["means3D"][2, 0] == 1.0 and arena.views["means3D"][3, 0] == 0.0
    # no view for this rank: the arena is cleared and the bitmap no longer trusted
    arena.reached_valid = True
    multiview.render_views_data_parallel(positional, {}, [], [], arena)
    assert not arena.reached_valid and float(arena.flat.abs().sum()) == 0.0


def test_dropin_ring_bookkeeping_without_a_gpu():
    """dreamscene_amd/dropin.py: the captured drop-in path is opt-in, never taken for CPU tensors or ineligible inputs, and a
    slot is leased exactly as long as the call that took it can still run a backward."""
    import gc
    from dreamscene_amd import dropin
    from dreamscene_amd.rasterizer import GaussianRasterizationSettings, RasterContext
    s = GaussianRasterizationSettings(64, 64, 0.5, 0.5, torch.ones(3), 1.0, torch.eye(4), torch.eye(4), 3, torch.zeros(3), False, False)
    P, K = 10, 16
    args = (torch.zeros(P, 3), torch.zeros(P, 3), torch.zeros(P, 1), torch.zeros(P, K, 3), None, torch.ones(P, 3),
            torch.zeros(P, 4), None)
    assert dropin.ENABLED is False                                   # (GSR_DROPIN_GRAPHS is not set under pytest)
    assert not dropin.eligible(s, *args, None)
    assert not dropin.eligible(s, *args, RasterContext(dropin_graphs=True))      # CPU tensors: the eager path raises its error
    # leases
    class Slot:
        busy, stamp = False, 0
    sl = Slot()
    le = dropin._Lease(sl)
    assert sl.busy
    le.release()
    assert not sl.busy
    le2 = dropin._Lease(sl)
    assert sl.busy
    del le2
    gc.collect()
    assert not sl.busy, "a lease that dies with its autograd graph must free the slot"
    # a context's host statistics are shared with its per-call snapshots
    from dreamscene_amd.rasterizer import HostStats
    hs = HostStats()
    rc = RasterContext(host_stats=hs)
    snap = rc.snapshot()
    snap.host_stats.wait_s += 1.5
    assert hs.wait_s == 1.5 and rc.snapshot().host_stats is hs


def test_segment_length_policy(monkeypatch):
    """GsrBinning.seg_len: 128-entry backward items for launches with little total work (one view, small scenes), 256 for
    the large ones; all views of a call share one value; an unknown capacity means 256."""
    from dreamscene_amd import rasterizer as R
    monkeypatch.delenv("GSR_SEG_LEN", raising=False)
    assert R.pick_seg_len(None) == 256 and R.pick_seg_len(0, 4) == 256
    assert R.pick_seg_len(4_600_000, 1) == 128          # one view of C3 through the reference's interface
    assert R.pick_seg_len(4_600_000, 4) == 256          # the 4-view step of C3
    assert R.pick_seg_len(1_000_000, 4) == 128          # 100 k Gaussians @512^2, 4 views
    monkeypatch.setenv("GSR_SEG_LEN", "64")
    assert R.pick_seg_len(50_000_000, 8) == 64
    assert R.RasterContext().seg_len is None and R.RasterContext(seg_len=128).snapshot().seg_len == 128
    from dreamscene_amd import _lib as L
    assert dict(L.GsrBinning._fields_)["seg_len"] is not None and L.GsrBinning.seg_len.offset == L.GsrBinning.bwd_items_cap.offset + 4


def test_side_stream_fork_proof_without_a_gpu():
    """rasterizer._SideStreams: a call may fork from an OLDER event of the caller's stream only when every input is the same
    live tensor object at the same autograd version as when it was last seen complete (the soundness argument of the internal
    streams). Here: the bookkeeping alone, no device."""
    import gc
    from dreamscene_amd import rasterizer as R
    sd = object.__new__(R._SideStreams)
    sd.known, sd.fork, sd.stats = {}, None, dict(calls=0, forks=0, reused_forks=0)
    a, b = torch.zeros(4), torch.ones(3)
    assert not sd.proves_complete([a, b])                 # never seen
    sd.remember([a, b])
    assert sd.proves_complete([a, b]) and sd.proves_complete([b])
    a.add_(1.0)                                           # an in-place write bumps the version: not proven any more
    assert not sd.proves_complete([a, b]) and sd.proves_complete([b])
    sd.remember([a])
    assert sd.proves_complete([a, b])
    c = a.detach()                                        # another OBJECT on the same storage: unknown (and shares the counter)
    assert not sd.proves_complete([c])
    sd.remember([c])
    c.mul_(2.0)                                           # writing through the alias invalidates the original as well
    assert not sd.proves_complete([a])
    # a dead tensor's id may be reused by a new one: the weak reference tells them apart
    t = torch.zeros(5)
    sd.remember([t])
    key = id(t)
    del t
    gc.collect()
    assert sd.known[key][0]() is None
    class Fake:                                           # (an object that lands on the recycled id must not be "known")
        _version = 0
    sd.known[id(Fake)] = sd.known[key]
    assert not sd.proves_complete([Fake])
    # the raw-pointer writer of this repo tells the counters (optim.FusedAdam ends with increment_version)
    p = torch.zeros(3, requires_grad=True)
    sd.remember([p])
    with torch.no_grad():
        torch.autograd.graph.increment_version(p)
    assert not sd.proves_complete([p])
    # policy: off unless asked for
    assert R._side_streams_wanted(None) == R.SIDE_STREAMS_DEFAULT
    assert R._side_streams_wanted(R.RasterContext(side_streams=3)) == 3
    assert R._side_streams_wanted(R.RasterContext(side_streams=99)) == R.SIDE_STREAMS_MAX
    # ONE accelerator of the per-view call at a time, by construction (RasterContext.per_view_accel)
    assert R.per_view_accel(None) == "off"
    assert R.per_view_accel(R.RasterContext(side_streams=2)) == "streams"
    assert R.per_view_accel(R.RasterContext(dropin_graphs=True)) == "graphs"
    assert R.per_view_accel(R.RasterContext(per_view_accel="graphs", side_streams=4)) == "graphs"
    assert R.per_view_accel(R.RasterContext(per_view_accel="off", side_streams=4, dropin_graphs=True)) == "off"
    with pytest.raises(ValueError):
        R.per_view_accel(R.RasterContext(side_streams=2, dropin_graphs=True))
    with pytest.raises(ValueError):
        R.per_view_accel(R.RasterContext(per_view_accel="both"))


def test_bench_picks_the_timed_kernel_instance():
    """bench.py's `roofline.traffic`: the counters of the EXACT template instance that was timed, never a max over a prefix
    (round 4 printed k_render_bwd<128>'s bytes for the timed k_render_bwd<256>)."""
    import bench
    names = ["k_render_bwd<128>", "k_render_bwd<256>", "k_render_fwd<false, 128>", "k_render_fwd<false, 256>",
             "k_preprocess<16, false, (anonymous namespace)::NoScene>", "k_preprocess_views<16>",
             "k_preprocess_bwd_views<16, false, true, 4>"]
    pk = lambda stage, seg, batched: bench.pick_kernel(names, bench.timed_kernel_name(stage, 16, seg, batched),
                                                       bench.STAGE_KERNEL.get(stage), stage)
    assert pk("render_bwd", 256, True) == "k_render_bwd<256>"
    assert pk("render_bwd", 128, False) == "k_render_bwd<128>"
    assert pk("render_fwd", 256, True) == "k_render_fwd<false, 256>"
    assert pk("preprocess", 256, True) == "k_preprocess_views<16>"
    assert pk("preprocess", 128, False) == "k_preprocess<16, false, (anonymous namespace)::NoScene>"
    assert pk("preprocess_bwd", 256, True) == "k_preprocess_bwd_views<16, false, true, 4>"
    # two instances and no exact match: no number rather than a wrong one
    assert bench.pick_kernel(["k_render_bwd<64>", "k_render_bwd<128>"], "k_render_bwd<256>", "k_render_bwd", "render_bwd") is None
    # an old profile with ONE un-templated entry still resolves
    assert bench.pick_kernel(["k_render_bwd", "k_render_fwd<false>"], "k_render_bwd<256>", "k_render_bwd", "render_bwd") == "k_render_bwd"
    # launch-accurate bytes: K1 / K8 read the parameter rows once per launch
    P, N, HW, K, D = 500_000, 3_069_904, 1024 * 1024, 16, 3
    one = bench.algorithmic_bytes("preprocess", P, N, HW, K, D, views=1)
    four = bench.algorithmic_bytes("preprocess", P, N, HW, K, D, views=4)
    assert four < 4 * one and four - one == 3 * P * 48


def test_exchange_capacity_policy_is_a_function_of_what_every_rank_sees():
    """GradExchange's speculated capacities (row messages, multiview._RowMessages) follow ONLY numbers every rank reads from the
    same message headers -- the largest counts -- so that the fixed-size collectives stay equal-sized without a collective:
    they grow to 1.25 x the largest count (rounded), never shrink, never exceed the row set; a poisoned owner message (first
    phase of the sparse reduce-scatter overflowed: its own count unknown) moves only the first capacity."""
    from dreamscene_amd import multiview
    arena = multiview.GradArena(100_000, 16, torch.device("cpu"))
    ex = multiview.GradExchange(arena, sh_degree=3, mode="rows")
    assert ex._rows_cap == 0 and ex.strict and ex.overflowed_steps == 0
    ex._grow_cap(10_000)
    c1 = ex._rows_cap
    assert c1 >= 12_500 and c1 % 1024 == 0
    ex._grow_cap(5_000)
    assert ex._rows_cap == c1                              # never shrinks
    ex._grow_cap(10_000_000)
    assert ex._rows_cap == (100_000 + 1023) // 1024 * 1024 # never beyond the row set
    # slices of the sparse reduce-scatter: a multiple of 64 rows, W of them cover the set
    for W in (2, 3, 8):
        per = ex._slice_rows(W)
        assert per % 64 == 0 and per * W >= arena.P and (per - 64) * W < arena.P + 64 * W


def test_exchange_slice_capacities(monkeypatch):
    from dreamscene_amd import multiview
    import torch.distributed as dist
    arena = multiview.GradArena(100_000, 16, torch.device("cpu"))
    ex = multiview.GradExchange(arena, sh_degree=3, mode="sparse_rs")
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 8)
    ex._rs_caps = [1024, 2048]
    ex._grow_rs_caps(2_000, 3_000)
    assert ex._rs_caps[0] >= 2_500 and ex._rs_caps[1] >= 3_750 and all(c % 512 == 0 for c in ex._rs_caps)
    held = list(ex._rs_caps)
    ex._grow_rs_caps(4_000, 0x7FFFFFFF)                     # poisoned owner message: only the first capacity moves
    assert ex._rs_caps[0] >= 5_000 and ex._rs_caps[1] == held[1]
    lim = (ex._slice_rows(8) + 1023) // 1024 * 1024
    ex._grow_rs_caps(10 ** 8, 10 ** 8)
    assert ex._rs_caps == [lim, lim]


def test_arena_knows_when_its_rows_outside_the_bitmap_are_zero():
    """GradArena.zero_outside_ok (what lets K8 clear only the rows the reached bitmap names, GsrGrads.zero_outside): true of a fresh
    arena and behind a HIP backward that overwrote it; a torch op on the arena or a view of it (version counter), touch(), or an
    accumulate on top of a torch write end it; the next overwriting backward restores it."""
    import torch
    from dreamscene_amd import multiview, rasterizer as R
    a = multiview.GradArena(100, 4, "cpu")
    assert a.zero_outside_ok() and R._arena_zero_outside(a, False) == 1 and R._arena_zero_outside(a, True) == 0
    a.views["shs"].mul_(2.0)
    assert not a.zero_outside_ok() and R._arena_zero_outside(a, False) == 0
    tok = object()
    R._arena_written(a, False, tok)                 # (K8 writes through raw pointers: the counter stays)
    assert a.zero_outside_ok() and a.reached_valid and a._mask_owner is tok
    R._arena_written(a, True, None)                 # an accumulating view: the bitmap is OR-ed, its owner keeps it
    assert a.zero_outside_ok() and a._mask_owner is tok
    a.flat.add_(1.0)
    R._arena_written(a, True, None)                 # ... on top of a torch write: nothing is known any more
    assert not a.zero_outside_ok()
    R._arena_written(a, False, None)
    assert a.zero_outside_ok() and a._mask_owner is None
    a.touch()
    assert not a.zero_outside_ok() and not a.reached_valid
    assert R._arena_zero_outside(None, False) == 0
    # a backward with per-view scales replaces the bitmap without writing the `scales` region: only the regions it wrote are known
    b = multiview.GradArena(100, 4, "cpu")
    R._arena_written(b, False, None, ("means3D", "opacities", "shs", "rotations"))
    assert not b.zero_outside_ok() and b.zero_outside_ok(("means3D", "shs")) and not b.zero_outside_ok(("scales",))
    assert R._arena_zero_outside(b, False, ("means3D", "opacities", "shs", "rotations")) == 1
    assert R._arena_zero_outside(b, False, ("means3D", "opacities", "shs", "rotations", "scales")) == 0
    R._arena_written(b, False, None, None)
    assert b.zero_outside_ok()
    # the slow check of the invariant itself
    assert b.verify_zero_outside()
    b.views["scales"][7, 1] = 1.0
    assert not b.verify_zero_outside() and b.verify_zero_outside(("means3D", "shs"))
    b.reached[0] = 1 << 7
    assert b.verify_zero_outside()
