"""-m gpu: the BASELINE.json configurations at their FULL sizes (C2 .. C5).

Two kinds of checks per configuration:
  * against the scalar C oracle on the same seeded inputs (it needs 0.3 .. 4 s per view at these sizes): every integer
    artefact bit-exact -- including n_contrib and the bits of final_T: the hard gates (alpha < 1/255, T < 1e-4) see the
    same bits in both implementations (SEMANTICS.md section 4/6) -- and images and gradients within
    1e-5 * max|ref| of the entry's own tensor for EVERY entry (no outlier allowance; tests/util.py: rel_scale);
  * size-independent properties of the rasterizer that need no oracle: the sorted list is ordered by (tile, depth bits)
    with ties in ascending Gaussian index (stable sort of the emission order) and the tile ranges partition it; the
    pair count is the sum of the tile rectangles; alpha + final_T = 1; image(white bg) - image(black bg) = final_T;
    the forward is bit-reproducible; the backward is linear in the upstream gradients.
The multi-GPU halves of C4 / C5 (one view per GPU + all-reduce) are covered by tests/test_multiview_gloo.py; here the
views of one rank are rendered."""
import os

import numpy as np
import pytest
import torch

from tests.util import oracle_view, rel_scale, same_bits, settings_for

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = 1e-5
OUTLIERS = 0             # entries allowed beyond TOL: none

CONFIGS = {
    "C2": dict(scene="object", P=100_000, res=512, K=16, D=3, cams=[0]),
    "C3": dict(scene="object", P=500_000, res=1024, K=16, D=3, cams=[0]),
    "C4": dict(scene="object", P=500_000, res=800, K=16, D=3, cams=[2, 5]),   # two of the 8 sampled cameras
    "C5": dict(scene="indoor", P=2_000_000, res=1024, K=4, D=1, cams=[1]),
    # the reference's actual initial state: every Gaussian at opacity 0.1 (gs_renderer.py:598) => ~87 layers blend
    # before T < 1e-4 stops a pixel (SURVEY.md section 8d "init" variant)
    "C2-init": dict(scene="object", P=100_000, res=512, K=16, D=3, cams=[0], init_opacity=True),
    "C3-init": dict(scene="object", P=500_000, res=1024, K=16, D=3, cams=[0], init_opacity=True),
    # what the reference's training augmentation hands the rasterizer (scene_gaussian.py:1005-1008): per-axis scale noise
    # s + n * (sqrt(0.2) s / 4), then clamp(.., 0.0) -- some axes collapse to EXACTLY zero (flat / needle-shaped splats
    # whose conic is ill-conditioned), on top of a share of strongly anisotropic ones
    # (Round 3 held dL/drotations and dL/dscales to 3e-5 here: the ORDER in which K7's fp32 atomics arrived moved their worst
    # entry between 2.7e-6 and 1.5e-5 from run to run. K7's wave results are now added across waves in double -- the backward
    # is bit-reproducible, test_backward_is_bit_reproducible below -- and every tensor is back at 1e-5: measured 1.4e-6.)
    "C2-needles": dict(scene="object", P=100_000, res=512, K=16, D=3, cams=[0], needles=True, smooth_upstream=True),
}
_scene_cache = {}


def _scene(cfg):
    from dreamscene_amd import synth
    key = (cfg["scene"], cfg["P"], bool(cfg.get("init_opacity")), bool(cfg.get("needles")))
    if key not in _scene_cache:
        _scene_cache.clear()          # one full-size scene at a time on the host
        if cfg["scene"] == "object":
            _scene_cache[key] = synth.g_object(cfg["P"], seed=0, K=cfg["K"], init_opacity=bool(cfg.get("init_opacity")))
        else:
            _scene_cache[key] = synth.g_indoor(seed=0, per_wall=cfg["P"] // 5, K=cfg["K"])
        if cfg.get("needles"):
            g = _scene_cache[key]
            rng = np.random.default_rng(77)
            s = g["scales"].astype(np.float32)
            s[::7, 1] *= 0.01                                     # 1 : 100 needles / flat discs
            s[3::11, 0] *= 3.0                                    # elongated along one axis
            noise = rng.standard_normal(s.shape).astype(np.float32) * 8.0   # 8x the trainers' sigma: ~13 % of the axes clamp to 0
            g["scales"] = np.maximum(s + noise * (np.float32(np.sqrt(0.2)) * s / 4.0), 0.0).astype(np.float32)
            assert (g["scales"] == 0.0).mean() > 0.05
    g = _scene_cache[key]
    H = W = cfg["res"]
    cams = (synth.object_cameras if cfg["scene"] == "object" else synth.indoor_cameras)(8, H, W)
    return g, [cams[i] for i in cfg["cams"]]


def _upstream(cfg, H, W, seed):
    """dL/dimage, dL/d(depth_alpha). Default: white noise. `smooth_upstream`: a smooth field of the
    same magnitude -- what a loss against a target image produces. A streak-shaped splat covers 10^4 pixels; with
    white-noise gradients its per-pixel terms cancel to 1 % of their magnitude and ANY fp32 accumulation order (the
    lineage's per-thread atomics first of all) is only good to ~1e-5 .. 1e-4 of the exact sum, which says nothing about
    the arithmetic under test: the needle case is about the conditioning of the covariance chain, not of noise sums."""
    from dreamscene_amd import synth
    if not cfg.get("smooth_upstream"):
        return synth.upstream_grads(H, W, seed=seed)
    y, x = np.mgrid[0:H, 0:W].astype(np.float32)
    gi = np.stack([1e-3 * np.sin(0.031 * x + c) * np.cos(0.023 * y - 0.5 * c) for c in range(3)]).astype(np.float32)
    gda = np.stack([1e-3 * np.cos(0.017 * x + 0.029 * y), 1e-3 * np.sin(0.013 * x - 0.019 * y + 1.0)]).astype(np.float32)
    return gi, gda


def _forward(g_dev, cam, bg, D, want_keys=True, rc=None):
    from dreamscene_amd import rasterizer as R
    s = settings_for(cam, bg, D, DEV)
    out, st = R.rasterize_forward_raw(s, g_dev["means3D"], g_dev["opacities"], g_dev["shs"], None, g_dev["scales"],
                                      g_dev["rotations"], None, want_keys=want_keys, rc=rc)
    torch.cuda.synchronize()
    return out, st


def _frac_over(a, ref, tol=TOL):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    ref = np.asarray(ref, dtype=np.float64).reshape(-1)
    scale = rel_scale(ref)
    e = np.abs(a - ref)
    return float((e > tol * scale).mean()), float(e.max() / scale)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_size_vs_oracle(built_lib, c_oracle, name):
    from dreamscene_amd import rasterizer as R, synth
    cfg = CONFIGS[name]
    g, cams = _scene(cfg)
    P, K, D = g["means3D"].shape[0], cfg["K"], cfg["D"]
    g_dev = {k: torch.tensor(v, device=DEV) for k, v in g.items()}
    bg = np.array([1.0, 1.0, 1.0], np.float32)
    for ci, cam in enumerate(cams):
        H, W = cam.image_height, cam.image_width
        gi, gda = _upstream(cfg, H, W, ci)
        out, st = _forward(g_dev, cam, bg, D)
        o = R.rasterize_backward_raw(st, torch.tensor(gi, device=DEV), torch.tensor(gda, device=DEV))
        torch.cuda.synchronize()
        v = oracle_view(c_oracle, cam, P, K, D, bg)
        f = c_oracle.forward(v, g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
        b = c_oracle.backward(v, f, gi, gda, g["means3D"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
        # integer artefacts: bit-exact
        assert np.array_equal(out["radii"].cpu().numpy(), f["radii"]), "radii"
        assert np.array_equal(out["tiles_touched"].cpu().numpy().view(np.uint32), f["tiles_touched"]), "tiles_touched"
        assert out["N"] == f["N"] and out["N"] > 0, "pair count"
        assert np.array_equal(out["point_list"].cpu().numpy().view(np.uint32), f["point_list"]), "sorted value list"
        assert np.array_equal(out["keys_sorted"].cpu().numpy().view(np.uint64), f["keys"]), "sorted keys"
        assert np.array_equal(out["ranges"].cpu().numpy().view(np.uint32), f["ranges"]), "tile ranges"
        # images and gradients
        report = {}
        for nm, a, r in (("image", out["color"], f["image"]), ("depth_alpha", out["depth_alpha"], f["depth_alpha"]),
                         ("final_T", out["final_T"], f["final_T"])):
            report[nm] = _frac_over(a.cpu().numpy(), r)
        for hk, ok in (("dL_dmeans3D", "dL_dmeans3D"), ("dL_dmeans2D", "dL_dmeans2D"), ("dL_dopacities", "dL_dopacity"),
                       ("dL_dshs", "dL_dshs"), ("dL_dscales", "dL_dscales"), ("dL_drotations", "dL_drotations")):
            report[hk] = _frac_over(o[hk].cpu().numpy(), b[ok])
        print(f"[{name} cam {cfg['cams'][ci]}] P={P} N={out['N']} " +
              " ".join(f"{k}:{m:.1e}/{fr:.0e}" for k, (fr, m) in report.items()))
        nc = out["n_contrib"].cpu().numpy().view(np.uint32)
        assert np.array_equal(nc, f["n_contrib"]), f"n_contrib differs at {(nc != f['n_contrib']).sum()} pixels"
        assert np.array_equal(out["final_T"].cpu().numpy().view(np.uint32), f["final_T"].view(np.uint32)), "final_T bits"
        for k, (frac, mx) in report.items():
            assert mx <= TOL and frac <= OUTLIERS, f"{name} {k}: {frac:.2e} of the entries beyond 1e-5 (max {mx:.2e})"
        del out, st, o


@pytest.mark.parametrize("name", ["C2-needles", "C2-init"])
def test_backward_is_bit_reproducible(built_lib, name):
    """Eight forward + backward runs of the same inputs give the same BITS in every gradient: K7 reduces a splat's sums
    over the pixels of a wave in a fixed order and adds the waves' results in double (global_atomic_add_f64; the sum of
    fp32 addends is exact in double whatever the arrival order), K8 rounds once. Needles = the case whose worst
    dL/drotations entry used to move between 2.7e-6 and 1.5e-5 with the order of fp32 atomics; init = ~87 layers per pixel,
    the most cross-wave traffic per Gaussian."""
    from dreamscene_amd import rasterizer as R
    cfg = CONFIGS[name]
    g, cams = _scene(cfg)
    cam, D = cams[0], cfg["D"]
    g_dev = {k: torch.tensor(v, device=DEV) for k, v in g.items()}
    bg = np.array([1.0, 1.0, 1.0], np.float32)
    gi, gda = _upstream(cfg, cam.image_height, cam.image_width, 0)
    gi_d, gda_d = torch.tensor(gi, device=DEV), torch.tensor(gda, device=DEV)
    first = None
    for rep in range(8):
        out, st = _forward(g_dev, cam, bg, D, want_keys=False)
        o = R.rasterize_backward_raw(st, gi_d, gda_d)
        torch.cuda.synchronize()
        got = {k: o[k].clone() for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacities", "dL_dshs", "dL_dscales", "dL_drotations")}
        if first is None:
            first = got
            assert all(float(v.abs().max()) > 0 for v in got.values())
        else:
            for k in first:
                same_bits(first[k], got[k], f"{name}: {k}, run {rep} vs run 0")
        del out, st, o


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_size_properties(built_lib, name):
    from dreamscene_amd import rasterizer as R, synth
    cfg = CONFIGS[name]
    g, cams = _scene(cfg)
    cam = cams[0]
    D = cfg["D"]
    H, W = cam.image_height, cam.image_width
    g_dev = {k: torch.tensor(v, device=DEV) for k, v in g.items()}
    white, black = np.ones(3, np.float32), np.zeros(3, np.float32)
    out, st = _forward(g_dev, cam, white, D)
    N = int(out["N"])

    # -- binning: order, stability, ranges, pair count
    keys = out["keys_sorted"].view(torch.int64)           # tile << 32 | depth bits: < 2^45, positive as int64
    vals = out["point_list"].view(torch.int32).to(torch.int64)
    assert keys.numel() == N and vals.numel() == N
    assert bool((keys[1:] >= keys[:-1]).all()), "sorted list is not ordered by (tile, depth)"
    tie = keys[1:] == keys[:-1]
    assert bool((vals[1:][tie] > vals[:-1][tie]).all()), "equal (tile, depth) keys are not in ascending Gaussian index"
    tiles = ((H + 15) // 16) * ((W + 15) // 16)
    tile_of = keys >> 32
    t = torch.arange(tiles, device=DEV, dtype=torch.int64)
    lo = torch.searchsorted(tile_of, t, right=False)
    hi = torch.searchsorted(tile_of, t, right=True)
    ranges = out["ranges"].view(torch.int32).to(torch.int64).reshape(tiles, 2)
    nonempty = hi > lo
    assert bool((ranges[nonempty, 0] == lo[nonempty]).all() and (ranges[nonempty, 1] == hi[nonempty]).all()), "ranges"
    assert bool((ranges[~nonempty, 0] == ranges[~nonempty, 1]).all()), "empty tiles must have empty ranges"
    tt = out["tiles_touched"].view(torch.int32).to(torch.int64)
    assert int(tt.sum()) == N, "pair count != sum of the tile rectangles"
    assert bool(((tt > 0) == (out["radii"] > 0)).all()), "visible <=> touches a tile"
    # depth bits of a pair are the depth of its Gaussian
    depth_bits = out["splat"][:, 6].contiguous().view(torch.int32).to(torch.int64)
    assert bool(((keys & 0xFFFFFFFF) == depth_bits[vals]).all()), "key depth != depth of the Gaussian"

    # -- compositing identities and bit-reproducibility, for both forward variants (the host otherwise picks the variant
    #    per call from the previous view's statistics; the two differ in the association of the transmittance product)
    st2 = None
    for mode in (0, 1):
        rc = R.RasterContext(fwd_variant=mode)
        out_w, st2 = _forward(g_dev, cam, white, D, rc=rc)
        out_b, _ = _forward(g_dev, cam, black, D, want_keys=False, rc=rc)
        out_r, _ = _forward(g_dev, cam, white, D, rc=rc)
        da = out_w["depth_alpha"]
        assert float((da[1] + out_w["final_T"] - 1.0).abs().max()) <= 1e-5, "alpha + final_T != 1"
        assert float(((out_w["color"] - out_b["color"]) - out_w["final_T"][None]).abs().max()) <= 1e-6, "bg term != final_T"
        assert torch.equal(out_w["final_T"], out_b["final_T"]) and torch.equal(out_w["n_contrib"], out_b["n_contrib"])
        for k in ("color", "depth_alpha", "final_T", "n_contrib", "radii", "point_list", "ranges"):
            assert torch.equal(out_w[k], out_r[k]), f"forward (variant {mode}) not reproducible: {k}"
        del out_b, out_r

    # -- the backward is linear in the upstream gradients (fp32 atomics: compared at 1e-4 of the tensor's scale)
    g1 = [torch.tensor(x, device=DEV) for x in synth.upstream_grads(H, W, seed=11)]
    g2 = [torch.tensor(x, device=DEV) for x in synth.upstream_grads(H, W, seed=12)]
    names = ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacities", "dL_dshs", "dL_dscales", "dL_drotations")

    def bwd(gi, gda):
        o = R.rasterize_backward_raw(st2, gi, gda)    # (several backward passes over one saved forward state)
        torch.cuda.synchronize()
        return {k: o[k].clone() for k in names}
    b1, b2 = bwd(*g1), bwd(*g2)
    b3 = bwd(2.0 * g1[0] - 3.0 * g2[0], 2.0 * g1[1] - 3.0 * g2[1])
    for k in names:
        ref = 2.0 * b1[k].double() - 3.0 * b2[k].double()
        scale = rel_scale(ref)
        e = float((b3[k].double() - ref).abs().max())
        assert e <= 1e-4 * scale, f"backward not linear in {k}: {e:.2e} (scale {scale:.2e})"
    # culled Gaussians receive exactly zero
    culled = out["radii"] == 0
    for k in names:
        assert float(b1[k][culled].abs().max() if bool(culled.any()) else 0.0) == 0.0, f"{k} of culled Gaussians"


@pytest.mark.parametrize("name,n_views", [("C4", 8), ("C5", 4)])
def test_multi_view_sum_at_full_size_equals_the_oracle_sum(built_lib, c_oracle, name, n_views):
    """BASELINE.json configs[3] / [4] are multi-GPU: 8 cameras @800^2 one per GPU (C4), the 2 M indoor scene with 4 cameras on 4
    GPUs (C5), the per-view parameter gradients summed by one all-reduce. What that all-reduce has to deliver is the SUM over
    the step's views (training/object_trainer.py:302-382); here all views of the configuration go through ONE batched call
    on one GPU, summed on the device into the GradArena (K8's accumulate form -- the buffer the exchange works on), and the
    arena is compared with the sum of the oracle's per-view gradients: every view of C4 / C5 at full size, not the one or two
    cameras of test_full_size_vs_oracle; per-view images and radii against the oracle as well."""
    from dreamscene_amd import multiview, synth
    from dreamscene_amd.rasterizer import RasterContext
    from dreamscene_amd.views import GaussianRasterizerViews
    cfg = dict(CONFIGS[name], cams=list(range(n_views)))
    g, cams = _scene(cfg)
    P, K, D = g["means3D"].shape[0], cfg["K"], cfg["D"]
    bg = np.array([1.0, 1.0, 1.0], np.float32)
    params = {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in g.items()}
    ups = [_upstream(cfg, cam.image_height, cam.image_width, i) for i, cam in enumerate(cams)]
    arena = multiview.GradArena(P, K, torch.device(DEV))
    sl = [settings_for(cam, bg, D, DEV) for cam in cams]
    rast = GaussianRasterizerViews(sl, context=RasterContext(grad_arena=arena))
    for rep in range(2):                # (the first call of a new (P, H, W) runs view by view and learns the pair counts)
        m2d = torch.zeros((n_views, P, 3), device=DEV, requires_grad=True)
        outs = rast(means3D=params["means3D"], means2D=m2d, opacities=params["opacities"], shs=params["shs"],
                    scales=params["scales"], rotations=params["rotations"])
        ts, gs = [], []
        for (img, _, da), (gi, gda) in zip(outs, ups):
            ts += [img, da]
            gs += [torch.tensor(gi, device=DEV), torch.tensor(gda, device=DEV)]
        (g2d,) = torch.autograd.grad(ts, [m2d], gs)
        torch.cuda.synchronize()
    ref = {k: None for k in ("dL_dmeans3D", "dL_dshs", "dL_dopacity", "dL_dscales", "dL_drotations")}
    for j, cam in enumerate(cams):
        v = oracle_view(c_oracle, cam, P, K, D, bg)
        f = c_oracle.forward(v, g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
        b = c_oracle.backward(v, f, ups[j][0], ups[j][1], g["means3D"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
        assert np.array_equal(outs[j][1].cpu().numpy(), f["radii"]), f"view {j}: radii"
        fr, mx = _frac_over(outs[j][0].detach().cpu().numpy(), f["image"])
        assert mx <= TOL and fr <= OUTLIERS, f"view {j}: image {mx:.2e}"
        fr, mx = _frac_over(g2d[j].cpu().numpy(), b["dL_dmeans2D"])
        assert mx <= TOL and fr <= OUTLIERS, f"view {j}: dL/dmeans2D {mx:.2e}"
        for k in ref:
            x = np.asarray(b[k], dtype=np.float64)
            ref[k] = x if ref[k] is None else ref[k] + x
        del f, b
    got = dict(dL_dmeans3D=arena.views["means3D"], dL_dshs=arena.views["shs"], dL_dopacity=arena.views["opacities"],
               dL_dscales=arena.views["scales"], dL_drotations=arena.views["rotations"])
    for k, r in ref.items():
        fr, mx = _frac_over(got[k].cpu().numpy().reshape(r.shape), r)
        print(f"[{name} sum of {n_views} views] {k}: {mx:.1e}")
        assert mx <= TOL and fr <= OUTLIERS, f"{name}: sum over {n_views} views of {k}: max {mx:.2e} ({fr:.1e} of the entries beyond 1e-5)"


def test_c3_four_views_through_one_captured_call_equal_the_oracle_sum(built_lib, c_oracle):
    """The path `python bench.py` TIMES: BASELINE.json configs[2] -- C3, 500 k Gaussians @1024^2 -- with the 4
    views of a step through ONE `graph.CapturedViews` call: K1 / K8 of all views in one launch each, K6 / K7 of the batch (256-entry
    items at this size), gradients summed in the GradArena, the launches REPLAYED from captured hipGraphs (the third call and
    every later one; the first two run eagerly and learn the pair counts). Compared after a replay: every view's radii (bit-
    exact), image and means2D gradient against the scalar C oracle, and the arena against the float64 sum of the oracle's
    per-view gradients -- what the optimizer consumes (training/object_trainer.py:302-382)."""
    from concurrent.futures import ThreadPoolExecutor
    from dreamscene_amd import multiview, synth
    from dreamscene_amd.graph import CapturedViews
    from dreamscene_amd.rasterizer import RasterContext
    n_views = 4
    cfg = dict(CONFIGS["C3"], cams=list(range(n_views)))
    g, cams = _scene(cfg)
    P, K, D = g["means3D"].shape[0], cfg["K"], cfg["D"]
    bg = np.array([1.0, 1.0, 1.0], np.float32)
    params = {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in g.items()}
    ups = [_upstream(cfg, cam.image_height, cam.image_width, i) for i, cam in enumerate(cams)]
    ups_dev = [(torch.tensor(a, device=DEV), torch.tensor(b, device=DEV)) for a, b in ups]
    arena = multiview.GradArena(P, K, torch.device(DEV))
    sl = [settings_for(cam, bg, D, DEV) for cam in cams]
    rast = CapturedViews(context=RasterContext(grad_arena=arena))
    for rep in range(5):
        m2d = torch.zeros((n_views, P, 3), device=DEV, requires_grad=True)
        outs = rast(sl, means3D=params["means3D"], means2D=m2d, opacities=params["opacities"], shs=params["shs"],
                    scales=params["scales"], rotations=params["rotations"])
        ts, gs = [], []
        for (img, _, da), (gi, gda) in zip(outs, ups_dev):
            ts += [img, da]
            gs += [gi, gda]
        (g2d,) = torch.autograd.grad(ts, [m2d], gs)
        torch.cuda.synchronize()
    assert rast.stats["replays"] >= 2 and rast.stats["overflows"] == 0, rast.stats     # the compared step WAS a graph replay
    seg = int(rast._cap.states[0].binning.seg_len)
    assert seg == 256, seg                                                              # the <256> kernels, as in the timed region

    def one(j):          # (ctypes releases the GIL: the four scalar oracle views run side by side)
        v = oracle_view(c_oracle, cams[j], P, K, D, bg)
        f = c_oracle.forward(v, g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
        b = c_oracle.backward(v, f, ups[j][0], ups[j][1], g["means3D"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
        return f, b
    with ThreadPoolExecutor(max_workers=n_views) as ex:
        res = list(ex.map(one, range(n_views)))
    ref = {k: None for k in ("dL_dmeans3D", "dL_dshs", "dL_dopacity", "dL_dscales", "dL_drotations")}
    for j, (f, b) in enumerate(res):
        assert np.array_equal(outs[j][1].cpu().numpy(), f["radii"]), f"view {j}: radii"
        fr, mx = _frac_over(outs[j][0].detach().cpu().numpy(), f["image"])
        assert mx <= TOL and fr <= OUTLIERS, f"view {j}: image {mx:.2e}"
        fr, mx = _frac_over(outs[j][2].detach().cpu().numpy(), f["depth_alpha"])
        assert mx <= TOL and fr <= OUTLIERS, f"view {j}: depth_alpha {mx:.2e}"
        fr, mx = _frac_over(g2d[j].cpu().numpy(), b["dL_dmeans2D"])
        assert mx <= TOL and fr <= OUTLIERS, f"view {j}: dL/dmeans2D {mx:.2e}"
        for k in ref:
            x = np.asarray(b[k], dtype=np.float64)
            ref[k] = x if ref[k] is None else ref[k] + x
    got = dict(dL_dmeans3D=arena.views["means3D"], dL_dshs=arena.views["shs"], dL_dopacity=arena.views["opacities"],
               dL_dscales=arena.views["scales"], dL_drotations=arena.views["rotations"])
    for k, r in ref.items():
        fr, mx = _frac_over(got[k].cpu().numpy().reshape(r.shape), r)
        print(f"[C3, 4 views, one captured call] {k}: {mx:.1e}")
        assert mx <= TOL and fr <= OUTLIERS, f"C3 captured: sum over 4 views of {k}: max {mx:.2e} ({fr:.1e} of the entries beyond 1e-5)"


def test_c2_vs_the_independent_float64_autograd_oracle(built_lib):
    """BASELINE.json configs[1] at its full size (100 k Gaussians @512^2) against the INDEPENDENT restatement: the vectorised
    PyTorch oracle in float64 with libm's exp and autograd's backward (oracle/torch_oracle.py) -- no expression tree, no
    operator order and no hand-derived chain rule in common with the HIP kernels (the scalar C oracle shares the defined
    exp / power arithmetic with them by construction, SEMANTICS.md section 4; this one shares nothing but the algorithm).
    float64 takes a hard gate the other way on about one (pixel, splat) pair per 10^6 pixels and a radius = ceil(3 sqrt(l))
    the other way on a handful of Gaussians; everything that is not downstream of such a flip must agree at 1e-5. Measured:
    1 pixel of 262 144 (1.9e-5), one dL/dscales entry at 2.1e-5, all other entries of all tensors <= 7e-6. Allowed: <= 4
    pixels / entries per tensor beyond 1e-5, none beyond 1e-4 (gradients) / 4e-3 (the alpha_min T step of a flipped gate)."""
    from dreamscene_amd import rasterizer as R, synth
    from tests.test_oracle_consistency import _torch_run
    P, K, D, res = 100_000, 16, 3, 512
    g = synth.g_object(P, seed=0, K=K)
    cam = synth.object_cameras(1, res, res)[0]
    bg = np.ones(3, np.float32)
    gi, gda = synth.upstream_grads(res, res, 0)
    torch.set_num_threads(min(64, max(1, (os.cpu_count() or 2) // 2)))
    r = _torch_run(g, cam, bg, D, gi=gi, gda=gda, cam_grad=False)
    t = {k: torch.tensor(v, device=DEV) for k, v in g.items()}
    out, st = _forward(t, cam, bg, D, want_keys=False)
    o = R.rasterize_backward_raw(st, torch.tensor(gi, device=DEV), torch.tensor(gda, device=DEV))
    torch.cuda.synchronize()
    dr = np.abs(out["radii"].cpu().numpy().astype(np.int64) - r["radii"].astype(np.int64))
    assert dr.max() <= 1 and int((dr > 0).sum()) <= 8, (int(dr.max()), int((dr > 0).sum()))
    nc = out["n_contrib"].cpu().numpy().view(np.uint32)
    assert int((nc != r["aux"]["n_contrib"]).sum()) <= 8, "float64 and the HIP path disagree on more than a handful of gates"
    d_img = np.abs(out["color"].cpu().numpy().astype(np.float64) - r["img"]).max(axis=0)
    assert int((d_img > 1e-5).sum()) <= 4 and float(d_img.max()) <= 4e-3, (int((d_img > 1e-5).sum()), float(d_img.max()))
    d_da = np.abs(out["depth_alpha"].cpu().numpy().astype(np.float64) - r["da"]).max(axis=0)
    sc_da = rel_scale(r["da"])
    assert int((d_da > 1e-5 * sc_da).sum()) <= 4 and float(d_da.max()) <= 4e-3 * sc_da
    for tk, hk in (("means3D", "dL_dmeans3D"), ("scales", "dL_dscales"), ("rotations", "dL_drotations"),
                   ("opacities", "dL_dopacities"), ("shs", "dL_dshs"), ("means2D", "dL_dmeans2D")):
        ref = np.asarray(r["grads"][tk], dtype=np.float64)
        e = np.abs(o[hk].cpu().numpy().astype(np.float64).reshape(ref.shape) - ref)
        sc = rel_scale(ref)
        n_over = int((e > 1e-5 * sc).sum())
        print(f"[C2 vs float64 autograd] {hk}: max {e.max() / sc:.1e}, {n_over} entries beyond 1e-5")
        assert n_over <= 4 and float(e.max()) <= 1e-4 * sc, f"{hk}: {n_over} entries beyond 1e-5, max {e.max() / sc:.2e}"


def _window_vs_float64(P, res, win, init_opacity, radii_allow, min_depth, label, threads, flips_allow=4, beside_fp32=False):
    """The HIP path against the independent float64 autograd oracle on a crop: the full scene and camera, the loss restricted to the
    pixels of the tile window `win` (upstream gradients zero elsewhere), the oracle compositing only those tiles."""
    from dreamscene_amd import rasterizer as R, synth
    from tests.test_oracle_consistency import _torch_run
    K, D = 16, 3
    y0, y1, x0, x1 = win[0] * 16, win[1] * 16, win[2] * 16, win[3] * 16
    g = synth.g_object(P, seed=0, K=K, init_opacity=init_opacity)
    cam = synth.object_cameras(1, res, res)[0]
    bg = np.ones(3, np.float32)
    gi, gda = synth.upstream_grads(res, res, 0)
    mask = np.zeros((res, res), np.float32)
    mask[y0:y1, x0:x1] = 1.0
    gi, gda = gi * mask, gda * mask
    torch.set_num_threads(min(threads, max(1, (os.cpu_count() or 2) // 2)))
    r = _torch_run(g, cam, bg, D, gi=gi, gda=gda, cam_grad=False, tile_window=win)
    # beside_fp32: the same independent restatement evaluated in float32 -- how far ANY fp32 evaluation of the algorithm is from
    # float64 in this state (hard gates taken the other way: everything behind them on that pixel moves)
    r32 = _torch_run(g, cam, bg, D, dt=torch.float32, gi=gi, gda=gda, cam_grad=False, tile_window=win) if beside_fp32 else None
    t = {k: torch.tensor(v, device=DEV) for k, v in g.items()}
    out, st = _forward(t, cam, bg, D, want_keys=False)
    o = R.rasterize_backward_raw(st, torch.tensor(gi, device=DEV), torch.tensor(gda, device=DEV))
    torch.cuda.synchronize()
    dr = np.abs(out["radii"].cpu().numpy().astype(np.int64) - r["radii"].astype(np.int64))
    assert dr.max() <= 1 and int((dr > 0).sum()) <= radii_allow, (int(dr.max()), int((dr > 0).sum()))
    nc = out["n_contrib"].cpu().numpy().view(np.uint32)[y0:y1, x0:x1]
    assert int(nc.max()) > min_depth, int(nc.max())                                # the window IS deep
    assert int((nc != r["aux"]["n_contrib"][y0:y1, x0:x1]).sum()) <= 8
    d_img = np.abs(out["color"].cpu().numpy().astype(np.float64) - r["img"]).max(axis=0)[y0:y1, x0:x1]
    n_img = int((d_img > 1e-5).sum())
    print(f"[{label} vs float64 autograd] image: {n_img} pixels beyond 1e-5 (max {d_img.max():.1e}), "
          f"{float(nc.mean()):.0f} splats blended per pixel")
    assert n_img <= flips_allow and float(d_img.max()) <= 4e-3, (n_img, float(d_img.max()))
    d_da = np.abs(out["depth_alpha"].cpu().numpy().astype(np.float64) - r["da"]).max(axis=0)[y0:y1, x0:x1]
    sc_da = rel_scale(r["da"])
    assert int((d_da > 1e-5 * sc_da).sum()) <= flips_allow and float(d_da.max()) <= 4e-3 * sc_da
    for tk, hk in (("means3D", "dL_dmeans3D"), ("scales", "dL_dscales"), ("rotations", "dL_drotations"),
                   ("opacities", "dL_dopacities"), ("shs", "dL_dshs"), ("means2D", "dL_dmeans2D")):
        ref = np.asarray(r["grads"][tk], dtype=np.float64)
        e = np.abs(o[hk].cpu().numpy().astype(np.float64).reshape(ref.shape) - ref)
        sc = rel_scale(ref)
        n_over = int((e > 1e-5 * sc).sum())
        if r32 is None:
            print(f"[{label} vs float64 autograd] {hk}: max {e.max() / sc:.1e}, {n_over} entries beyond 1e-5 (scale {sc:.2e})")
            assert n_over <= flips_allow and float(e.max()) <= 1e-4 * sc, f"{hk}: {n_over} entries beyond 1e-5, max {e.max() / sc:.2e}"
            continue
        e32 = np.abs(np.asarray(r32["grads"][tk], dtype=np.float64).reshape(ref.shape) - ref)
        n32 = int((e32 > 1e-5 * sc).sum())
        n_nz = int((ref != 0).sum())
        print(f"[{label} vs float64 autograd] {hk}: max {e.max() / sc:.1e}, {n_over} of {n_nz} non-zero entries beyond 1e-5 (scale "
              f"{sc:.2e}); the float32 evaluation of the independent restatement: max {e32.max() / sc:.1e}, {n32} entries")
        # no further from float64 than an independent float32 evaluation of the same algorithm is, and nowhere by more than the
        # alpha_min step of a gate taken the other way
        assert n_over <= 1.5 * n32 + 8 and float(e.max()) <= max(2.0 * float(e32.max()), 1e-4 * sc) and float(e.max()) <= 4e-3 * sc, \
            f"{hk}: {n_over} entries beyond 1e-5 (float32 restatement: {n32}), max {e.max() / sc:.2e} ({e32.max() / sc:.2e})"
        assert n_over <= 0.005 * n_nz, f"{hk}: {n_over} of {n_nz} non-zero entries beyond 1e-5"


def test_c3_window_vs_the_independent_float64_autograd_oracle(built_lib):
    """BASELINE.json configs[2] -- the metric's own configuration, 500 k Gaussians @1024^2 -- against the INDEPENDENT float64
    restatement (oracle/torch_oracle.py: libm exp, autograd backward; nothing in common with the kernels but the algorithm), on a
    CROP: the full scene and camera, the loss restricted to the 128 x 128 pixels of the 8 x 8 tiles at the image centre (upstream
    gradients zero elsewhere), the oracle compositing only those tiles (`tile_window`; its per-tile autograd tensors are 90 GB
    for the whole image, 17 GB and 35 s for the window). C3's statistics where they are deepest -- lists of ~4 000 entries, ~600
    blended per pixel -- instead of C2's. (the independent evidence stopped at 100 k @512^2.) Same allowance
    as at C2: float64 takes a hard gate the other way on about one pixel per 10^5; <= 4 pixels / entries per tensor beyond 1e-5,
    none beyond 1e-4 (gradients) / 4e-3 (the alpha_min T step of a flipped gate)."""
    _window_vs_float64(500_000, 1024, (28, 36, 28, 36), False, radii_allow=40, min_depth=300, label="C3 window", threads=32)


def test_initial_state_window_vs_the_independent_float64_autograd_oracle(built_lib):
    """The state the reference's training STARTS in -- every opacity 0.1 (gs_renderer.py:598) -- at C2's size, same crop and same
    allowances: nothing saturates early there, a pixel of the window blends ~2 200 splats (up to ~3 000) where the trained state
    blends a few hundred, so fp32 compositing and K7's sums are at their longest. (the init state is where the kernels
    spend their time -- K7 0.13 of the roofline -- and had been checked against the co-defined C oracle only.) The window takes
    3.6e7 hard gate decisions (16 384 pixels x ~2 200 splats) where C3's takes ~1e7: float64 takes 7 of them the other way in
    the IMAGE (measured; each shows as one pixel with a T step of up to 2e-3; <= 16 allowed) -- and behind every such gate ~2 000
    splats of that pixel see a different T, so the count of gradient entries beyond 1e-5 of their tensor's scale is no longer a
    handful: 43 ... 544 per tensor for the scalar C oracle (whose arithmetic the kernels share), 57 ... 725 for the SAME independent
    restatement evaluated in float32 by torch, max 5e-4 ... 3e-3 for both (CPU run, round 6). A property of the algorithm's hard
    gates in fp32 at this depth, not of a kernel. What is asserted for the gradients here: the HIP path is no further from float64
    than that independent float32 evaluation is (entries beyond 1e-5: <= 1.5 x its count + 8; max: <= 2 x its max), nowhere beyond
    the 4e-3 of a flipped gate, and at most 0.5 % of a tensor's non-zero entries are affected at all (measured on the GPU: 0.03 ...
    0.24 %; 189 / 212 / 65 / 43 / 544 / 180 entries against 235 / 260 / 82 / 56 / 716 / 226 for the float32 restatement)."""
    _window_vs_float64(100_000, 512, (12, 20, 12, 20), True, radii_allow=8, min_depth=1500, label="C2 init-state window", threads=32,
                       flips_allow=16, beside_fp32=True)


def test_c3_steps_with_changing_cameras_leave_the_same_arena_as_full_clears(built_lib):
    """GsrGrads.zero_outside at BASELINE.json configs[2]'s size (500 k Gaussians @1024^2: the 1 024-Gaussians-per-workgroup form of
    K8, sixteen 64-row chunks per workgroup, gridDim.x * 64 apart): six 4-view steps with a DIFFERENT camera set each -- through the
    eager batched module and through graph.CapturedViews, into one arena each -- against the same step into a poisoned arena that
    has to be cleared in full. Same bits, bitmap included."""
    from dreamscene_amd import multiview, synth
    from dreamscene_amd.graph import CapturedViews
    from dreamscene_amd.rasterizer import RasterContext
    from dreamscene_amd.views import GaussianRasterizerViews
    P, K, D, res, V = 500_000, 16, 3, 1024, 4
    dev = torch.device(DEV)
    g = synth.g_object(P, seed=0, K=K)
    cams = synth.object_cameras(12, res, res)
    bg = np.ones(3, np.float32)
    params = {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in g.items()}
    gi, gda = (torch.tensor(x, device=DEV) for x in synth.upstream_grads(res, res, 0))
    arena_e, arena_c = multiview.GradArena(P, K, dev), multiview.GradArena(P, K, dev)
    cap = CapturedViews(context=RasterContext(grad_arena=arena_c))

    def run(rast_call, arena):
        m2d = torch.zeros((V, P, 3), device=DEV, requires_grad=True)
        outs = rast_call(m2d)
        (g2d,) = torch.autograd.grad([x for (img, _, da) in outs for x in (img, da)], [m2d], [gi, gda] * V)
        torch.cuda.synchronize()
        return g2d.clone(), arena.flat.clone(), arena.reached.clone()
    kw = dict(means3D=params["means3D"], opacities=params["opacities"], shs=params["shs"], scales=params["scales"],
              rotations=params["rotations"])
    trusted = 0
    warm = multiview.GradArena(P, K, dev)      # (the first call at a size learns the pair counts one view at a time: another path)
    sl0 = [settings_for(cams[j], bg, D, DEV) for j in range(V)]
    run(lambda m: GaussianRasterizerViews(sl0, context=RasterContext(grad_arena=warm))(means2D=m, **kw), warm)

    def same(got, ref, what, exact):
        # a stale row shows as a non-zero where the reference has a zero; the captured step may cut K7's work into other segments
        # than the eager one (last-bit differences in the sums), so only the eager pair is compared bit for bit
        assert torch.equal(got != 0, ref != 0), f"{what}: zero pattern"
        if exact:
            assert torch.equal(got, ref), what
        else:
            sc = float(ref.abs().max())
            assert float((got - ref).abs().max()) <= 1e-6 * sc, what
    for step, first in enumerate([0, 0, 0, 4, 8, 2, 6, 0]):
        sl = [settings_for(cams[(first + j) % 12], bg, D, DEV) for j in range(V)]
        fresh = multiview.GradArena(P, K, dev)
        fresh.flat.fill_(9.0)
        ref = run(lambda m: GaussianRasterizerViews(sl, context=RasterContext(grad_arena=fresh))(means2D=m, **kw), fresh)
        trusted += int(arena_e.zero_outside_ok())
        got_e = run(lambda m: GaussianRasterizerViews(sl, context=RasterContext(grad_arena=arena_e))(means2D=m, **kw), arena_e)
        got_c = run(lambda m: cap(sl, means2D=m, **kw), arena_c)
        for name, got in (("eager", got_e), ("captured", got_c)):
            same(got[0], ref[0], f"step {step} ({name}): dL_dmeans2D", name == "eager")
            same(got[1], ref[1], f"step {step} ({name}): arena", name == "eager")
            assert torch.equal(got[2], ref[2]), f"step {step} ({name}): reached bitmap"
        # the invariant itself, the slow way: every row outside the bitmap is zero
        assert arena_e.verify_zero_outside() and arena_c.verify_zero_outside() and fresh.verify_zero_outside(), step
    assert trusted == 8 and cap.stats["replays"] >= 4 and cap.stats.get("bwd_zero_outside") == 3, (trusted, cap.stats)
