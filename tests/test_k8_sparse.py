"""-m gpu: K8's sparse form (csrc/preprocess.hip, k_preprocess_bwd_views<K, PVS, REACHED>). A workgroup of 256
Gaussians of which K7 reached at most 128 compacts them and runs the chain rule on those only; the others get zeros
from coalesced clears. The scene below puts all three cases side by side -- workgroups nothing reached (opacity below
1/255: visible, never blended), workgroups mostly reached (they stay dense) and mixed ones -- and the gradients are
compared with the C oracle for single-view calls (write, then accumulate) and for one batched 4-view call."""
import numpy as np
import pytest
import torch

from tests.util import err, oracle_view, rel_scale, settings_for, small_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5
P, H, W, K, D = 2048, 80, 96, 16, 3


def _scene():
    g, _ = small_scene(P=P, H=H, W=W, K=K, seed=29, scale_mul=2.5)
    op = g["opacities"].reshape(-1)
    op[:768] = np.clip(op[:768], 0.05, 0.6)        # thin, translucent cloud: most of it is reached (dense workgroups)
    op[768:1024:2] = 0.002                         # every other one of this workgroup can never pass alpha >= 1/255
    op[1024:1536] = 0.003                          # two workgroups nothing reaches
    op[1536:] = np.where(np.arange(512) % 5 == 0, op[1536:], 0.001)   # 20 % candidates: sparse workgroups
    g["opacities"] = op.reshape(g["opacities"].shape).astype(np.float32)
    return g


def _oracle_sum(c_oracle, g, cams, bg, ups):
    ref = {k: 0.0 for k in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dshs")}
    m2, reached = [], np.zeros(P, bool)
    for cam, (gi, gda) in zip(cams, ups):
        v = oracle_view(c_oracle, cam, P, K, D, bg)
        f = c_oracle.forward(v, g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
        b = c_oracle.backward(v, f, gi, gda, g["means3D"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
        for k in ref:
            ref[k] = ref[k] + np.asarray(b[k], dtype=np.float64)
        m2.append(np.asarray(b["dL_dmeans2D"]))
        reached |= (np.abs(np.asarray(b["dL_dopacity"]).reshape(P, -1)).sum(1) +
                    np.abs(np.asarray(b["dL_dshs"]).reshape(P, -1)).sum(1)) > 0
    return ref, m2, reached


def _compare(arena, ref):
    for ak, rk in [("means3D", "dL_dmeans3D"), ("scales", "dL_dscales"), ("rotations", "dL_drotations"),
                   ("opacities", "dL_dopacity"), ("shs", "dL_dshs")]:
        a = arena.views[ak].cpu().numpy().reshape(-1)
        r = np.asarray(ref[rk]).reshape(-1)
        assert err(a, r) <= TOL * rel_scale(r), ak


def test_sparse_and_dense_workgroups_vs_oracle(built_lib, c_oracle):
    from dreamscene_amd import multiview, rasterizer as R, synth
    from dreamscene_amd.views import GaussianRasterizerViews
    g = _scene()
    cams = synth.object_cameras(5, H, W, radius=3.0)[1:]
    bg = np.array([0.1, 0.3, 0.9], np.float32)
    ups = [synth.upstream_grads(H, W, seed=k) for k in range(4)]
    ref, ref_m2, reached = _oracle_sum(c_oracle, g, cams, bg, ups)
    per_wg = reached.reshape(-1, 256).sum(1)
    assert per_wg.min() == 0 and per_wg.max() > 128 and ((per_wg > 0) & (per_wg <= 128)).any(), per_wg   # all three cases
    t = {k: torch.tensor(v, device=DEV) for k, v in g.items()}
    dev = torch.device(DEV)
    # (1) one call per view: the first writes the arena, the others accumulate into it
    arena = multiview.GradArena(P, K, dev)
    arena.flat.fill_(7.0)                          # stale contents: the first call must overwrite every row
    for j, cam in enumerate(cams):
        s = settings_for(cam, bg, D, DEV)
        _, st = R.rasterize_forward_raw(s, t["means3D"], t["opacities"], t["shs"], None, t["scales"], t["rotations"], None,
                                        want_aux=False)
        o = R.rasterize_backward_raw(st, torch.tensor(ups[j][0], device=DEV), torch.tensor(ups[j][1], device=DEV),
                                     arena=arena, accumulate=j > 0)
        a, r = o["dL_dmeans2D"].cpu().numpy().reshape(-1), ref_m2[j].reshape(-1)
        assert err(a, r) <= TOL * rel_scale(r), ("dL_dmeans2D", j)
    torch.cuda.synchronize()
    _compare(arena, ref)
    # (2) the same four views through one batched call
    arena2 = multiview.GradArena(P, K, dev)
    arena2.flat.fill_(-3.0)
    sets = [settings_for(c, bg, D, dev) for c in cams]
    tt = {k: v.clone().requires_grad_(True) for k, v in t.items()}
    rast = GaussianRasterizerViews(sets, context=R.RasterContext(grad_arena=arena2))
    m2d = torch.zeros((4, P, 3), device=dev, requires_grad=True)
    outs = rast(means3D=tt["means3D"], means2D=m2d, shs=tt["shs"], opacities=tt["opacities"], scales=tt["scales"],
                rotations=tt["rotations"])
    grads = torch.autograd.grad([x for (img, _, da) in outs for x in (img, da)], [m2d],
                                [torch.tensor(y, device=dev) for k in range(4) for y in ups[k]])
    torch.cuda.synchronize()
    _compare(arena2, ref)
    for j in range(4):
        a, r = grads[0][j].cpu().numpy().reshape(-1), ref_m2[j].reshape(-1)
        assert err(a, r) <= TOL * rel_scale(r), ("batched dL_dmeans2D", j)


def _step_sets(n_sets=3):
    """Camera sets whose reached rows differ (the previous set's rows must be cleared, not all of them)."""
    from dreamscene_amd import synth
    cams = synth.object_cameras(4 * n_sets + 1, H, W, radius=3.0)[1:]
    return [cams[4 * s:4 * s + 4] for s in range(n_sets)]


def _eager_step(rastmod, R, sets, t, ups, dev, arena, per_view_scales=False):
    """One 4-view step through GaussianRasterizerViews into `arena` (None: plain gradients) -> (param grads, means2D grads).
    per_view_scales: scales [V,P,3] (the trainers' scale noise): the scale gradients then go to a [V,P,3] tensor of their own and
    the arena's `scales` region is NOT written."""
    tt = {k: v.clone().requires_grad_(True) for k, v in t.items()}
    if per_view_scales:
        tt["scales"] = (t["scales"].unsqueeze(0) * torch.tensor([1.0, 1.01, 0.99, 1.02], device=dev).view(4, 1, 1)).requires_grad_(True)
    rast = rastmod(sets, context=R.RasterContext(grad_arena=arena))
    m2d = torch.zeros((4, P, 3), device=dev, requires_grad=True)
    outs = rast(means3D=tt["means3D"], means2D=m2d, shs=tt["shs"], opacities=tt["opacities"], scales=tt["scales"],
                rotations=tt["rotations"])
    leaves = [m2d] if arena is not None else [m2d] + [tt[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    gr = torch.autograd.grad([x for (img, _, da) in outs for x in (img, da)], leaves,
                             [torch.tensor(y, device=dev) for k in range(4) for y in ups[k]])
    torch.cuda.synchronize()
    return gr


def test_rows_known_to_be_zero_are_not_written_again(built_lib):
    """GsrGrads.zero_outside: a backward that overwrites an arena whose rows outside the reached bitmap are known to be zero clears
    only the rows the bitmap names. Same bits as the backward into a fresh arena, step after step with CHANGING cameras; a torch op
    on the arena (version counter) or touch() brings the full clear back."""
    from dreamscene_amd import multiview, rasterizer as R
    from dreamscene_amd.views import GaussianRasterizerViews
    from dreamscene_amd import synth
    g = _scene()
    bg = np.array([0.1, 0.3, 0.9], np.float32)
    dev = torch.device(DEV)
    ups = [synth.upstream_grads(H, W, seed=k) for k in range(4)]
    t = {k: torch.tensor(v, device=DEV) for k, v in g.items()}
    camsets = _step_sets(3)
    arena = multiview.GradArena(P, K, dev)
    masks = []
    for step, cs in enumerate(camsets + camsets[:1]):
        sets = [settings_for(c, bg, D, dev) for c in cs]
        assert arena.zero_outside_ok() == (step != 2), step
        m2 = _eager_step(GaussianRasterizerViews, R, sets, t, ups, dev, arena)
        fresh = multiview.GradArena(P, K, dev)
        fresh.flat.fill_(5.0)                    # (a torch write: the reference run clears everything)
        assert not fresh.zero_outside_ok()
        m2_ref = _eager_step(GaussianRasterizerViews, R, sets, t, ups, dev, fresh)
        assert torch.equal(arena.flat, fresh.flat), f"step {step}: arena differs from the fully cleared one"
        assert torch.equal(arena.reached, fresh.reached), step
        assert torch.equal(m2[0], m2_ref[0]), step
        masks.append(arena.reached.clone())
        if step == 1:
            arena.views["shs"].mul_(2.0)         # a torch op on a view of the arena: the next backward must not trust the bitmap
    assert not torch.equal(masks[0], masks[1]), "the camera sets were meant to reach different rows"
    arena.touch()
    assert not arena.zero_outside_ok()


def test_a_step_with_per_view_scales_leaves_the_scales_region_alone(built_lib):
    """bench.py's parity block caught this one (round 6): a step with per-view scales replaces the arena's bitmap WITHOUT writing
    its `scales` region; the next ordinary step must not trust the bitmap for that region -- GradArena.zero_outside_ok(regions)."""
    from dreamscene_amd import multiview, rasterizer as R, synth
    from dreamscene_amd.views import GaussianRasterizerViews
    g = _scene()
    bg = np.array([0.1, 0.3, 0.9], np.float32)
    dev = torch.device(DEV)
    ups = [synth.upstream_grads(H, W, seed=k) for k in range(4)]
    t = {k: torch.tensor(v, device=DEV) for k, v in g.items()}
    camsets = _step_sets(3)
    arena = multiview.GradArena(P, K, dev)
    for step, (si, pvs) in enumerate([(0, False), (1, True), (2, False), (0, True), (1, True), (2, False)]):
        sets = [settings_for(c, bg, D, dev) for c in camsets[si]]
        _eager_step(GaussianRasterizerViews, R, sets, t, ups, dev, arena, per_view_scales=pvs)
        fresh = multiview.GradArena(P, K, dev)
        fresh.flat.fill_(2.0)
        _eager_step(GaussianRasterizerViews, R, sets, t, ups, dev, fresh, per_view_scales=pvs)
        for name in ("means3D", "rotations", "opacities", "shs") + (() if pvs else ("scales",)):
            assert torch.equal(arena.views[name], fresh.views[name]), f"step {step} (per-view scales: {pvs}): {name}"
        assert arena.zero_outside_ok(("means3D", "shs")) and arena.zero_outside_ok() == (not pvs)
        assert arena.verify_zero_outside()              # (the regions the last backward wrote: the slow check of the invariant)


@pytest.mark.parametrize("with_arena", [True, False])
def test_captured_backward_keeps_its_results_sparse(built_lib, with_arena):
    """graph.CapturedViews owns its gradient tensors: graph C's K8 is captured with zero_outside and every replay is checked
    against the eager module on a fresh arena -- with changing cameras, a foreign write into the arena between two replays, an
    in-place edit of a returned gradient, and another module writing the same arena in between."""
    from dreamscene_amd import graph, multiview, rasterizer as R, synth
    from dreamscene_amd.views import GaussianRasterizerViews
    g = _scene()
    bg = np.array([0.1, 0.3, 0.9], np.float32)
    dev = torch.device(DEV)
    ups = [synth.upstream_grads(H, W, seed=k) for k in range(4)]
    up_t = [torch.tensor(y, device=dev) for k in range(4) for y in ups[k]]
    t = {k: torch.tensor(v, device=DEV) for k, v in g.items()}
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    leaves = {k: t[k].clone().requires_grad_(True) for k in names}
    arena = multiview.GradArena(P, K, dev) if with_arena else None
    rast = graph.CapturedViews(context=R.RasterContext(grad_arena=arena))
    camsets = _step_sets(3)
    order = [0, 0, 0, 1, 2, 0, 1, 1, 2, 0]        # (the first calls are eager warm-ups, then the capture)
    for it, si in enumerate(order):
        sets = [settings_for(c, bg, D, dev) for c in camsets[si]]
        m2d = torch.zeros((4, P, 3), device=dev, requires_grad=True)
        outs = rast(sets, means3D=leaves["means3D"], means2D=m2d, shs=leaves["shs"], opacities=leaves["opacities"],
                    scales=leaves["scales"], rotations=leaves["rotations"])
        flat = [x for (img, _, da) in outs for x in (img, da)]
        if with_arena:
            got = torch.autograd.grad(flat, [m2d], up_t)
            got_params = arena.flat.clone()
        else:
            got = torch.autograd.grad(flat, [m2d] + [leaves[k] for k in names], up_t)
        torch.cuda.synchronize()
        fresh = multiview.GradArena(P, K, dev)
        fresh.flat.fill_(-1.0)
        ref_m2 = _eager_step(GaussianRasterizerViews, R, sets, t, ups, dev, fresh)
        assert torch.equal(got[0], ref_m2[0]), f"call {it}: dL_dmeans2D"
        if with_arena:
            assert torch.equal(got_params, fresh.flat), f"call {it}: arena"
        else:
            for k, gk in zip(names, got[1:]):
                assert torch.equal(gk.reshape(-1), fresh.views[k].reshape(-1)), f"call {it}: dL_d{k}"
        # what a caller may do between two steps
        if it == 5:
            if with_arena:
                arena.flat.add_(3.0)                         # a torch write into the arena
            else:
                got[1].add_(3.0)                             # an in-place edit of a returned gradient (the capture's own tensor)
        if it == 6:
            got[0].mul_(2.0)                                 # ... of the per-view rows
        if it == 7 and with_arena:
            _eager_step(GaussianRasterizerViews, R, [settings_for(c, bg, D, dev) for c in camsets[0]], t, ups, dev, arena)
    st = rast.stats
    assert st["replays"] >= 5, st


@pytest.mark.parametrize("captured", [False, True])
def test_one_view_with_per_view_scales_into_a_trusted_arena(built_lib, captured):
    """tools/fuzz_views.py seeds 90 / 160 (round 6): ONE view whose scales come as [1,P,3]. The library sees an ordinary single view --
    dL_dscales is then one of the summed outputs GsrGrads.zero_outside bit 0 speaks for -- while the wrapper hands it a fresh
    [1,P,3] tensor of its own and trusted the ARENA: rows nothing reached kept whatever the allocator's block held."""
    from dreamscene_amd import graph, multiview, rasterizer as R, synth
    from dreamscene_amd.views import GaussianRasterizerViews
    g = _scene()
    bg = np.array([0.1, 0.3, 0.9], np.float32)
    dev = torch.device(DEV)
    gi, gda = (torch.tensor(x, device=dev) for x in synth.upstream_grads(H, W, seed=0))
    t = {k: torch.tensor(v, device=DEV) for k, v in g.items()}
    cams = [cs[0] for cs in _step_sets(3)]
    names = ("means3D", "shs", "opacities", "rotations")
    leaves = {k: t[k].clone().requires_grad_(True) for k in names}
    arena = multiview.GradArena(P, K, dev)
    rast_c = graph.CapturedViews(context=R.RasterContext(grad_arena=arena)) if captured else None
    for it, ci in enumerate([0, 0, 0, 1, 2, 0, 1]):
        sets = [settings_for(cams[ci], bg, D, dev)]
        sc = (t["scales"].unsqueeze(0) * 1.01).requires_grad_(True)
        poison = torch.full((4 * P * 3 + 64,), float("nan"), device=dev)      # what the next torch.empty of that size will hold
        del poison
        m2d = torch.zeros((1, P, 3), device=dev, requires_grad=True)
        if captured:
            outs = rast_c(sets, means3D=leaves["means3D"], means2D=m2d, shs=leaves["shs"], opacities=leaves["opacities"], scales=sc,
                          rotations=leaves["rotations"])
        else:
            outs = GaussianRasterizerViews(sets, context=R.RasterContext(grad_arena=arena))(
                means3D=leaves["means3D"], means2D=m2d, shs=leaves["shs"], opacities=leaves["opacities"], scales=sc,
                rotations=leaves["rotations"])
        g_m2d, g_sc = torch.autograd.grad([outs[0][0], outs[0][2]], [m2d, sc], [gi, gda])
        got_sc, got_m2d, got_arena = g_sc.clone(), g_m2d.clone(), arena.flat.clone()
        # the same view with nothing to trust: no arena, fresh tensors
        ref_leaves = {k: t[k].clone().requires_grad_(True) for k in names}
        sc2 = (t["scales"].unsqueeze(0) * 1.01).requires_grad_(True)
        m2 = torch.zeros((1, P, 3), device=dev, requires_grad=True)
        o2 = GaussianRasterizerViews(sets)(means3D=ref_leaves["means3D"], means2D=m2, shs=ref_leaves["shs"],
                                           opacities=ref_leaves["opacities"], scales=sc2, rotations=ref_leaves["rotations"])
        r_m2, r_sc, r_mean = torch.autograd.grad([o2[0][0], o2[0][2]], [m2, sc2, ref_leaves["means3D"]], [gi, gda])
        torch.cuda.synchronize()
        assert torch.isfinite(got_sc).all(), f"call {it}: uninitialised rows in dL/dscales"
        assert torch.equal(got_sc, r_sc) and torch.equal(got_m2d, r_m2), f"call {it}"
        assert torch.equal(arena.views["means3D"], r_mean), f"call {it}: arena"
