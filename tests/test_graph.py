"""-m gpu: the captured (hipGraph) step -- dreamscene_amd/graph.py -- replays exactly what the eager batched path runs:
per-view outputs bit-identical, gradients equal to fp32-atomic order, new cameras / FoV / SH degree / backgrounds picked
up from the packed camera block at replay time, capacity overflow handled by an exact eager step + re-capture."""
import numpy as np
import pytest
import torch

from tests.util import same_bits, settings_for, small_scene, tol_ok

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(P=3000, H=112, W=144, K=16, seed=17, scale_mul=6.0):
    g, _ = small_scene(P=P, H=H, W=W, K=K, seed=seed, scale_mul=scale_mul)
    t = {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in g.items()}
    return g, t


def _eager(sets, t, gis, gdas, scales=None):
    from dreamscene_amd.views import GaussianRasterizerViews
    V, P = len(sets), t["means3D"].shape[0]
    leaves = [t[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    rast = GaussianRasterizerViews(sets)
    for _ in range(2):      # second call: batched
        m2d = torch.zeros((V, P, 3), device=DEV, requires_grad=True)
        outs = rast(means3D=t["means3D"], means2D=m2d, shs=t["shs"], opacities=t["opacities"],
                    scales=t["scales"] if scales is None else scales, rotations=t["rotations"])
        grads = torch.autograd.grad([x for (img, _, da) in outs for x in (img, da)], leaves + [m2d],
                                    [y for k in range(V) for y in (gis[k], gdas[k])])
    return [(a.clone(), b.clone(), c.clone()) for a, b, c in outs], [x.clone() for x in grads]


@pytest.mark.parametrize("V,K,D,direct", [(4, 16, 3, True), (1, 4, 1, True), (3, 9, 2, False), (4, 16, 3, False)])
def test_captured_step_equals_eager_and_follows_the_cameras(built_lib, monkeypatch, V, K, D, direct):
    """direct: the backward graph reads the caller's gradient tensors (their addresses repeat); not direct: the gradients
    are copied into the module's static buffers first (what happens when the addresses keep changing)."""
    from dreamscene_amd import graph, synth
    from dreamscene_amd.graph import CapturedViews
    if not direct:
        monkeypatch.setattr(graph, "MAX_DIRECT_GRAPHS", 0)
    P, H, W = 3000, 112, 144
    g, t = _setup(P, H, W, K)
    leaves = [t[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    cams = synth.object_cameras(8, H, W, radius=3.0)
    gis = [torch.tensor(synth.upstream_grads(H, W, seed=k)[0], device=DEV) for k in range(V)]
    gdas = [torch.tensor(synth.upstream_grads(H, W, seed=k)[1], device=DEV) for k in range(V)]
    rast = CapturedViews()
    # five steps with DIFFERENT cameras, backgrounds, FoV and active SH degree: steps 0-1 run eagerly (they learn the pair
    # counts), step 2 captures, steps 3-4 replay with new cameras
    for step in range(5):
        sets = []
        for k in range(V):
            c = cams[(step + 2 * k) % 8]
            s = settings_for(c, [0.1 * step, 0.4, 1.0 - 0.2 * k], D if (step + k) % 3 else 0, DEV)
            if step == 4:                        # a different field of view at replay time (GsrView.dynamic)
                s = s._replace(tanfovx=s.tanfovx * 1.25, tanfovy=s.tanfovy * 1.25)
            sets.append(s)
        ref_outs, ref_grads = _eager(sets, t, gis, gdas)
        m2d = torch.zeros((V, P, 3), device=DEV, requires_grad=True)
        outs = rast(sets, means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], scales=t["scales"],
                    rotations=t["rotations"])
        grads = torch.autograd.grad([x for (img, _, da) in outs for x in (img, da)], leaves + [m2d],
                                    [y for k in range(V) for y in (gis[k], gdas[k])])
        for (img, radii, da), (rimg, rradii, rda) in zip(outs, ref_outs):
            assert torch.equal(radii, rradii), step
            assert torch.equal(img, rimg) and torch.equal(da, rda), step      # same kernels, same order inside a view
        for a, b in zip(grads, ref_grads):
            assert tol_ok(a.reshape(b.shape).cpu().numpy(), b.cpu().numpy(), atol=2e-6), step   # (fp32 atomics order)
    assert rast.stats["captures"] == 1 and rast.stats["replays"] == 3 and rast.stats["eager_steps"] == 2, rast.stats


def test_captured_step_with_arena_per_view_scales_and_stats(built_lib):
    """The trainers' configuration: gradients delivered in a GradArena, fresh per-view scale noise ([V,P,3] scales, their
    own gradient per view), densification statistics of the last view -- all through the replayed graphs."""
    from dreamscene_amd import densify, multiview, synth
    from dreamscene_amd.graph import CapturedViews
    from dreamscene_amd.rasterizer import RasterContext
    from dreamscene_amd.views import GaussianRasterizerViews
    V, P, H, W, K, D = 4, 2500, 96, 128, 16, 3
    g, t = _setup(P, H, W, K, seed=23)
    cams = synth.object_cameras(8, H, W, radius=3.0)
    gis = [torch.tensor(synth.upstream_grads(H, W, seed=k)[0], device=DEV) for k in range(V)]
    gdas = [torch.tensor(synth.upstream_grads(H, W, seed=k)[1], device=DEV) for k in range(V)]
    gen = torch.Generator().manual_seed(5)
    arena_c, arena_e = multiview.GradArena(P, K, torch.device(DEV)), multiview.GradArena(P, K, torch.device(DEV))
    stats_c, stats_e = densify.DensifyStats(P, torch.device(DEV)), densify.DensifyStats(P, torch.device(DEV))
    rc_c = RasterContext(grad_arena=arena_c, densify_stats=stats_c.tensors())
    rc_e = RasterContext(grad_arena=arena_e, densify_stats=stats_e.tensors())
    rast_c = CapturedViews(context=rc_c)
    for step in range(5):
        sets = [settings_for(cams[(step + k) % 8], [1, 1, 1], D, DEV) for k in range(V)]
        noise = torch.randn((V, P, 3), generator=gen).to(DEV)
        sc = torch.clamp(t["scales"][None] + noise * ((0.2 ** 0.5) * t["scales"][None] / 4), 0.0).detach().requires_grad_(True)
        res = {}
        for name, rast, arena in (("e", GaussianRasterizerViews(sets, context=rc_e), arena_e), ("c", rast_c, arena_c)):
            m2d = torch.zeros((V, P, 3), device=DEV, requires_grad=True)
            kw = dict(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], scales=sc,
                      rotations=t["rotations"])
            outs = rast(sets, **kw) if name == "c" else rast(**kw)
            g2d, gsc = torch.autograd.grad([x for (img, _, da) in outs for x in (img, da)], [m2d, sc],
                                           [y for k in range(V) for y in (gis[k], gdas[k])])
            res[name] = ([tuple(x.clone() for x in o) for o in outs], g2d.clone(), gsc.clone(), arena.flat.clone())
        for (a, b) in zip(res["c"][0], res["e"][0]):
            assert all(torch.equal(x, y) for x, y in zip(a, b)), step
        assert tol_ok(res["c"][1].cpu().numpy(), res["e"][1].cpu().numpy(), atol=2e-6)
        assert tol_ok(res["c"][2].cpu().numpy(), res["e"][2].cpu().numpy(), atol=2e-6)
        assert tol_ok(res["c"][3].cpu().numpy(), res["e"][3].cpu().numpy(), atol=2e-6)
    assert torch.equal(stats_c.denom, stats_e.denom) and torch.equal(stats_c.max_radii2D, stats_e.max_radii2D)
    np.testing.assert_allclose(stats_c.xyz_gradient_accum.cpu().numpy(), stats_e.xyz_gradient_accum.cpu().numpy(),
                               rtol=1e-5, atol=1e-9)
    assert int(stats_c.denom.max()) == 5 and rast_c.stats["captures"] == 1


def test_captured_step_overflow_falls_back_and_recaptures(built_lib):
    """The splats grow until a view's pair count exceeds the captured capacity: that step is redone eagerly (exact), the
    next one captures again with more room; results stay equal to the eager path throughout."""
    from dreamscene_amd import synth
    from dreamscene_amd.graph import CapturedViews
    V, P, H, W, K, D = 2, 2000, 192, 256, 4, 1          # 192 tiles: 2000 screen-filling splats exceed the 131072-pair capacity
    g, t = _setup(P, H, W, K, seed=41, scale_mul=1.0)
    cams = synth.object_cameras(4, H, W, radius=3.0)
    gis = [torch.tensor(synth.upstream_grads(H, W, seed=k)[0], device=DEV) for k in range(V)]
    gdas = [torch.tensor(synth.upstream_grads(H, W, seed=k)[1], device=DEV) for k in range(V)]
    sets = [settings_for(cams[k + 1], [0, 0, 0], D, DEV) for k in range(V)]
    rast = CapturedViews(headroom=1.0)
    rast_cap_before = None
    for step, mul in enumerate([1.0, 1.0, 1.0, 1.0, 12.0, 12.0, 12.0, 12.0]):
        with torch.no_grad():
            t["scales"].copy_(torch.tensor(g["scales"], device=DEV) * mul)
        ref_outs, _ = _eager(sets, t, gis, gdas)
        m2d = torch.zeros((V, P, 3), device=DEV, requires_grad=True)
        outs = rast(sets, means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], scales=t["scales"],
                    rotations=t["rotations"])
        torch.autograd.backward([x for (img, _, da) in outs for x in (img, da)],
                                [y for k in range(V) for y in (gis[k], gdas[k])])
        for (img, radii, da), (rimg, rradii, rda) in zip(outs, ref_outs):
            assert torch.equal(radii, rradii) and torch.equal(img, rimg) and torch.equal(da, rda), step
        if step == 3:
            rast_cap_before = rast._cap.cap
    assert rast.stats["overflows"] == 1 and rast.stats["captures"] == 2, rast.stats
    assert rast._cap.cap > rast_cap_before


def test_fresh_input_tensors_are_staged_not_recaptured(built_lib):
    """The reference's trainers hand over ACTIVATIONS (a new tensor per step: gs_renderer.py:464-488), so a capture keyed
    on input addresses never hits. After PTR_MISSES_TO_STAGE captures lost to new addresses alone the inputs are copied
    into static buffers of the capture and the key is shapes only: no further captures, same results as the eager path."""
    from dreamscene_amd import synth
    from dreamscene_amd.graph import CapturedViews, PTR_MISSES_TO_STAGE, WARM_CALLS
    V, P, H, W, K, D = 2, 2500, 96, 128, 16, 3
    g, raw = _setup(P, H, W, K, seed=29)
    cams = synth.object_cameras(8, H, W, radius=3.0)
    gis = [torch.tensor(synth.upstream_grads(H, W, seed=k)[0], device=DEV) for k in range(V)]
    gdas = [torch.tensor(synth.upstream_grads(H, W, seed=k)[1], device=DEV) for k in range(V)]
    rast = CapturedViews()
    keep = []                   # hold every step's inputs alive: the allocator must not hand the same block back
    steps = WARM_CALLS + PTR_MISSES_TO_STAGE + 4
    for step in range(steps):
        # fresh tensors with values that move from step to step (what an optimizer + activations produce)
        t = {k: (v.detach() * (1.0 + 0.01 * step) if k in ("scales", "opacities") else v.detach() + 0.0).requires_grad_(True)
             for k, v in raw.items()}
        keep.append(t)
        sets = [settings_for(cams[(step + 3 * k) % 8], [1, 1, 1], D, DEV) for k in range(V)]
        ref_outs, ref_grads = _eager(sets, t, gis, gdas)
        leaves = [t[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]
        m2d = torch.zeros((V, P, 3), device=DEV, requires_grad=True)
        outs = rast(sets, means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], scales=t["scales"],
                    rotations=t["rotations"])
        grads = torch.autograd.grad([x for (img, _, da) in outs for x in (img, da)], leaves + [m2d],
                                    [y for k in range(V) for y in (gis[k], gdas[k])])
        for (img, radii, da), (rimg, rradii, rda) in zip(outs, ref_outs):
            assert torch.equal(radii, rradii) and torch.equal(img, rimg) and torch.equal(da, rda), step
        for a, b in zip(grads, ref_grads):
            assert tol_ok(a.reshape(b.shape).cpu().numpy(), b.cpu().numpy(), atol=2e-6), step
    assert rast.stats["staged_inputs"] is True
    assert rast.stats["captures"] == PTR_MISSES_TO_STAGE + 1, rast.stats      # by address, ..., then once on shapes
    assert rast.stats["replays"] == steps - WARM_CALLS, rast.stats


def test_backward_of_an_overwritten_replay_is_refused(built_lib):
    """The static state of a capture belongs to its LATEST replay: backward of an earlier forward must raise, not
    silently differentiate the later step (the eager path keeps per-call state and supports this pattern)."""
    from dreamscene_amd import synth
    from dreamscene_amd.graph import CapturedViews, WARM_CALLS
    V, P, H, W, K, D = 2, 2000, 96, 128, 4, 1
    g, t = _setup(P, H, W, K, seed=31)
    cams = synth.object_cameras(4, H, W, radius=3.0)
    rast = CapturedViews()

    def fwd(step):
        sets = [settings_for(cams[(step + k) % 4], [0, 0, 0], D, DEV) for k in range(V)]
        m2d = torch.zeros((V, P, 3), device=DEV, requires_grad=True)
        return rast(sets, means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], scales=t["scales"],
                    rotations=t["rotations"])
    for step in range(WARM_CALLS + 1):          # warm-up calls + the capturing one, each with its own backward
        sum(img.sum() for img, _, _ in fwd(step)).backward()
    first = fwd(10)
    loss_first = sum(img.sum() for img, _, _ in first)
    second = fwd(11)                            # overwrites the static state `first` refers to
    with pytest.raises(RuntimeError, match="overwritten by a later forward"):
        loss_first.backward()
    sum(img.sum() for img, _, _ in second).backward()       # the latest forward still differentiates
    assert rast.stats["captures"] == 1


def test_an_eval_render_between_forward_and_backward_leaves_the_capture_alone(built_lib):
    """a trainer that renders an eval view (torch.no_grad()) between a step's forward and its backward. While a
    differentiable captured forward waits for its backward, forward-only calls of the same CapturedViews run eagerly: the step's
    backward still differentiates ITS forward (same gradients as without the eval render), the eval render is the eager one's."""
    from dreamscene_amd import synth
    from dreamscene_amd.graph import CapturedViews, WARM_CALLS
    from dreamscene_amd.views import GaussianRasterizerViews
    V, P, H, W, K, D = 2, 2500, 96, 128, 4, 1
    g, t = _setup(P, H, W, K, seed=37)
    cams = synth.object_cameras(6, H, W, radius=3.0)
    rast = CapturedViews()
    leaves = [t[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]

    def sets_of(step):
        return [settings_for(cams[(step + k) % 6], [0.2, 0.3, 0.4], D, DEV) for k in range(V)]

    def fwd(step):
        m2d = torch.zeros((V, P, 3), device=DEV, requires_grad=True)
        return rast(sets_of(step), means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], scales=t["scales"],
                    rotations=t["rotations"])
    for step in range(WARM_CALLS + 2):          # warm-up, capture, one replay -- each with its own backward
        sum(img.sum() + da.sum() for img, _, da in fwd(step)).backward()
    for x in leaves:
        x.grad = None
    sum(img.sum() + da.sum() for img, _, da in fwd(20)).backward()          # reference: step 20 without anything in between
    ref = [x.grad.clone() for x in leaves]
    for x in leaves:
        x.grad = None
    outs = fwd(20)
    loss = sum(img.sum() + da.sum() for img, _, da in outs)
    replays = rast.stats["replays"]
    with torch.no_grad():                        # the eval render: other cameras, forward only
        ev = rast(sets_of(3), means3D=t["means3D"], means2D=torch.zeros((V, P, 3), device=DEV), opacities=t["opacities"],
                  shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
        ev_ref = GaussianRasterizerViews(sets_of(3))(means3D=t["means3D"], means2D=torch.zeros((V, P, 3), device=DEV),
                                                     opacities=t["opacities"], shs=t["shs"], scales=t["scales"],
                                                     rotations=t["rotations"])
    assert rast.stats["replays"] == replays, "the forward-only call replayed the capture whose backward is pending"
    for (a, _, b), (c, _, d) in zip(ev, ev_ref):
        assert torch.equal(a, c) and torch.equal(b, d)
    loss.backward()                              # no "overwritten by a later forward"
    torch.cuda.synchronize()
    for x, r, n in zip(leaves, ref, ("means3D", "shs", "opacities", "scales", "rotations")):
        same_bits(x.grad, r, f"dL/d{n} with an eval render in between")
    with torch.no_grad():                        # nothing pending any more: forward-only calls replay again
        rast(sets_of(4), means3D=t["means3D"], means2D=torch.zeros((V, P, 3), device=DEV), opacities=t["opacities"],
             shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    assert rast.stats["replays"] == replays + 1


def test_staged_inputs_go_back_to_zero_copy_when_addresses_settle(built_lib):
    """`_staged` used to be sticky. A caller that first hands over fresh tensors (staging kicks in) and then
    settles on persistent ones gets the zero-copy capture (keyed on addresses) back after PTR_REPEATS_TO_UNSTAGE calls."""
    from dreamscene_amd import synth
    from dreamscene_amd.graph import CapturedViews, PTR_MISSES_TO_STAGE, PTR_REPEATS_TO_UNSTAGE, WARM_CALLS
    V, P, H, W, K, D = 2, 2000, 96, 128, 4, 1
    g, raw = _setup(P, H, W, K, seed=33)
    cams = synth.object_cameras(4, H, W, radius=3.0)
    sets = [settings_for(cams[k], [1, 1, 1], D, DEV) for k in range(V)]
    gis = [torch.tensor(synth.upstream_grads(H, W, seed=k)[0], device=DEV) for k in range(V)]
    gdas = [torch.tensor(synth.upstream_grads(H, W, seed=k)[1], device=DEV) for k in range(V)]
    rast = CapturedViews()
    keep = []

    def step(t):
        m2d = torch.zeros((V, P, 3), device=DEV, requires_grad=True)
        outs = rast(sets, means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], scales=t["scales"],
                    rotations=t["rotations"])
        leaves = [t[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]
        grads = torch.autograd.grad([x for (img, _, da) in outs for x in (img, da)], leaves + [m2d],
                                    [y for k in range(V) for y in (gis[k], gdas[k])])
        return [tuple(x.clone() for x in o) for o in outs], [x.clone() for x in grads]
    for s_ in range(WARM_CALLS + PTR_MISSES_TO_STAGE + 2):          # fresh tensors every call -> staged
        t = {k: (v.detach() + 0.0).requires_grad_(True) for k, v in raw.items()}
        keep.append(t)
        step(t)
    assert rast.stats["staged_inputs"] is True
    fixed = {k: (v.detach() + 0.0).requires_grad_(True) for k, v in raw.items()}
    ref_outs, ref_grads = _eager(sets, fixed, gis, gdas)
    for s_ in range(PTR_REPEATS_TO_UNSTAGE + 3):                     # the same tensors call after call -> zero-copy again
        outs, grads = step(fixed)
        for (img, radii, da), (rimg, rradii, rda) in zip(outs, ref_outs):
            assert torch.equal(radii, rradii) and torch.equal(img, rimg) and torch.equal(da, rda), s_
        for a, b in zip(grads, ref_grads):
            assert tol_ok(a.reshape(b.shape).cpu().numpy(), b.cpu().numpy(), atol=2e-6), s_
    assert rast.stats["staged_inputs"] is False, rast.stats
