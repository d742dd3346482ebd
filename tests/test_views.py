"""Several views of the same Gaussians through one call (dreamscene_amd/views.py, gsr_forward_project_batch): per-view
results identical to GaussianRasterizer, gradients = the sum over the views."""
import numpy as np
import pytest
import torch

from tests.util import settings_for, small_scene, tol_ok


@pytest.mark.gpu
@pytest.mark.parametrize("K,D,use_arena,noisy_scales,V", [(16, 3, False, False, 4), (4, 1, True, False, 4),
                                                           (16, 3, False, True, 4), (16, 3, False, False, 3),
                                                           (9, 2, True, False, 7)])
def test_batched_views_match_sequential(built_lib, K, D, use_arena, noisy_scales, V):
    """(V = 3, 7: the compositing kernels map workgroup b to item b / V of view b % V -- not only powers of two.)"""
    from dreamscene_amd import multiview, rasterizer as R, synth
    from dreamscene_amd.rasterizer import GaussianRasterizer
    from dreamscene_amd.views import GaussianRasterizerViews
    dev = torch.device("cuda:0")
    P, H, W = 1500, 112, 144
    g, _ = small_scene(P=P, H=H, W=W, K=K, seed=17)
    cams = synth.object_cameras(V + 1, H, W, radius=3.0)[1:]
    # per-view background and active SH degree (scene_render's bg / sh_deg augmentation draws them per view)
    sets = [settings_for(c, [0.2 * k, 0.4, 1.0 - 0.3 * k], D if k != 2 else 0, dev) for k, c in enumerate(cams)]
    t = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in g.items()}
    leaves = [t[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    # the trainers add fresh noise to the activated scales of every view (scene_gaussian.py:1004-1008): [V,P,3] scales
    gen = torch.Generator().manual_seed(3)
    noise = torch.randn((V, P, 3), generator=gen).to(dev)
    view_scales = (lambda: torch.clamp(t["scales"][None] + noise * ((0.2 ** 0.5) * t["scales"][None] / 4), 0.0)) \
        if noisy_scales else None
    gis = [torch.tensor(synth.upstream_grads(H, W, seed=k)[0], device=dev) for k in range(V)]
    gdas = [torch.tensor(synth.upstream_grads(H, W, seed=k)[1], device=dev) for k in range(V)]

    def sequential():
        outs, tot, m2ds = [], None, []
        for k, s in enumerate(sets):
            m2d = torch.zeros((P, 3), device=dev, requires_grad=True)
            sck = view_scales()[k] if noisy_scales else t["scales"]
            img, radii, da = GaussianRasterizer(s)(means3D=t["means3D"], means2D=m2d, shs=t["shs"],
                                                    opacities=t["opacities"], scales=sck, rotations=t["rotations"])
            gr = torch.autograd.grad([img, da], leaves + [m2d], [gis[k], gdas[k]])
            outs.append((img, radii, da))
            m2ds.append(gr[-1].clone())     # (a torch.autograd.grad result of the captured ring aliases its slot: dropin.py)
            tot = [a.clone() for a in gr[:-1]] if tot is None else [a + b for a, b in zip(tot, gr[:-1])]
        return outs, tot, torch.stack(m2ds)

    ref_outs, ref_grads, ref_m2d = sequential()         # also leaves the capacity hint the batched path needs
    arena = multiview.GradArena(P, K, dev) if use_arena else None
    rast = GaussianRasterizerViews(sets, context=R.RasterContext(grad_arena=arena))
    for rep in range(2):
        m2d = torch.zeros((V, P, 3), device=dev, requires_grad=True)
        outs = rast(means3D=t["means3D"], means2D=m2d, shs=t["shs"], opacities=t["opacities"],
                    scales=view_scales() if noisy_scales else t["scales"], rotations=t["rotations"])
        # with an arena the parameter gradients are delivered THERE, not through autograd (allow_unused)
        grads = torch.autograd.grad([x for (img, _, da) in outs for x in (img, da)], leaves + [m2d],
                                    [y for k in range(V) for y in (gis[k], gdas[k])], allow_unused=use_arena)
    for (img, radii, da), (rimg, rradii, rda) in zip(outs, ref_outs):
        assert torch.equal(radii, rradii)
        assert torch.equal(img, rimg) and torch.equal(da, rda)      # same kernels, same order inside a view
    assert tol_ok(grads[-1].cpu().numpy(), ref_m2d.cpu().numpy(), atol=2e-6)    # (fp32 atomics: order varies run to run)
    got = [arena.views[n] for n in ("means3D", "shs", "opacities", "scales", "rotations")] if use_arena else grads[:-1]
    for a, b in zip(got, ref_grads):
        assert tol_ok(a.reshape(b.shape).cpu().numpy(), b.cpu().numpy(), atol=2e-6)   # fp32 summation order over views


@pytest.mark.gpu
def test_batched_views_fall_back_without_hint_or_on_big_grids(built_lib):
    """First call (no capacity hint yet) and mixed image sizes run view by view; results still correct."""
    from dreamscene_amd import rasterizer as R, synth
    from dreamscene_amd.views import rasterize_views_forward_raw
    dev = torch.device("cuda:0")
    P, K, D = 700, 4, 1
    g, _ = small_scene(P=P, H=80, W=80, K=K, seed=23)
    t = {k: torch.tensor(v, device=dev) for k, v in g.items()}
    cams_a = synth.object_cameras(3, 83, 91, radius=3.0)[1:]            # a size nobody rendered before: no hint
    sets = [settings_for(c, [1, 1, 1], D, dev) for c in cams_a]
    res1 = rasterize_views_forward_raw(sets, t["means3D"], t["opacities"], t["shs"], None, t["scales"], t["rotations"], None)
    res2 = rasterize_views_forward_raw(sets, t["means3D"], t["opacities"], t["shs"], None, t["scales"], t["rotations"], None)
    for (o1, _), (o2, _) in zip(res1, res2):
        assert torch.equal(o1["color"], o2["color"]) and torch.equal(o1["radii"], o2["radii"]) and o1["N"] == o2["N"]


@pytest.mark.gpu
def test_batched_views_recover_from_capacity_overflow(built_lib):
    """The speculated pair capacity of the batch is too small for the next call (the splats grew): every view is redone
    exactly; results equal the exact two-phase forward."""
    from dreamscene_amd import rasterizer as R, synth
    from dreamscene_amd.views import rasterize_views_forward_raw
    dev = torch.device("cuda:0")
    P, K, D, H, W, V = 2000, 4, 1, 96, 128, 3
    g, _ = small_scene(P=P, H=H, W=W, K=K, seed=41, scale_mul=1.0)
    t = {k: torch.tensor(v, device=dev) for k, v in g.items()}
    cams = synth.object_cameras(V + 1, H, W, radius=3.0)[1:]
    sets = [settings_for(c, [0, 0, 0], D, dev) for c in cams]
    args = lambda sc: (t["means3D"], t["opacities"], t["shs"], None, sc, t["rotations"], None)
    for _ in range(2):                                    # unbatched, then batched: the hint is the small N
        small = rasterize_views_forward_raw(sets, *args(t["scales"]))
    big_scales = t["scales"] * 6.0
    res = rasterize_views_forward_raw(sets, *args(big_scales))
    assert all(o["N"] > 2.5 * s_["N"] for (o, _), (s_, _) in zip(res, small)), "the test must overflow the speculation"
    for s, (o, _) in zip(sets, res):
        ref, _ = R.rasterize_forward_raw(s, *args(big_scales), mode="sync")
        assert o["N"] == ref["N"]
        assert torch.equal(o["radii"], ref["radii"])
        assert torch.equal(o["color"], ref["color"]) and torch.equal(o["depth_alpha"], ref["depth_alpha"])


@pytest.mark.gpu
def test_batched_views_with_precomputed_colours(built_lib):
    """Inputs the fused K1 / K8 passes do not cover (colors_precomp): the batch still shares the sorts and runs K1 / K8
    view by view, later views added to view 0's gradients."""
    from dreamscene_amd import synth
    from dreamscene_amd.rasterizer import GaussianRasterizer
    from dreamscene_amd.views import GaussianRasterizerViews
    dev = torch.device("cuda:0")
    P, H, W, V = 900, 80, 96, 3
    g, _ = small_scene(P=P, H=H, W=W, K=1, seed=61)
    cams = synth.object_cameras(V + 1, H, W, radius=3.0)[1:]
    sets = [settings_for(c, [0, 0, 0], 0, dev) for c in cams]
    t = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in g.items() if k != "shs"}
    cols = torch.rand((P, 3), device=dev, requires_grad=True)
    leaves = [t["means3D"], cols, t["opacities"], t["scales"], t["rotations"]]
    gis = [torch.tensor(synth.upstream_grads(H, W, seed=k)[0], device=dev) for k in range(V)]
    gdas = [torch.tensor(synth.upstream_grads(H, W, seed=k)[1], device=dev) for k in range(V)]
    tot = None
    for k, s in enumerate(sets):
        m2d = torch.zeros((P, 3), device=dev, requires_grad=True)
        img, radii, da = GaussianRasterizer(s)(means3D=t["means3D"], means2D=m2d, colors_precomp=cols,
                                                opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
        gr = torch.autograd.grad([img, da], leaves, [gis[k], gdas[k]])
        tot = list(gr) if tot is None else [a + b for a, b in zip(tot, gr)]
    rast = GaussianRasterizerViews(sets)
    for rep in range(2):
        m2d = torch.zeros((V, P, 3), device=dev, requires_grad=True)
        outs = rast(means3D=t["means3D"], means2D=m2d, colors_precomp=cols, opacities=t["opacities"],
                    scales=t["scales"], rotations=t["rotations"])
        grads = torch.autograd.grad([x for (img, _, da) in outs for x in (img, da)], leaves,
                                    [y for k in range(V) for y in (gis[k], gdas[k])])
    for a, b in zip(grads, tot):
        assert tol_ok(a.cpu().numpy(), b.cpu().numpy(), atol=3e-6)


def test_views_of_one_call_must_share_image_size():
    """a mixed batch used to render and then fail in backward. Now it is refused up front (CPU: no GPU needed)."""
    from dreamscene_amd.rasterizer import GaussianRasterizationSettings
    from dreamscene_amd.views import GaussianRasterizerViews
    mk = lambda h, w, sm=1.0: GaussianRasterizationSettings(
        image_height=h, image_width=w, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3), scale_modifier=sm,
        viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0, campos=torch.zeros(3), prefiltered=False)
    GaussianRasterizerViews([mk(32, 48), mk(32, 48)])
    with pytest.raises(ValueError, match="same image_height"):
        GaussianRasterizerViews([mk(32, 48), mk(48, 32)])
    with pytest.raises(ValueError, match="scale_modifier"):
        GaussianRasterizerViews([mk(32, 48), mk(32, 48, sm=2.0)])
    with pytest.raises(ValueError):
        GaussianRasterizerViews([])


@pytest.mark.gpu
def test_batched_views_accept_means2D_none(built_lib):
    """GaussianRasterizer takes means2D=None (nobody wants the screen-space gradient: inference renders); so does the batched
    module -- same images, and a backward to the parameters alone."""
    from dreamscene_amd import synth
    from dreamscene_amd.views import GaussianRasterizerViews
    dev = torch.device("cuda:0")
    P, H, W, V = 800, 64, 80, 3
    g, _ = small_scene(P=P, H=H, W=W, K=16, seed=5)
    cams = synth.object_cameras(V + 1, H, W, radius=3.0)[1:]
    sets = [settings_for(c, [1.0, 1.0, 1.0], 3, dev) for c in cams]
    t = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in g.items()}
    kw = dict(means3D=t["means3D"], shs=t["shs"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    ref = GaussianRasterizerViews(sets)(means2D=torch.zeros((V, P, 3), device=dev, requires_grad=True), **kw)
    gref = torch.autograd.grad([o[0].sum() + o[2].sum() for o in ref], [t["means3D"], t["opacities"]], [torch.ones((), device=dev)] * V)
    out = GaussianRasterizerViews(sets)(means2D=None, **kw)
    got = torch.autograd.grad([o[0].sum() + o[2].sum() for o in out], [t["means3D"], t["opacities"]], [torch.ones((), device=dev)] * V)
    for a, b in zip(out, ref):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for a, b in zip(got, gref):
        assert tol_ok(a.cpu().numpy(), b.cpu().numpy(), atol=2e-6)
    with torch.no_grad():
        out2 = GaussianRasterizerViews(sets)(means2D=None, **kw)
    assert all(torch.equal(a[0], b[0]) for a, b in zip(out2, ref))
