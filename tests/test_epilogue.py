"""post-raster epilogue -- densification statistics fused into K8, one-launch Adam, importance
filtering threshold. Oracles: the reference's own functions (golden fixture) and torch.optim.Adam on the CPU, which is
the optimizer the reference instantiates (gs_renderer.py:653)."""
import numpy as np
import pytest
import torch

from tests.test_golden import load


def test_importance_prune_mask_matches_reference():
    from dreamscene_amd import densify
    d = load("prune.npz")
    scaling = torch.exp(torch.tensor(d["scaling"]))
    v = densify.v_importance(scaling, torch.tensor(d["imp"]), float(d["v_pow"]))
    np.testing.assert_allclose(v.numpy(), d["v_list"], rtol=1e-6)
    mask = densify.importance_prune_mask(torch.tensor(d["v_list"]), float(d["percent"]))
    assert np.array_equal(mask.numpy(), d["mask"])
    assert 0.35 < mask.float().mean() < 0.45


@pytest.mark.gpu
def test_fused_adam_matches_torch_adam(built_lib):
    from dreamscene_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    shapes = [(1001, 3), (1001, 1, 3), (1001, 15, 3), (1001, 1), (1001, 3), (1001, 4), (3,)]
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-2, 5e-3, 1e-3, 2.5e-3]
    names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "background"]
    cpu = [torch.randn(s, generator=gen).requires_grad_(True) for s in shapes]
    hip = [t.detach().clone().to(dev).requires_grad_(True) for t in cpu]
    mk = lambda ps: [{"params": [p], "lr": lr, "name": n} for p, lr, n in zip(ps, lrs, names)]
    ref = torch.optim.Adam(mk(cpu), lr=0.0, eps=1e-15)          # as gs_renderer.py:653
    opt = FusedAdam(mk(hip), lr=0.0, eps=1e-15)
    for step in range(6):
        for k, (a, b) in enumerate(zip(cpu, hip)):
            g = torch.randn(a.shape, generator=gen) * (10.0 ** (k - 3))
            if step == 2 and k == 3:
                g.zero_()                                        # an all-zero gradient still moves the moments
            a.grad = g.clone()
            b.grad = g.to(dev)
        for go, gr in zip(opt.param_groups, ref.param_groups):   # the reference rewrites lr every step
            if go["name"] == "xyz":
                go["lr"] = gr["lr"] = lrs[0] * (0.9 ** step)
        ref.step()
        opt.step(zero_grad=(step == 4))
        if step == 4:
            assert all(float(b.grad.abs().max()) == 0.0 for b in hip)
    for a, b, n in zip(cpu, hip, names):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().numpy(), rtol=2e-6, atol=1e-7, err_msg=n)
        sa, sb = ref.state[a], opt.state[b]
        assert float(sa["step"]) == float(sb["step"]) == 6.0
        # moments: entries that nearly cancel carry the rounding of the larger terms -> tolerance relative to the tensor
        for key in ("exp_avg", "exp_avg_sq"):
            r = sa[key].numpy()
            np.testing.assert_allclose(sb[key].cpu().numpy(), r, rtol=2e-6, atol=2e-6 * float(np.abs(r).max()),
                                       err_msg=f"{n} {key}")


@pytest.mark.gpu
def test_fused_adam_takes_arena_gradients(built_lib):
    """grads= : the optimizer consumes the flat GradArena the backward wrote (no .grad tensors at all)."""
    from dreamscene_amd import multiview
    from dreamscene_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    P, K = 777, 4
    arena = multiview.GradArena(P, K, dev)
    gen = torch.Generator().manual_seed(1)
    order = ["means3D", "shs", "opacities", "scales", "rotations"]
    params = [torch.randn(arena.views[n].shape, generator=gen).to(dev).requires_grad_(True) for n in order]
    ref_p = [p.detach().cpu().clone().requires_grad_(True) for p in params]
    arena.flat.copy_(torch.randn(arena.flat.shape, generator=gen).to(dev))
    for rp, n in zip(ref_p, order):
        rp.grad = arena.views[n].cpu().clone()
    opt = FusedAdam([{"params": [p], "lr": 1e-2} for p in params], eps=1e-15)
    ref = torch.optim.Adam([{"params": [p], "lr": 1e-2} for p in ref_p], eps=1e-15)
    opt.step(grads=[arena.views[n] for n in order], zero_grad=True)
    ref.step()
    for a, b in zip(ref_p, params):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().numpy(), rtol=2e-6, atol=1e-7)
    assert all(float(arena.views[n].abs().max()) == 0.0 for n in order)   # (alignment padding between regions is not touched)


@pytest.mark.gpu
@pytest.mark.parametrize("fused_scene", [False, True])
def test_densify_stats_fused_into_backward(built_lib, fused_scene):
    """stats.collect(): K8 updates max_radii2D / xyz_gradient_accum / denom exactly as the trainer's indexing ops would
    (object_trainer.py:386-390, gs_renderer.py:1061-1065), for two views in a row."""
    from dreamscene_amd import densify, scene, synth
    from dreamscene_amd.rasterizer import GaussianRasterizer, RasterContext
    from tests.util import settings_for, small_scene
    dev = torch.device("cuda:0")
    rc = RasterContext()
    g, _ = small_scene(P=900, H=96, W=96, K=16, seed=9)
    P = 900
    cams = synth.object_cameras(3, 96, 96, radius=3.0)
    t = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in g.items()}
    op = np.clip(g["opacities"], 1e-4, 1 - 1e-4)
    raw = (g["means3D"], np.log(g["scales"]), g["rotations"], np.log(op / (1 - op)), g["shs"][:, :1], g["shs"][:, 1:])
    model = tuple(torch.tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev, requires_grad=True) for a in raw)
    gi_np, gda_np = synth.upstream_grads(96, 96, seed=3)
    gi, gda = torch.tensor(gi_np, device=dev), torch.tensor(gda_np, device=dev)
    stats = densify.DensifyStats(P, dev)
    exp_r, exp_a, exp_d = (torch.zeros(P, device=dev) for _ in range(3))
    for cam in cams[1:]:
        s = settings_for(cam, [1, 1, 1], 3, dev)
        m2d = torch.zeros((P, 3), device=dev, requires_grad=True)
        with stats.collect(rc):      # the FORWARD inside the block takes the snapshot; the backward may run anywhere
            if fused_scene:
                img, radii, da, _ = scene.rasterize_models(s, [model], m2d, context=rc)
            else:
                img, radii, da = GaussianRasterizer(s, context=rc)(means3D=t["means3D"], means2D=m2d, shs=t["shs"],
                                                                    opacities=t["opacities"], scales=t["scales"],
                                                                    rotations=t["rotations"])
        ((img * gi).sum() + (da * gda).sum()).backward()
        vis = radii > 0
        exp_r[vis] = torch.max(exp_r[vis], radii[vis].float())
        exp_a[vis] += torch.norm(m2d.grad[vis, :2], dim=-1)
        exp_d[vis] += 1
    assert int(exp_d.max()) == 2 and int((exp_d == 0).sum()) >= 0
    assert torch.equal(stats.max_radii2D, exp_r)
    assert torch.equal(stats.denom, exp_d)
    np.testing.assert_allclose(stats.xyz_gradient_accum.cpu().numpy(), exp_a.cpu().numpy(), rtol=1e-6, atol=1e-12)
    assert not stats.mean_grad().isnan().any()
