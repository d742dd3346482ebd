"""-m gpu: batched importance scoring (views.importance_scores, score_flag in GaussianRasterizerViews) against the scalar C
oracle -- the `prune_list` loop of the reference (scene_gaussian.py:1063-1079): 48 sphere cameras, one score render each,
scores summed. 100 k Gaussians @512^2 (BASELINE C2's size), both score weights (SEMANTICS.md section 4)."""
import numpy as np
import pytest
import torch

from tests.util import oracle_view, rel_scale, settings_for

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
P, K, D, RES, NCAM = 100_000, 16, 3, 512, 48


@pytest.fixture(scope="module")
def scene():
    from dreamscene_amd import synth
    g = synth.g_object(P, seed=0, K=K)
    cams = synth.sphere_cameras(NCAM, RES, RES)
    return g, cams


@pytest.mark.parametrize("score_mode", [0, 1])
def test_sphere_camera_score_sum_vs_oracle(built_lib, c_oracle, scene, score_mode):
    from dreamscene_amd import views
    from dreamscene_amd.rasterizer import GaussianRasterizer, RasterContext
    g, cams = scene
    gd = {k: torch.tensor(v, device=DEV) for k, v in g.items()}
    bg = np.ones(3, np.float32)
    sl = [settings_for(c, bg, D, DEV, score_flag=True) for c in cams]
    rc = RasterContext(score_mode=score_mode)
    args = dict(means3D=gd["means3D"], opacities=gd["opacities"], shs=gd["shs"], scales=gd["scales"], rotations=gd["rotations"])
    # twice: the first call learns the pair counts (views one by one), the second one runs the batched launches
    for _ in range(2):
        total = views.importance_scores(sl, context=rc, **args)
    torch.cuda.synchronize()
    # the reference's loop: one GaussianRasterizer call per camera, scores added on the host side
    loop = torch.zeros(P, device=DEV)
    m2d = torch.zeros_like(gd["means3D"])
    with torch.no_grad():
        for s in sl[:8]:
            sc, img, radii, da = GaussianRasterizer(raster_settings=s, context=rc)(means2D=m2d, **args)
            loop += sc
    # C oracle, all 48 cameras
    ref = np.zeros(P, np.float64)
    ref8 = None
    for i, c in enumerate(cams):
        v = oracle_view(c_oracle, c, P, K, D, bg, score_mode=score_mode)
        f = c_oracle.forward(v, g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"],
                             score=True)
        ref += f["important_score"].astype(np.float64)
        if i == 7:
            ref8 = ref.copy()
    scale = rel_scale(ref)
    got = total.cpu().numpy().astype(np.float64)
    assert np.abs(ref).max() > 0
    assert np.abs(got - ref).max() <= 1e-5 * scale, f"48-camera score sum: {np.abs(got - ref).max():.3e} (scale {scale:.3e})"
    assert np.abs(loop.cpu().numpy() - ref8).max() <= 1e-5 * rel_scale(ref8)
    if score_mode == 0:
        # weight = opacity per contributing (pixel, splat): the kernels count pixels with integer atomics, so the number of
        # contributing pixels of every Gaussian over the 48 views is EXACT (no summation order involved)
        op = g["opacities"].reshape(-1).astype(np.float64)
        ok = op > 1e-4
        assert np.array_equal(np.rint(got[ok] / op[ok]), np.rint(ref[ok] / op[ok])), "contributing-pixel counts differ"


def test_views_module_returns_four_tuples(built_lib, scene):
    from dreamscene_amd.rasterizer import GaussianRasterizer
    from dreamscene_amd.views import GaussianRasterizerViews
    g, cams = scene
    gd = {k: torch.tensor(v, device=DEV) for k, v in g.items()}
    bg = np.ones(3, np.float32)
    sl = [settings_for(c, bg, D, DEV, score_flag=True) for c in cams[:4]]
    m2d = torch.zeros((4, P, 3), device=DEV)
    args = dict(opacities=gd["opacities"], shs=gd["shs"], scales=gd["scales"], rotations=gd["rotations"])
    rast = GaussianRasterizerViews(sl)
    for _ in range(2):
        outs = rast(means3D=gd["means3D"], means2D=m2d, **args)
    assert len(outs) == 4 and all(len(o) == 4 for o in outs)
    with torch.no_grad():
        for k in (0, 3):
            sc, img, radii, da = GaussianRasterizer(raster_settings=sl[k])(means3D=gd["means3D"], means2D=m2d[k], **args)
            assert torch.equal(img, outs[k][1]) and torch.equal(radii, outs[k][2]) and torch.equal(da, outs[k][3])
            assert float((sc - outs[k][0]).abs().max()) <= 1e-5 * rel_scale(sc)
    with pytest.raises(ValueError):
        GaussianRasterizerViews([sl[0], settings_for(cams[1], bg, D, DEV)])(means3D=gd["means3D"], means2D=m2d[:2], **args)
