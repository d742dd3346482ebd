"""What crosses the rasterizer boundary when the REFERENCE's own object_render / scene_render run (fp32) -- recorded by
tests/golden/make_golden.py from /root/reference over the scalar C oracle (tests/golden/raster_boundary.npz): the 12
settings fields the reference constructed, the activated / noise-augmented tensors it handed over, the outputs, the
upstream gradients autograd delivered (through the reference's disp post-processing) and the gradients returned.
Cases: object_render test=True and test=False (two seeds: SH degree dropped / random background / SH + scale noise),
scene_render (three models concatenated) test=True and test=False (scene_gaussian.py:673-893, 895-1044).
CPU: the oracle replays the records. GPU: the HIP rasterizer, behind the drop-in module, replays them: radii bit-exact,
images / depth / alpha at 1e-5. Gradients: the reference's disp normalisation ((disp - min) / (max - min),
scene_gaussian.py:1025-1032) puts the whole normalisation gradient on the two or three pixels that hold the extrema:
|dL/d(depth, alpha)| reaches 7e4 there against <= 1 elsewhere. An fp32 backward carries ~1e-7 x 7e4 of absolute noise
through those pixels whatever its operator order (the lineage's included), so with the upstream EXACTLY as recorded the
gradients are held to 3e-5 (and to identical bits over eight runs); with the same upstream clipped at |g| <= 50 (50 x the
typical weight; the 99.9th percentile is ~220) they are held to 1e-5 like everywhere else, against the oracle re-run on the
clipped upstream."""
import os

import numpy as np
import pytest
import torch

from tests.util import rel_scale, same_bits

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-5


def _cases():
    d = np.load(os.path.join(HERE, "golden", "raster_boundary.npz"))
    out = {}
    for name in d["cases"]:
        name = str(name)
        c = {"settings": {}, "inputs": {}, "upstream": {}, "grads": {}, "out": {}}
        for k in d.files:
            if k.startswith(name + "/"):
                _, grp, key = k.split("/")
                c[grp][key] = d[k]
        out[name] = c
    return out


CASES = _cases()


def _close(a, ref, what, tol=TOL):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    e = float(np.abs(a - ref).max()) if a.size else 0.0
    assert e <= tol * rel_scale(ref), f"{what}: max abs err {e:.3e} (max|ref| {np.abs(ref).max():.3e})"


def test_fixture_covers_the_reference_calls():
    assert set(CASES) == {"object_test", "object_train31", "object_train7", "scene_test", "scene_train11"}
    s = CASES["object_train31"]["settings"]
    assert int(s["sh_degree"]) == 0                      # the sh_deg_aug branch of scene_gaussian.py:938-947 was taken
    assert not np.array_equal(CASES["object_train31"]["inputs"]["scales"], CASES["object_test"]["inputs"]["scales"])
    assert CASES["scene_test"]["inputs"]["means3D"].shape[0] == 300 + 517 + 130     # three models, torch.cat order
    assert not np.array_equal(CASES["scene_train11"]["inputs"]["shs"], CASES["scene_test"]["inputs"]["shs"])


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_replays_the_record(c_oracle, name):
    c = CASES[name]
    s, a = c["settings"], c["inputs"]
    P, K = a["means3D"].shape[0], a["shs"].shape[1]
    v = c_oracle.make_view(P, K, int(s["sh_degree"]), int(s["image_height"]), int(s["image_width"]), float(s["tanfovx"]),
                           float(s["tanfovy"]), s["bg"], s["viewmatrix"], s["projmatrix"], s["campos"],
                           scale_modifier=float(s["scale_modifier"]))
    f = c_oracle.forward(v, a["means3D"], a["opacities"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
    assert np.array_equal(f["radii"], c["out"]["radii"]) and np.array_equal(f["image"], c["out"]["image"])
    b = c_oracle.backward(v, f, c["upstream"]["dL_dimage"], c["upstream"]["dL_ddepth_alpha"], a["means3D"], shs=a["shs"],
                          scales=a["scales"], rotations=a["rotations"])
    for k, ref in c["grads"].items():
        assert np.array_equal(np.asarray(b[k]).reshape(ref.shape), ref), k


def _hip_replay(c, up_img, up_da):
    from dreamscene_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    s, a = c["settings"], c["inputs"]
    t = lambda x: torch.tensor(np.asarray(x, np.float32), device=dev)
    settings = GaussianRasterizationSettings(
        image_height=int(s["image_height"]), image_width=int(s["image_width"]), tanfovx=float(s["tanfovx"]),
        tanfovy=float(s["tanfovy"]), bg=t(s["bg"]), scale_modifier=float(s["scale_modifier"]), viewmatrix=t(s["viewmatrix"]),
        projmatrix=t(s["projmatrix"]), sh_degree=int(s["sh_degree"]), campos=t(s["campos"]), prefiltered=False, score_flag=False)
    p = {k: t(v).requires_grad_(True) for k, v in a.items()}
    m2d = torch.zeros_like(p["means3D"], requires_grad=True)
    img, radii, da = GaussianRasterizer(raster_settings=settings)(
        means3D=p["means3D"], means2D=m2d, shs=p["shs"], colors_precomp=None, opacities=p["opacities"], scales=p["scales"],
        rotations=p["rotations"], cov3D_precomp=None)
    torch.autograd.backward([img, da], [t(up_img), t(up_da)])
    got = dict(dL_dmeans3D=p["means3D"].grad, dL_dmeans2D=m2d.grad, dL_dopacity=p["opacities"].grad, dL_dshs=p["shs"].grad,
               dL_dscales=p["scales"].grad, dL_drotations=p["rotations"].grad)
    return img.detach().cpu().numpy(), radii.cpu().numpy(), da.detach().cpu().numpy(), {k: v.cpu().numpy() for k, v in got.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_replays_the_record(built_lib, c_oracle, name):
    c = CASES[name]
    up_img, up_da = c["upstream"]["dL_dimage"], c["upstream"]["dL_ddepth_alpha"]
    img, radii, da, got = _hip_replay(c, up_img, up_da)
    assert np.array_equal(radii, c["out"]["radii"]), "radii"
    _close(img, c["out"]["image"], "image")
    _close(da, c["out"]["depth_alpha"], "depth_alpha")
    assert float(np.abs(up_da).max()) > 1e4            # the spike of the disp normalisation is in the record
    # With the upstream as recorded the disp normalisation puts |dL/d(depth, alpha)| = 7e4 on two pixels (<= 1 elsewhere).
    # Rounds 1-3 needed 3e-4 here: (i) the order of K7's fp32 atomics moved dL/dopacity between 1e-5 and 1e-4 from run to run --
    # the cross-wave sums are now added in double; (ii) s - R, the difference of two ~4e5 dot products that cancels in front of
    # an opaque object, carried eps x 4e5 -- K7 now carries the behind-state relative to the segment's depth (render.hip,
    # render_bwd_body). What is left is fp32 per-pixel arithmetic on 7e4-weighted terms: measured <= 2.4e-5, bar 3e-5, and
    # the eight runs below give the same bits.
    for k, ref in c["grads"].items():
        _close(got[k].reshape(ref.shape), ref, k + " (upstream as recorded)", tol=3e-5)
    for rep in range(7):
        _, _, _, again = _hip_replay(c, up_img, up_da)
        for k in got:
            same_bits(got[k], again[k], f"{k}: run {rep + 1} vs run 0")
    # the same record with the spike pixels clipped: the usual bar
    cl_img, cl_da = np.clip(up_img, -50.0, 50.0), np.clip(up_da, -50.0, 50.0)
    _, _, _, got = _hip_replay(c, cl_img, cl_da)
    s, a = c["settings"], c["inputs"]
    P, K = a["means3D"].shape[0], a["shs"].shape[1]
    v = c_oracle.make_view(P, K, int(s["sh_degree"]), int(s["image_height"]), int(s["image_width"]), float(s["tanfovx"]),
                           float(s["tanfovy"]), s["bg"], s["viewmatrix"], s["projmatrix"], s["campos"],
                           scale_modifier=float(s["scale_modifier"]))
    f = c_oracle.forward(v, a["means3D"], a["opacities"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
    b = c_oracle.backward(v, f, cl_img, cl_da, a["means3D"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
    for k in c["grads"]:
        ref = np.asarray(b[k])
        _close(got[k].reshape(ref.shape), ref, k + " (upstream clipped at 50)")
