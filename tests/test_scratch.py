"""-m gpu: the K7 -> K8 scratch contract of the C ABI (include/gsrast.h, GsrGrads.reach / .scratch_clean). With
scratch_clean the caller keeps `partials` + `reach` between calls, hands them over ALL ZERO and gets them back all zero --
no clear is launched; without it the library clears whatever it is given. Both protocols give the same gradients, for
every form of K8 (the sparse and dense views kernels restore the scratch themselves; the scene forms and the single-view
kernel with camera gradients get it cleared behind them)."""
import numpy as np
import pytest
import torch

from tests.util import rel_scale, same_bits, settings_for, small_scene, tol_ok

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _all_scratch_zero(R):
    n = 0
    for sc in R._SCRATCH.values():
        assert not sc.dirty
        assert int(torch.count_nonzero(sc.partials)) == 0, "partials not restored to zero"
        assert int(torch.count_nonzero(sc.reach)) == 0, "reach marks not restored to zero"
        n += 1
    return n


def _legacy_bind(gr, sc, k=0):
    """the protocol without the contract: garbage in, the library clears (and may leave anything behind)"""
    sc.partials[k].fill_(float("nan"))
    gr.partials = sc.partials[k].data_ptr()
    gr.reach = None
    gr.scratch_clean = 0


def _legacy_bind_reach(gr, sc, k=0):
    """reach given, but no promise: the library clears both first"""
    sc.partials[k].fill_(float("nan"))
    sc.reach[k].fill_(7)
    gr.partials = sc.partials[k].data_ptr()
    gr.reach = sc.reach[k].data_ptr()
    gr.scratch_clean = 0


@pytest.mark.parametrize("K,D,V,cam_grads", [(16, 3, 3, False), (4, 1, 2, False), (16, 3, 1, False), (16, 3, 1, True),
                                             (9, 2, 4, False)])
def test_scratch_is_returned_clean_and_protocols_agree(built_lib, monkeypatch, K, D, V, cam_grads):
    from dreamscene_amd import rasterizer as R, synth
    from dreamscene_amd.views import GaussianRasterizerViews
    dev = torch.device(DEV)
    P, H, W = 2500, 112, 144
    g, _ = small_scene(P=P, H=H, W=W, K=K, seed=13)
    cams = synth.object_cameras(V + 1, H, W, radius=3.0)[1:]
    t = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in g.items()}
    leaves = [t[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    gis = [torch.tensor(synth.upstream_grads(H, W, seed=k)[0], device=dev) for k in range(V)]
    gdas = [torch.tensor(synth.upstream_grads(H, W, seed=k)[1], device=dev) for k in range(V)]
    sets = [settings_for(c, [1, 1, 1], D, dev) for c in cams]
    if cam_grads:      # camera tensors that require grad: the single-view kernel that does not restore the scratch itself
        sets = [s._replace(viewmatrix=s.viewmatrix.clone().requires_grad_(True)) for s in sets]

    def run():
        R._SCRATCH.clear()
        res = None
        for rep in range(3):       # (first call of the views path runs view by view; the later ones batched)
            if V == 1:
                m2d = torch.zeros((P, 3), device=dev, requires_grad=True)
                img, radii, da = R.GaussianRasterizer(sets[0])(means3D=t["means3D"], means2D=m2d, shs=t["shs"],
                                                                opacities=t["opacities"], scales=t["scales"],
                                                                rotations=t["rotations"])
                extra = [sets[0].viewmatrix] if cam_grads else []
                res = torch.autograd.grad([img, da], leaves + [m2d] + extra, [gis[0], gdas[0]])
            else:
                m2d = torch.zeros((V, P, 3), device=dev, requires_grad=True)
                outs = GaussianRasterizerViews(sets)(means3D=t["means3D"], means2D=m2d, shs=t["shs"],
                                                     opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
                res = torch.autograd.grad([x for (img, _, da) in outs for x in (img, da)], leaves + [m2d],
                                          [y for k in range(V) for y in (gis[k], gdas[k])])
            torch.cuda.synchronize()
        return [x.detach().cpu().numpy() for x in res]

    clean = run()
    assert _all_scratch_zero(R) >= 1
    assert sum(float(np.abs(x).sum()) for x in clean) > 0
    monkeypatch.setattr(R, "_bind_scratch", _legacy_bind)
    legacy = run()
    monkeypatch.setattr(R, "_bind_scratch", _legacy_bind_reach)
    legacy_reach = run()
    for a, b, c in zip(clean, legacy, legacy_reach):
        assert np.isfinite(b).all() and np.isfinite(c).all()
        assert tol_ok(a, b, atol=2e-6) and tol_ok(a, c, atol=2e-6)      # (fp32 atomics: order varies run to run)


def test_scene_forms_clear_the_scratch_behind_them(built_lib):
    """raw-leaf scene input (dreamscene_amd/scene.py): K8's scene kernels read the sums directly; the library clears the
    scratch after them, so the contract holds for the next call on the same buffers."""
    from dreamscene_amd import rasterizer as R, scene
    from tests.test_golden import _cam_from_fixture, load
    from tests.test_scene import _models_from_fixture, _loss
    d = load("scene_render.npz")
    dev = torch.device(DEV)
    R._SCRATCH.clear()
    ref = None
    for rep in range(2):
        models = _models_from_fixture(d, device=dev)
        out = scene.scene_render(models, _cam_from_fixture(d), torch.tensor(d["bg"], device=dev), int(d["active_sh_degree"]),
                                 test=True)
        _loss(out, d, dev).backward()
        torch.cuda.synchronize()
        grads = [t.grad.detach().cpu().numpy() for m in models for t in m]
        assert _all_scratch_zero(R) >= 1
        if ref is None:
            ref = grads
        else:
            for a, b in zip(grads, ref):      # (run-to-run: the order of K7's fp32 atomics)
                assert float(np.abs(a - b).max()) <= 1e-5 * rel_scale(b)


def test_a_failed_enqueue_releases_the_scratch_and_the_next_call_rezeroes_it(built_lib, monkeypatch):
    """the persistent K7 -> K8 scratch is locked for the length of one enqueue (two host threads on a stream must
    not interleave K7(A), K7(B), K8(A)); a call that raises half-way releases the lock, leaves the scratch flagged dirty, and
    the next call zeroes it before use."""
    from dreamscene_amd import rasterizer as R, synth, _lib as L
    dev = torch.device(DEV)
    P, H, W, K, D = 1500, 96, 96, 16, 3
    g, _ = small_scene(P=P, H=H, W=W, K=K, seed=17)
    cam = synth.object_cameras(2, H, W, radius=3.0)[1]
    t = {k: torch.tensor(v, device=dev) for k, v in g.items()}
    s = settings_for(cam, [1, 1, 1], D, dev)
    gi, gda = (torch.tensor(x, device=dev) for x in synth.upstream_grads(H, W, 0))

    def backward():
        out, st = R.rasterize_forward_raw(s, t["means3D"], t["opacities"], t["shs"], None, t["scales"], t["rotations"], None)
        return R.rasterize_backward_raw(st, gi, gda)
    ref = backward()
    torch.cuda.synchronize()
    sc = list(R._SCRATCH.values())[-1]              # (most recently used: this call's)
    assert sc.partials.shape[:2] == (1, P)
    real_check = L.check

    def failing_check(code, what):
        if what == "gsr_backward":
            sc.partials.fill_(3.0)              # (what a half-enqueued K7 would leave behind)
            raise L.GsrError("injected failure")
        return real_check(code, what)
    monkeypatch.setattr(R.L, "check", failing_check)
    with pytest.raises(L.GsrError):
        backward()
    monkeypatch.setattr(R.L, "check", real_check)
    assert sc.dirty and not sc.lock.locked()
    again = backward()
    torch.cuda.synchronize()
    assert not sc.dirty and not sc.lock.locked()
    for k in ("dL_dmeans3D", "dL_dshs", "dL_dscales", "dL_drotations", "dL_dopacities"):
        same_bits(ref[k], again[k], k)
    assert int(torch.count_nonzero(sc.partials)) == 0
