"""-m gpu: the drop-in module replayed from captured graphs (dreamscene_amd/dropin.py). The reference's trainers call
GaussianRasterizer once per view, keep all outputs of a step, then backpropagate through all of them
(training/object_trainer.py:302-382, scene_gaussian.py:966-1021); the captured path must be indistinguishable from the
eager one: same outputs (bit for bit), same gradients (the backward is bit-reproducible), outputs that stay valid however
long the caller keeps them, any order of backwards."""
import numpy as np
import pytest
import torch

from tests.util import same_bits, settings_for

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _graphs_on(monkeypatch):
    """The captured path is opt-in (GSR_DROPIN_GRAPHS=1); these tests switch it on for modules built without a context,
    which is how the reference builds them (scene_gaussian.py:966)."""
    from dreamscene_amd import dropin
    monkeypatch.setattr(dropin, "ENABLED", True)
    yield
    dropin.reset()


def _scene(P=20_000, K=16, res=256, n_cams=4, seed=3):
    from dreamscene_amd import synth
    g = synth.g_object(P, seed=seed, K=K)
    cams = synth.object_cameras(8, res, res)[:n_cams]
    ups = [tuple(torch.tensor(x, device=DEV) for x in synth.upstream_grads(res, res, i)) for i in range(n_cams)]
    return g, cams, ups


def _step(params, cams, ups, D, context=None, order=None):
    """One trainer-style step: V forwards, then the backward of every view (one loss over all of them)."""
    from dreamscene_amd.rasterizer import GaussianRasterizer
    outs, m2ds = [], []
    for cam in cams:
        m2d = torch.zeros_like(params["means3D"], requires_grad=True)
        s = settings_for(cam, np.ones(3, np.float32), D, DEV)
        img, radii, da = GaussianRasterizer(raster_settings=s, context=context)(
            means3D=params["means3D"], means2D=m2d, shs=params["shs"], opacities=params["opacities"],
            scales=params["scales"], rotations=params["rotations"])
        outs.append((img, radii, da))
        m2ds.append(m2d)
    grads = [None] * len(cams)
    for j in (order if order is not None else reversed(range(len(cams)))):
        img, _, da = outs[j]
        gr = torch.autograd.grad([img, da], [params[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")] + [m2ds[j]],
                                 list(ups[j]))
        grads[j] = [t.clone() for t in gr]          # (the captured path returns a slot's static tensors: copy before the next step)
    torch.cuda.synchronize()
    return outs, grads


def test_captured_dropin_equals_eager_bit_for_bit(built_lib):
    from dreamscene_amd import dropin
    from dreamscene_amd.rasterizer import RasterContext
    dropin.reset()
    g, cams, ups = _scene()
    params = {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in g.items()}
    ref_outs, ref_grads = _step(params, cams, ups, 3, context=RasterContext(dropin_graphs=False))
    ref_outs = [tuple(t.clone() for t in o) for o in ref_outs]
    kept = []
    for step in range(5):       # steps 0-1 warm up (eager), 2-3 capture the four slots, 4 replays all of them
        order = [2, 0, 3, 1] if step == 3 else None
        outs, grads = _step(params, cams, ups, 3, order=order)
        kept.append(outs)
        for j in range(len(cams)):
            for a, b, what in zip(outs[j], ref_outs[j], ("image", "radii", "depth_alpha")):
                assert torch.equal(a, b), f"step {step} view {j}: {what} differs from the eager path"
            for a, b, what in zip(grads[j], ref_grads[j], ("means3D", "shs", "opacities", "scales", "rotations", "means2D")):
                same_bits(a, b, f"step {step} view {j}: dL/d{what} vs the eager path")
    st = list(dropin.stats().values())
    assert len(st) == 1 and st[0]["slots"] == 4 and st[0]["replays"] >= 8 and st[0]["no_slot"] == 0, st
    # outputs handed out by earlier steps are the caller's own tensors: later replays did not touch them
    for outs in kept:
        for j in range(len(cams)):
            assert torch.equal(outs[j][0], ref_outs[j][0]) and torch.equal(outs[j][2], ref_outs[j][2])


def test_forward_only_and_dropped_graphs_free_their_slots(built_lib):
    """no_grad calls (video_inference) take no lease; a call whose outputs are dropped without a backward frees its slot when
    the autograd graph dies; more views in flight than slots fall back to the eager path."""
    from dreamscene_amd import dropin
    from dreamscene_amd.rasterizer import GaussianRasterizer
    dropin.reset()
    g, cams, ups = _scene(P=8_000, res=128)
    params = {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in g.items()}
    s = settings_for(cams[0], np.ones(3, np.float32), 3, DEV)
    rast = GaussianRasterizer(raster_settings=s)
    kw = dict(means3D=params["means3D"], means2D=None, shs=params["shs"], opacities=params["opacities"], scales=params["scales"],
              rotations=params["rotations"])
    with torch.no_grad():
        imgs = [rast(**kw)[0] for _ in range(6)]
    assert all(torch.equal(imgs[0], im) for im in imgs[1:])
    st = list(dropin.stats().values())[0]
    assert st["slots"] == 1 and st["replays"] >= 3, st
    for _ in range(3):                        # graphs dropped without backward: the lease dies with them
        img, _, da = rast(**dict(kw, means2D=torch.zeros_like(params["means3D"], requires_grad=True)))
        assert torch.equal(img, imgs[0])
        del img, da
    assert list(dropin.stats().values())[0]["slots"] <= 2
    held = [rast(**dict(kw, means2D=torch.zeros_like(params["means3D"], requires_grad=True))) for _ in range(dropin.MAX_SLOTS + 3)]
    st = list(dropin.stats().values())[0]
    assert st["slots"] == dropin.MAX_SLOTS and st["no_slot"] >= 1, st
    for img, _, da in held:
        assert torch.equal(img, imgs[0])
    (gm,) = torch.autograd.grad([held[-1][0]], [params["means3D"]], [ups[0][0]])       # an eager-fallback call differentiates too
    assert float(gm.abs().max()) > 0
    del held


def test_fresh_input_tensors_every_call(built_lib):
    """The trainers hand over activations: new tensors every call (gs_renderer.py:464-488). Slots keyed on addresses that
    do not come back stage their inputs; results stay those of the eager path."""
    from dreamscene_amd import dropin
    from dreamscene_amd.rasterizer import GaussianRasterizer, RasterContext
    dropin.reset()
    g, cams, ups = _scene(P=8_000, res=128, n_cams=2)
    raw = {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in g.items()}
    keep_alive = []

    def act():
        p = {k: v * 1.0 for k, v in raw.items()}          # fresh non-leaf tensors (and kept alive: the addresses cannot repeat)
        keep_alive.append(p)
        return p
    ref = None
    for step in range(8):
        ctx = RasterContext(dropin_graphs=False) if step == 0 else None
        total = None
        for j, cam in enumerate(cams):
            p = act()
            m2d = torch.zeros_like(p["means3D"], requires_grad=True)
            img, radii, da = GaussianRasterizer(raster_settings=settings_for(cam, np.ones(3, np.float32), 3, DEV), context=ctx)(
                means3D=p["means3D"], means2D=m2d, shs=p["shs"], opacities=p["opacities"], scales=p["scales"], rotations=p["rotations"])
            loss = (img * ups[j][0]).sum() + (da * ups[j][1]).sum()
            total = loss if total is None else total + loss
        for v in raw.values():
            v.grad = None
        total.backward()
        torch.cuda.synchronize()
        got = {k: v.grad.clone() for k, v in raw.items()}
        if ref is None:
            ref = got
        else:
            for k in ref:
                same_bits(ref[k], got[k], f"step {step}: d/d{k} vs the eager path")
    st = list(dropin.stats().values())[0]
    assert st["replays"] >= 4, st


def test_leaf_grads_do_not_alias_a_slot(built_lib):
    """a leaf passed straight into the module (opacities [P,1], means2D) accumulates over two backward calls
    (gradient accumulation, zero_grad(set_to_none=False)): `.grad` must be the leaf's own memory -- g1 + g2 -- not a view of
    the slot's static gradient buffer, which the slot's next backward overwrites (2 * g2)."""
    from dreamscene_amd import dropin
    from dreamscene_amd.rasterizer import GaussianRasterizer, RasterContext
    dropin.reset()
    g, cams, ups = _scene(P=8_000, res=128, n_cams=1)
    params = {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in g.items()}
    s = settings_for(cams[0], np.ones(3, np.float32), 3, DEV)
    m2d = torch.zeros_like(params["means3D"], requires_grad=True)
    leaves = [params[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")] + [m2d]

    def run(ctx, scale):
        img, _, da = GaussianRasterizer(raster_settings=s, context=ctx)(
            means3D=params["means3D"], means2D=m2d, shs=params["shs"], opacities=params["opacities"], scales=params["scales"],
            rotations=params["rotations"])
        ((img * ups[0][0]).sum() * scale + (da * ups[0][1]).sum()).backward()
    for t in leaves:
        t.grad = None
    eager = RasterContext(dropin_graphs=False)
    run(eager, 1.0)
    run(eager, 3.0)
    torch.cuda.synchronize()
    ref = [t.grad.clone() for t in leaves]
    for rep in range(4):          # warm-up (eager), capture, replay, replay: the last two accumulations are both replays
        for t in leaves:
            t.grad = None
        run(None, 1.0)
        run(None, 3.0)
        torch.cuda.synchronize()
        for t, r, what in zip(leaves, ref, ("means3D", "shs", "opacities", "scales", "rotations", "means2D")):
            same_bits(t.grad, r, f"rep {rep}: accumulated .grad of {what}")
    # a third accumulation through the replayed slot, against three eager ones
    run(None, 7.0)
    torch.cuda.synchronize()
    assert st_ok(dropin)
    run_ref = [t.grad.clone() for t in leaves]
    for t in leaves:
        t.grad = None
    run(eager, 1.0); run(eager, 3.0); run(eager, 7.0)
    torch.cuda.synchronize()
    for t, r, what in zip(leaves, run_ref, ("means3D", "shs", "opacities", "scales", "rotations", "means2D")):
        same_bits(t.grad, r, f"three accumulated backwards: {what}")


def st_ok(dropin):
    st = list(dropin.stats().values())
    return len(st) == 1 and st[0]["replays"] >= 2
