"""-m gpu: consecutive GaussianRasterizer calls rotated over internal streams (rasterizer._SideStreams, RasterContext.side_streams /
GSR_SIDE_STREAMS). The reference's trainers call the module once per view (scene_gaussian.py:966-1021, loop
training/object_trainer.py:302-382); with internal streams the results must be exactly those of the same calls on the caller's
stream -- same bits for outputs and gradients -- whatever the caller does between the calls: persistent inputs (calls may
overlap), in-place edits, fresh tensors, one backward over all views or one per view, with and without the captured ring.
The forward runs on an internal stream; the backward on the caller's (autograd sees the caller's stream for the node)."""
import numpy as np
import pytest
import torch

from tests.util import same_bits, settings_for

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAMES = ("means3D", "shs", "opacities", "scales", "rotations")


def _scene(P=30_000, K=16, res=256, n_cams=4, seed=5):
    from dreamscene_amd import synth
    g = synth.g_object(P, seed=seed, K=K)
    cams = synth.object_cameras(8, res, res)[:n_cams]
    ups = [tuple(torch.tensor(x, device=DEV) for x in synth.upstream_grads(res, res, i)) for i in range(n_cams)]
    return g, cams, ups


def _forwards(params, settings, context):
    from dreamscene_amd.rasterizer import GaussianRasterizer
    outs, m2ds = [], []
    for s in settings:
        m2d = torch.zeros_like(params["means3D"], requires_grad=True)
        outs.append(GaussianRasterizer(raster_settings=s, context=context)(
            means3D=params["means3D"], means2D=m2d, shs=params["shs"], opacities=params["opacities"],
            scales=params["scales"], rotations=params["rotations"]))
        m2ds.append(m2d)
    return outs, m2ds


def _step(params, settings, ups, context, one_backward=False):
    outs, m2ds = _forwards(params, settings, context)
    leaves = [params[k] for k in NAMES]
    if one_backward:          # the trainers: ONE loss over all views, one backward
        loss = sum((o[0] * u[0]).sum() + (o[2] * u[1]).sum() for o, u in zip(outs, ups))
        gr = torch.autograd.grad(loss, leaves + m2ds)
        grads = [t.clone() for t in gr]
    else:                     # bench.py's drop-in pattern: the backwards one by one, last view first
        grads = [None] * len(outs)
        for j in reversed(range(len(outs))):
            gr = torch.autograd.grad([outs[j][0], outs[j][2]], leaves + [m2ds[j]], list(ups[j]))
            grads[j] = [t.clone() for t in gr]
    torch.cuda.synchronize()
    return [tuple(t.clone() for t in o) for o in outs], grads


@pytest.mark.parametrize("one_backward", [False, True])
def test_internal_streams_change_nothing(built_lib, monkeypatch, one_backward):
    from dreamscene_amd import dropin, rasterizer as R
    from dreamscene_amd.rasterizer import RasterContext
    # (with the captured ring switched on in the environment as well the call would raise: the two accelerators of the per-view
    #  call are mutually exclusive by construction since round 6 -- RasterContext.per_view_accel, tests/test_host_logic.py)
    monkeypatch.setattr(dropin, "ENABLED", True)
    with pytest.raises(ValueError):
        R.per_view_accel(RasterContext(side_streams=3))
    monkeypatch.setattr(dropin, "ENABLED", False)
    dropin.reset()
    g, cams, ups = _scene()
    params = {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in g.items()}
    settings = [settings_for(c, np.ones(3, np.float32), 3, DEV) for c in cams]     # persistent camera tensors
    ref_outs, ref_grads = _step(params, settings, ups, RasterContext(side_streams=0, dropin_graphs=False), one_backward)
    before = {k: dict(v) for k, v in R.side_stream_stats().items()}
    for rep in range(5):
        outs, grads = _step(params, settings, ups, RasterContext(side_streams=3), one_backward)
        for j in range(len(cams)):
            for a, b, what in zip(outs[j], ref_outs[j], ("image", "radii", "depth_alpha")):
                assert torch.equal(a, b), f"rep {rep} view {j}: {what} differs from the caller-stream path"
        if one_backward:
            for a, b, what in zip(grads, ref_grads, NAMES + tuple(f"means2D[{j}]" for j in range(len(cams)))):
                # (the sum over the views is formed by autograd in fp32: the same order on both paths)
                same_bits(a, b, f"rep {rep}: dL/d{what} (one backward over all views)")
        else:
            for j in range(len(cams)):
                for a, b, what in zip(grads[j], ref_grads[j], NAMES + ("means2D",)):
                    same_bits(a, b, f"rep {rep} view {j}: dL/d{what}")
    st = R.side_stream_stats()
    calls = sum(v["calls"] - before.get(k, {}).get("calls", 0) for k, v in st.items())
    reused = sum(v["reused_forks"] - before.get(k, {}).get("reused_forks", 0) for k, v in st.items())
    assert calls == 5 * len(cams)
    # persistent inputs: within a step every call after the first is proven unchanged and forks from the older event
    assert reused >= 5 * (len(cams) - 1) - 1, (calls, reused)
    dropin.reset()


def test_edits_and_fresh_tensors_between_calls_are_seen(built_lib):
    """What makes the early fork sound: a call whose inputs changed since the last fork event (in-place edit, new tensor, new
    camera tensor) forks from NOW. Every call is compared with the same call on the caller's stream."""
    from dreamscene_amd import rasterizer as R
    from dreamscene_amd.rasterizer import GaussianRasterizer, RasterContext
    g, cams, ups = _scene(P=20_000, res=192)
    params = {k: torch.tensor(v, device=DEV) for k, v in g.items()}
    cam_t = [settings_for(c, np.ones(3, np.float32), 3, DEV) for c in cams]
    on, off = RasterContext(side_streams=4), RasterContext(side_streams=0)

    def render(ctx, s, p):
        with torch.no_grad():
            return GaussianRasterizer(raster_settings=s, context=ctx)(
                means3D=p["means3D"], means2D=None, shs=p["shs"], opacities=p["opacities"], scales=p["scales"],
                rotations=p["rotations"])
    rng = np.random.default_rng(0)
    for it in range(24):
        s = cam_t[it % len(cam_t)]
        kind = it % 4
        if kind == 1:        # in-place edit of a parameter right before the call (enqueued on the caller's stream)
            params["means3D"].add_(torch.tensor(rng.normal(0, 0.01, size=(1, 3)).astype(np.float32), device=DEV))
        elif kind == 2:      # a fresh tensor (the trainers' activations)
            params["opacities"] = (params["opacities"] * 0.999).clamp_(0.0, 1.0)
        elif kind == 3:      # a freshly built camera (new bg tensor, written by a kernel on the caller's stream)
            s = s._replace(bg=torch.rand(3, device=DEV))
        a = render(on, s, params)
        b = render(off, s, params)
        for x, y, what in zip(a, b, ("image", "radii", "depth_alpha")):
            assert torch.equal(x, y), f"iteration {it} (kind {kind}): {what}"
    torch.cuda.synchronize()


def test_module_without_a_context_follows_the_environment(built_lib, monkeypatch):
    from dreamscene_amd import dropin, rasterizer as R
    from dreamscene_amd.rasterizer import GaussianRasterizer
    monkeypatch.setattr(dropin, "ENABLED", False)      # (with GSR_DROPIN_GRAPHS=1 exported the ring would take these calls first)
    g, cams, ups = _scene(P=10_000, res=128, n_cams=2)
    params = {k: torch.tensor(v, device=DEV) for k, v in g.items()}
    s = settings_for(cams[0], np.ones(3, np.float32), 3, DEV)
    kw = dict(means3D=params["means3D"], means2D=None, shs=params["shs"], opacities=params["opacities"], scales=params["scales"],
              rotations=params["rotations"])
    with torch.no_grad():
        ref = GaussianRasterizer(raster_settings=s)(**kw)
        n0 = sum(v["calls"] for v in R.side_stream_stats().values())
        monkeypatch.setenv("GSR_SIDE_STREAMS", "2")
        got = [GaussianRasterizer(raster_settings=s)(**kw) for _ in range(4)]
        # a stream of the caller's own: the module follows it (its internal streams are per caller stream)
        own = torch.cuda.Stream()
        own.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(own):
            got.append(GaussianRasterizer(raster_settings=s)(**kw))
        torch.cuda.current_stream().wait_stream(own)
    assert sum(v["calls"] for v in R.side_stream_stats().values()) == n0 + 5
    for o in got:
        assert torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1]) and torch.equal(o[2], ref[2])
    torch.cuda.synchronize()
