"""The reference's OWN interface (one GaussianRasterizer call per view) under the module's switches, A/B in one process:
internal streams (RasterContext.side_streams) x captured ring (RasterContext.dropin_graphs), four call patterns:
  fb4     four forwards, then four backwards, one torch.autograd.grad per view (bench.py's `dropin_views_per_s`)
  one_bw  four forwards, ONE backward over the summed losses (the trainers: object_trainer.py:302-382)
  per     forward + backward of a view right after each other
  fwd     forward only under no_grad (video_inference, the importance-score loop)
Prints views/s and the host's enqueue time per view (the time after which the host is done and only waits for the GPU).
usage: python tools/bench_dropin.py [--gaussians P] [--res R] [--seconds S] [--streams 0,2,4] [--graphs 0,1] [--patterns fb4,fwd]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamscene_amd import _lib, dropin, rasterizer as R, synth  # noqa: E402
from dreamscene_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, RasterContext  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=500_000)
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--seconds", type=float, default=1.0)
ap.add_argument("--streams", default="0,2,4")
ap.add_argument("--graphs", default="0,1")
ap.add_argument("--patterns", default="fb4,one_bw,per,fwd")
ap.add_argument("--init-opacity", action="store_true")
args = ap.parse_args()
P, res = args.gaussians, args.res
V, K, D = 4, 16, 3
dev = torch.device("cuda", 0)
_lib.load()
g = synth.g_object(P, seed=0, K=K, init_opacity=args.init_opacity)
cams = synth.object_cameras(8, res, res)[:V]
params = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in g.items()}
gi_np, gda_np = synth.upstream_grads(res, res, seed=0)
gi, gda = torch.tensor(gi_np, device=dev), torch.tensor(gda_np, device=dev)
t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
sets = [GaussianRasterizationSettings(image_height=res, image_width=res, tanfovx=c.tanfovx, tanfovy=c.tanfovy,
                                      bg=t([1.0, 1.0, 1.0]), scale_modifier=1.0, viewmatrix=t(c.world_view_transform),
                                      projmatrix=t(c.full_proj_transform), sh_degree=D, campos=t(c.camera_center),
                                      prefiltered=False, score_flag=False) for c in cams]
leaves = [params[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]
kw = lambda m2d: dict(means3D=params["means3D"], means2D=m2d, shs=params["shs"], opacities=params["opacities"],
                      scales=params["scales"], rotations=params["rotations"])


def make(ctx):
    rasts = [GaussianRasterizer(raster_settings=s, context=ctx) for s in sets]

    def fb4():
        outs = []
        for r in rasts:
            m2d = torch.zeros_like(params["means3D"], requires_grad=True)
            img, _, da = r(**kw(m2d))
            outs.append((img, da, m2d))
        for img, da, m2d in reversed(outs):
            torch.autograd.grad([img, da], leaves + [m2d], [gi, gda])

    def one_bw():
        outs, m2ds = [], []
        for r in rasts:
            m2d = torch.zeros_like(params["means3D"], requires_grad=True)
            img, _, da = r(**kw(m2d))
            outs += [img, da]
            m2ds.append(m2d)
        torch.autograd.grad(outs, leaves + m2ds, [gi, gda] * V)

    def per():
        for r in rasts:
            m2d = torch.zeros_like(params["means3D"], requires_grad=True)
            img, _, da = r(**kw(m2d))
            torch.autograd.grad([img, da], leaves + [m2d], [gi, gda])

    def fwd():
        with torch.no_grad():
            for r in rasts:
                r(**kw(None))
    return dict(fb4=fb4, one_bw=one_bw, per=per, fwd=fwd)


def timed(fn, seconds):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    n, enq = 0, 0.0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(5):
            fn()
        n += 5
    enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return n * V / dt, 1e6 * enq / (n * V)


rows = []
for gr in [int(x) for x in args.graphs.split(",")]:
    for ns in [int(x) for x in args.streams.split(",")]:
        if gr and ns > 1:
            continue                 # the two accelerators are mutually exclusive (RasterContext.per_view_accel)
        dropin.reset()
        ctx = RasterContext(side_streams=ns, dropin_graphs=bool(gr))
        fns = make(ctx)
        for pat in args.patterns.split(","):
            try:
                vps, enq_us = timed(fns[pat], args.seconds)
                row = dict(P=P, res=res, graphs=gr, streams=ns, pattern=pat, views_per_s=round(vps, 1),
                           host_enqueue_us_per_view=round(enq_us, 1))
            except Exception as e:          # a combination that fails must not take the others with it
                row = dict(P=P, res=res, graphs=gr, streams=ns, pattern=pat, error=repr(e)[:300])
                torch.cuda.synchronize()
            rows.append(row)
            print(json.dumps(row), flush=True)
print("SIDE", json.dumps(R.side_stream_stats()))
print("RINGS", json.dumps(dropin.stats()))
