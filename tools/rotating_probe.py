"""The `rotating_cameras` leg of bench.py on its own, through ONE of its two paths, for kernel traces:
    python tools/rotating_probe.py eager|captured [steps]
64 cameras sampled like the reference's random cameras (radius 5.2-5.5, polar 60-90 deg, FoV 0.32-0.60: config.py:88-99), 4 new ones per
step, fused Adam on the arena's sums in between. Prints views/s and, for the captured path, its statistics."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamscene_amd import _lib, multiview, synth  # noqa: E402
from dreamscene_amd.graph import CapturedViews  # noqa: E402
from dreamscene_amd.optim import FusedAdam  # noqa: E402
from dreamscene_amd.rasterizer import GaussianRasterizationSettings, RasterContext  # noqa: E402
from dreamscene_amd.views import GaussianRasterizerViews  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else "eager"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
P, H, W, K, D, V = 500_000, 1024, 1024, 16, 3, 4
dev = torch.device("cuda", 0)
_lib.load()
g = synth.g_object(P, seed=0, K=K)
params = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in g.items()}
gi, gda = (torch.tensor(x, device=dev) for x in synth.upstream_grads(H, W, seed=0))
t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
rng = np.random.default_rng(7)
cams = [synth.orbit_camera(float(rng.uniform(5.2, 5.5)), float(rng.uniform(60.0, 90.0)), 360.0 * i / 64.0,
                           float(rng.uniform(0.32, 0.60)), H, W) for i in range(64)]
sl_r = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=t([1, 1, 1]),
                                      scale_modifier=1.0, viewmatrix=t(c.world_view_transform), projmatrix=t(c.full_proj_transform),
                                      sh_degree=D, campos=t(c.camera_center), prefiltered=False, score_flag=False) for c in cams]
arena = multiview.GradArena(P, K, dev)
ctx = RasterContext(grad_arena=arena)
names = ("means3D", "scales", "rotations", "opacities", "shs")
opt = FusedAdam([params[n] for n in names], lr=2e-5, eps=1e-15)
arena_grads = [arena.views[n].view(params[n].shape) for n in names]
cap = CapturedViews(context=ctx) if path == "captured" else None
if path == "both":          # bench.py's order: the eager path first, then a CapturedViews created afterwards, in one process
    cap = None


def step(i):
    sl = [sl_r[(V * i + j) % 64] for j in range(V)]
    m2d = torch.zeros((V, P, 3), device=dev, requires_grad=True)
    if cap is not None:
        outs = cap(sl, means3D=params["means3D"], means2D=m2d, opacities=params["opacities"], shs=params["shs"],
                   scales=params["scales"], rotations=params["rotations"])
    else:
        outs = GaussianRasterizerViews(sl, context=ctx)(means3D=params["means3D"], means2D=m2d, shs=params["shs"],
                                                        opacities=params["opacities"], scales=params["scales"],
                                                        rotations=params["rotations"])
    torch.autograd.grad([x for (img, _, da) in outs for x in (img, da)], [m2d], [gi, gda] * V)
    opt.step(grads=arena_grads)


for i in range(8):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step(8 + i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{path}: {steps * V / dt:.1f} views/s, {dt / steps * 1e3:.3f} ms per step", cap.stats if cap is not None else "")
if path == "both":
    cap = CapturedViews(context=ctx)
    for i in range(8):
        step(i)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(steps):
            step(8 + i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"then captured (rep {rep}): {steps * V / dt:.1f} views/s, {dt / steps * 1e3:.3f} ms per step", cap.stats)
