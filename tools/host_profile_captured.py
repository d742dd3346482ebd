"""Host-side cost of a CAPTURED step (graph.CapturedViews), cProfile over the steady state.
Usage: python tools/host_profile_captured.py [P] [res]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamscene_amd import _lib, multiview, rasterizer as R, synth  # noqa: E402
from dreamscene_amd.graph import CapturedViews  # noqa: E402
from dreamscene_amd.rasterizer import GaussianRasterizationSettings  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
res = int(sys.argv[2]) if len(sys.argv) > 2 else 512
V, K, D = 4, 16, 3
dev = torch.device("cuda", 0)
_lib.load()
g = synth.g_object(P, seed=0, K=K)
cams = synth.object_cameras(8, res, res)[:V]
params = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in g.items()}
gi_np, gda_np = synth.upstream_grads(res, res, seed=0)
gi, gda = torch.tensor(gi_np, device=dev), torch.tensor(gda_np, device=dev)
t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
sets = [GaussianRasterizationSettings(image_height=res, image_width=res, tanfovx=c.tanfovx, tanfovy=c.tanfovy,
                                      bg=t([1.0, 1.0, 1.0]), scale_modifier=1.0, viewmatrix=t(c.world_view_transform),
                                      projmatrix=t(c.full_proj_transform), sh_degree=D, campos=t(c.camera_center),
                                      prefiltered=False, score_flag=False) for c in cams]
arena = multiview.GradArena(P, K, dev)
hs = R.HostStats()
rast = CapturedViews(context=R.RasterContext(grad_arena=arena, host_stats=hs))


def step():
    means2D = torch.zeros((V,) + tuple(params["means3D"].shape), device=dev, requires_grad=True)
    outs = rast(sets, means3D=params["means3D"], means2D=means2D, opacities=params["opacities"], shs=params["shs"],
                scales=params["scales"], rotations=params["rotations"])
    torch.autograd.grad([x for (img, _, da) in outs for x in (img, da)], [means2D], [gi, gda] * V)


for _ in range(30):
    step()
torch.cuda.synchronize()
hs.wait_s = 0.0
t0 = time.perf_counter()
for _ in range(300):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / 300:.3f} ms/step of which waiting {1e3 * hs.wait_s / 300:.3f}; wall {1e3 * (t2 - t0) / 300:.3f} ms/step; {rast.stats}")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(25)
