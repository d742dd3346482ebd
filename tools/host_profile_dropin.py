"""Host-side cost of the reference's own interface (one GaussianRasterizer call per view: four forwards, then four backwards),
cProfile over the steady state. Usage: python tools/host_profile_dropin.py [P] [res]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamscene_amd import _lib, rasterizer as R, synth  # noqa: E402
from dreamscene_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402

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
hs = R.HostStats()
rasts = [GaussianRasterizer(raster_settings=s, context=R.RasterContext(host_stats=hs)) for s in sets]
leaves = [params[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]


def step():
    outs = []
    for rast in rasts:
        m2d = torch.zeros_like(params["means3D"], requires_grad=True)
        img, radii, da = rast(means3D=params["means3D"], means2D=m2d, shs=params["shs"], opacities=params["opacities"],
                              scales=params["scales"], rotations=params["rotations"])
        outs.append((img, da, m2d))
    for img, da, m2d in reversed(outs):
        torch.autograd.grad([img, da], leaves + [m2d], [gi, gda])


for _ in range(20):
    step()
torch.cuda.synchronize()
hs.wait_s = 0.0
N = 200
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"P={P} res={res}: enqueue {1e3 * (t1 - t0) / N / V:.3f} ms/view of which waiting for the pair count "
      f"{1e3 * hs.wait_s / N / V:.3f}; wall {1e3 * (t2 - t0) / N / V:.3f} ms/view = {N * V / (t2 - t0):.0f} views/s")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
