"""Forward-only throughput with and without score_flag (importance scoring, scene_gaussian.py:546-671, prune_list :1063-1079)."""
import os, sys, time, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamscene_amd import synth, rasterizer as R
from dreamscene_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda:0")
P, H, W, K, D = 500000, 1024, 1024, 16, 3
g = synth.g_object(P, seed=0, K=K); cams = synth.object_cameras(8, H, W)
p = {k: torch.tensor(v, device=dev) for k, v in g.items()}
t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
res = {}
for flag in (False, True):
    for mode in (0, 1):
        rs = [GaussianRasterizer(context=R.RasterContext(score_mode=mode), raster_settings=GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=t([1, 1, 1]),
              scale_modifier=1.0, viewmatrix=t(c.world_view_transform), projmatrix=t(c.full_proj_transform), sh_degree=D,
              campos=t(c.camera_center), prefiltered=False, score_flag=flag)) for c in cams]
        m2d = torch.zeros_like(p["means3D"])
        def one(r):
            with torch.no_grad():
                return r(means3D=p["means3D"], means2D=m2d, shs=p["shs"], opacities=p["opacities"], scales=p["scales"], rotations=p["rotations"])
        for i in range(8): one(rs[i % 8])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(40): one(rs[i % 8])
        torch.cuda.synchronize()
        res[f"score_flag={flag},mode={mode}"] = round(40 / (time.perf_counter() - t0), 1)
        if not flag: break
print(json.dumps(res))
