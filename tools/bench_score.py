"""Forward-only paths: importance scoring (prune_list: 48 sphere cameras with score_flag, scores summed --
scene_gaussian.py:546-671, 1063-1079) through one GaussianRasterizer call per camera (the reference's loop) vs
views.importance_scores (batched); and `video_inference` (240 orbit views, training/object_trainer.py:81-118) through the
per-view module vs captured forward-only graphs. usage: python tools/bench_score.py [P] [res]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamscene_amd import synth, views, rasterizer as R
from dreamscene_amd.graph import CapturedViews
from dreamscene_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
K, D = 16, 3
g = synth.g_object(P, seed=0, K=K)
p = {k: torch.tensor(v, device=dev) for k, v in g.items()}
t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)


def settings(c, flag):
    return GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=t([1, 1, 1]),
                                         scale_modifier=1.0, viewmatrix=t(c.world_view_transform),
                                         projmatrix=t(c.full_proj_transform), sh_degree=D, campos=t(c.camera_center),
                                         prefiltered=False, score_flag=flag)


def timed(fn, reps):
    fn(); fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


res = {"P": P, "res": H}
args = dict(means3D=p["means3D"], opacities=p["opacities"], shs=p["shs"], scales=p["scales"], rotations=p["rotations"])
m2d = torch.zeros_like(p["means3D"])
sph = synth.sphere_cameras(48, H, W)
for mode in (0, 1):
    rc = R.RasterContext(score_mode=mode)
    sl = [settings(c, True) for c in sph]
    rasts = [GaussianRasterizer(raster_settings=s, context=rc) for s in sl]

    def loop():
        imp = None
        with torch.no_grad():
            for r in rasts:
                sc = r(means2D=m2d, **args)[0]
                imp = sc if imp is None else imp.add_(sc)
        return imp

    def batched():
        return views.importance_scores(sl, context=rc, **args)
    a, b = timed(loop, 3), timed(batched, 3)
    res[f"prune_list_48cams_mode{mode}"] = {"per_view_loop_ms": round(a * 1e3, 2), "importance_scores_ms": round(b * 1e3, 2),
                                             "speedup": round(a / b, 2), "views_per_s_batched": round(48 / b, 1)}

# video_inference: 240 orbit views, forward only
orbit = [synth.orbit_camera(5.35, 75.0, 360.0 * i / 240, 0.46, H, W) for i in range(240)]
sl = [settings(c, False) for c in orbit]
rasts = [GaussianRasterizer(raster_settings=s) for s in sl]


def per_view():
    with torch.no_grad():
        for r in rasts:
            r(means2D=m2d, **args)


cap = CapturedViews()
m2d4 = torch.zeros((4,) + tuple(p["means3D"].shape), device=dev)


def captured():
    with torch.no_grad():
        for i in range(0, 240, 4):
            cap(sl[i:i + 4], means3D=p["means3D"], means2D=m2d4, opacities=p["opacities"], shs=p["shs"], scales=p["scales"],
                rotations=p["rotations"])


def eager_views():
    with torch.no_grad():
        for i in range(0, 240, 4):
            views.GaussianRasterizerViews(sl[i:i + 4])(means2D=m2d4, **args)


a, b, c = timed(per_view, 2), timed(eager_views, 2), timed(captured, 2)
res["video_inference_240_views"] = {"per_view_module_views_per_s": round(240 / a, 1), "views_module_views_per_s": round(240 / b, 1),
                                    "captured_forward_only_views_per_s": round(240 / c, 1), "capture_stats": dict(cap.stats)}
print(json.dumps(res))
