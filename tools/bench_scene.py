"""Fused multi-model path (dreamscene_amd/scene.py) vs the reference's glue (per-model activations + torch.cat) in
front of the drop-in rasterizer, fwd+bwd per view, gradients accumulated into the leaves' .grad as a trainer does.
usage: python tools/bench_scene.py [--models 5] [--per-model 400000] [--res 1024] [--K 4] [--noise]"""
import argparse, json, os, sys, time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", type=int, default=5)
    ap.add_argument("--per-model", type=int, default=400000)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--K", type=int, default=4)
    ap.add_argument("--views", type=int, default=40)
    ap.add_argument("--noise", action="store_true")
    ap.add_argument("--scene", default="indoor", choices=["indoor", "object"])
    a = ap.parse_args()
    from dreamscene_amd import scene, synth
    from dreamscene_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    H = W = a.res
    K, D = a.K, {1: 0, 4: 1, 9: 2, 16: 3}[a.K]
    M, n = a.models, a.per_model
    if a.scene == "indoor":
        g = synth.g_indoor(seed=0, per_wall=max(1, (M * n) // 5), K=K)
        cams = synth.indoor_cameras(4, H, W)
    else:
        g = synth.g_object(M * n, seed=0, K=K)
        cams = synth.object_cameras(4, H, W)
    P = g["means3D"].shape[0]
    cuts = [P * m // M for m in range(M + 1)]
    op = np.clip(g["opacities"], 1e-4, 1 - 1e-4)
    raw = (g["means3D"], np.log(np.maximum(g["scales"], 1e-12)), g["rotations"] * 1.7, np.log(op / (1 - op)),
           g["shs"][:, :1, :], g["shs"][:, 1:, :])
    models = [tuple(torch.tensor(np.ascontiguousarray(x[cuts[m]:cuts[m + 1]], dtype=np.float32), device=dev,
                                 requires_grad=True) for x in raw) for m in range(M)]
    leaves = [t for m in models for t in m]
    gi_np, gda_np = synth.upstream_grads(H, W, seed=0)
    gi, gda = torch.tensor(gi_np, device=dev), torch.tensor(gda_np, device=dev)
    t = lambda v: torch.tensor(np.asarray(v, dtype=np.float32), device=dev)
    sets = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy,
                                          bg=t([1, 1, 1]), scale_modifier=1.0, viewmatrix=t(c.world_view_transform),
                                          projmatrix=t(c.full_proj_transform), sh_degree=D, campos=t(c.camera_center),
                                          prefiltered=False, score_flag=False) for c in cams]

    def unfused(s):
        xyz = torch.cat([m[0] for m in models])
        m2d = torch.zeros_like(xyz, requires_grad=True) + 0
        opac = torch.cat([torch.sigmoid(m[3]) for m in models])
        scales = torch.cat([torch.exp(m[1]) for m in models])
        rots = torch.cat([torch.nn.functional.normalize(m[2]) for m in models])
        shs = torch.cat([torch.cat((m[4], m[5]), dim=1) for m in models])
        if a.noise:
            shs = shs + torch.randn_like(shs) * ((0.2 ** 0.5) * shs)
            scales = torch.clamp(scales + torch.randn_like(scales) * ((0.2 ** 0.5) * scales / 4), 0.0)
        img, radii, da = GaussianRasterizer(s)(means3D=xyz, means2D=m2d, shs=shs, opacities=opac, scales=scales,
                                                rotations=rots)
        ((img * gi).sum() + (da * gda).sum() + 0.01 * scales.mean()).backward()

    def fused(s):
        m2d = torch.zeros((P, 3), device=dev, requires_grad=True) + 0
        sn = torch.randn((P, 3), device=dev) if a.noise else None
        hn = torch.randn((P, K, 3), device=dev) if a.noise else None
        img, radii, da, scales = scene.rasterize_models(s, models, m2d, sn, hn)
        ((img * gi).sum() + (da * gda).sum() + 0.01 * scales.mean()).backward()

    V = 4

    def fused_views(_s):
        m2d = torch.zeros((V, P, 3), device=dev, requires_grad=True)
        sn = torch.randn((V, P, 3), device=dev) if a.noise else None
        hn = torch.randn((V, P, K, 3), device=dev) if a.noise else None
        outs = scene.rasterize_models_views(sets[:V], models, m2d, sn, hn)
        sum((img * gi).sum() + (da * gda).sum() + 0.01 * sc.mean() for img, _, da, sc in outs).backward()

    res = {}
    for name, fn in (("unfused", unfused), ("fused", fused), ("fused_views", fused_views)):
        for x in leaves:
            x.grad = None
        for i in range(8):
            fn(sets[i % len(sets)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.views):
            fn(sets[i % len(sets)])
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / a.views * 1e3 / (V if name == "fused_views" else 1)
    print(json.dumps(dict(workload=f"{a.scene}: {M} models x {P // M} Gaussians, K={K}, {W}x{H}, noise={a.noise}",
                          ms_per_view_unfused=round(res["unfused"], 3), ms_per_view_fused=round(res["fused"], 3),
                          ms_per_view_fused_4_views_per_call=round(res["fused_views"], 3),
                          speedup=round(res["unfused"] / res["fused"], 3))))


if __name__ == "__main__":
    main()
