"""One object-training step the way the reference structures it (training/object_trainer.py:293-400), two ways:
  A) drop-in only: per-view GaussianRasterizer calls inside the reference's loop, torch.optim.Adam, the trainer's
     boolean-mask statistics updates;
  B) this repo's step: GaussianRasterizerViews (one call for the C_batch_size views, per-view noisy scales), densification
     statistics inside K8 (DensifyStats), FusedAdam;
  C) as B, but the raw leaves go straight into the kernels (scene.rasterize_models_views: exp / sigmoid / normalize /
     cat(f_dc, f_rest) and the per-view scale noise fused into K1 / K8).
Same parameters, cameras, noise and losses; reports steps/s and the time of the pieces.
usage: python tools/bench_train_step.py [--gaussians 500000] [--res 1024] [--views 4]"""
import argparse, json, math, os, sys, time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=500000)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    from dreamscene_amd import densify, synth
    from dreamscene_amd.optim import FusedAdam
    from dreamscene_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from dreamscene_amd.views import GaussianRasterizerViews
    dev = torch.device("cuda:0")
    P, H, W, V, K, D = a.gaussians, a.res, a.res, a.views, 16, 3
    g = synth.g_object(P, seed=0, K=K)
    cams = synth.object_cameras(8, H, W)
    t = lambda x: torch.tensor(np.asarray(x, dtype=np.float32), device=dev)
    sets = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=t([1, 1, 1]),
                                          scale_modifier=1.0, viewmatrix=t(c.world_view_transform),
                                          projmatrix=t(c.full_proj_transform), sh_degree=D, campos=t(c.camera_center),
                                          prefiltered=False, score_flag=False) for c in cams[:V]]
    op = np.clip(g["opacities"], 1e-4, 1 - 1e-4)
    raw0 = dict(_xyz=g["means3D"], _features_dc=g["shs"][:, :1], _features_rest=g["shs"][:, 1:],
                _opacity=np.log(op / (1 - op)), _scaling=np.log(g["scales"]), _rotation=g["rotations"])
    lrs = dict(_xyz=1.6e-4, _features_dc=2.5e-3, _features_rest=1.25e-4, _opacity=5e-2, _scaling=5e-3, _rotation=1e-3)
    targets = torch.rand((V, 3, H, W), device=dev)

    def make():
        leaves = {k: torch.tensor(np.ascontiguousarray(v, dtype=np.float32), device=dev, requires_grad=True)
                  for k, v in raw0.items()}
        groups = [{"params": [leaves[k]], "lr": lrs[k], "name": k} for k in lrs]
        return leaves, groups

    def activations(lv):
        return (torch.exp(lv["_scaling"]), torch.nn.functional.normalize(lv["_rotation"]), torch.sigmoid(lv["_opacity"]),
                torch.cat((lv["_features_dc"], lv["_features_rest"]), dim=1))

    def losses(images, depth_alphas, scales):
        loss = sum(((img - targets[k]) ** 2).mean() for k, img in enumerate(images))
        loss = loss + 1e-3 * sum(da[0].diff(dim=0).abs().mean() + da[0].diff(dim=1).abs().mean() for da in depth_alphas)
        return loss + 1e-2 * torch.mean(torch.stack(scales), dim=-1).mean()

    res = {}
    # ---------------- A: drop-in only
    lv, groups = make()
    opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    max_radii2D = torch.zeros(P, device=dev)
    grad_accum, denom = torch.zeros((P, 1), device=dev), torch.zeros((P, 1), device=dev)

    def step_a():
        images, das, scs = [], [], []
        for k in range(V):
            scales, rots, opac, shs = activations(lv)
            vsp = torch.zeros_like(lv["_xyz"], requires_grad=True) + 0
            vsp.retain_grad()
            scales = torch.clamp(scales + torch.randn_like(scales) * ((0.2 ** 0.5) * scales / 4), 0.0)
            img, radii, da = GaussianRasterizer(sets[k])(means3D=lv["_xyz"], means2D=vsp, shs=shs, opacities=opac,
                                                          scales=scales, rotations=rots)
            images.append(img); das.append(da); scs.append(scales)
        losses(images, das, scs).backward()
        vis = radii > 0
        max_radii2D[vis] = torch.max(max_radii2D[vis], radii[vis].float())
        grad_accum[vis] += torch.norm(vsp.grad[vis, :2], dim=-1, keepdim=True)
        denom[vis] += 1
        opt.step()
        opt.zero_grad(set_to_none=True)

    # ---------------- B: views + fused epilogue
    lv_b, groups_b = make()
    opt_b = FusedAdam(groups_b, lr=0.0, eps=1e-15)
    stats = densify.DensifyStats(P, dev)
    from dreamscene_amd.rasterizer import RasterContext
    rc_b = RasterContext()
    rast = GaussianRasterizerViews(sets, context=rc_b)

    def step_b():
        scales, rots, opac, shs = activations(lv_b)
        vsp = torch.zeros((V, P, 3), device=dev, requires_grad=True)
        sc = torch.clamp(scales[None] + torch.randn((V, P, 3), device=dev) * ((0.2 ** 0.5) * scales[None] / 4), 0.0)
        with stats.collect(rc_b):      # the LAST view's statistics count, like the reference's trainers
            outs = rast(means3D=lv_b["_xyz"], means2D=vsp, shs=shs, opacities=opac, scales=sc, rotations=rots)
        losses([o[0] for o in outs], [o[2] for o in outs], list(sc)).backward()
        opt_b.step(set_to_none=True)

    # ---------------- C: raw leaves straight into the views kernels (activations + noise fused)
    from dreamscene_amd import scene
    lv_c, groups_c = make()
    opt_c = FusedAdam(groups_c, lr=0.0, eps=1e-15)
    stats_c = densify.DensifyStats(P, dev)
    model_c = tuple(lv_c[k] for k in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest"))

    rc_c = scene.SceneContext()

    def step_c():
        vsp = torch.zeros((V, P, 3), device=dev, requires_grad=True)
        with stats_c.collect(rc_c):
            outs = scene.rasterize_models_views(sets, [model_c], vsp, scale_noise=torch.randn((V, P, 3), device=dev),
                                                context=rc_c)
        losses([o[0] for o in outs], [o[2] for o in outs], [o[3] for o in outs]).backward()
        opt_c.step(set_to_none=True)

    for name, fn in (("A_drop_in_only", step_a), ("B_views_fused_epilogue", step_b), ("C_raw_leaves_views", step_c)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        res[name] = {"ms_per_step": round(dt * 1e3, 3), "steps_per_s": round(1 / dt, 2), "views_per_s": round(V / dt, 1)}
    res["speedup_B_over_A"] = round(res["A_drop_in_only"]["ms_per_step"] / res["B_views_fused_epilogue"]["ms_per_step"], 3)
    res["speedup_C_over_A"] = round(res["A_drop_in_only"]["ms_per_step"] / res["C_raw_leaves_views"]["ms_per_step"], 3)
    res["workload"] = f"{P} Gaussians, K=16, {V} views @{W}x{H}, per-view scale noise, L2 + TV(depth) + scale loss, Adam, stats"
    print(json.dumps(res))


if __name__ == "__main__":
    main()
