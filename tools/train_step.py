"""One object-training step shaped like the reference's (training/object_trainer.py:293-400), three ways -- the function behind
`bench.py`'s `trainer_step` entry and `tools/bench_train_step.py`.

What a step does, in the reference's order:
  per view (C_batch_size = 4, configs/objects/sample.yaml:60): a fresh random camera; `object_render(test=False)` --
    activations of the raw leaves on FRESH tensors (exp / sigmoid / normalize / cat, gs_renderer.py:168-182), SH degree dropped to
    0 w.p. 0.1, background replaced w.p. 0.66 (configs/objects/sample.yaml:74), per-view scale noise + clamp
    (scene_gaussian.py:1005-1008), the rasterizer, then the disp post-processing WITH its boolean-mask minimum
    (scene_gaussian.py:1023-1032: `disp[alpha <= 0.1].min()` is a host synchronisation);
  loss: the guidance loss (Stable Diffusion: out of scope) stood in for by an L2 against fixed random
    targets -- the same tensors in, one scalar out --, tv_loss(images) + tv_loss(depths) (utils/system_utils.py:39-47) and the
    scale loss mean(scales) (object_trainer.py:372-375); loss.backward();
  densification statistics of the LAST view (object_trainer.py:379-384, gs_renderer.py:1061-1065);
  Adam over the six parameter groups (gs_renderer.py:615-653), zero_grad.
Legs:
  as_imported   the package exactly as a DreamScene checkout imports it: render_api.object_render (the restated glue) around
                `GaussianRasterizer`, torch.optim.Adam, the trainer's boolean-mask statistics updates;
  views_fused   `GaussianRasterizerViews` (the four views through one call), statistics inside K8, `FusedAdam`, the disp glue on
                the device (same arithmetic, the masked minimum without the host read);
  raw_leaves    the raw leaves straight into the kernels (scene.rasterize_models_views: activations and scale noise fused into K1 /
                K8), statistics inside K8, FusedAdam.
All three draw their augmentations from generators seeded alike and report ms per step."""
from __future__ import annotations

import math
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SH_DEG_AUG, BG_AUG, SHS_AUG, SCALE_AUG = 0.1, 0.66, 0.0, 1.0       # config.py:20-23, configs/objects/sample.yaml:74
LAMBDA_TV, LAMBDA_SCALE, LAMBDA_GUIDANCE = 1.0, 1.0, 0.1            # config.py:49-51, configs/objects/sample.yaml:62
LRS = dict(_xyz=1.6e-4, _features_dc=2.5e-3, _features_rest=1.25e-4, _opacity=5e-2, _scaling=5e-3, _rotation=1e-3)
LEAF_ORDER = ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")


def tv_loss(x: torch.Tensor) -> torch.Tensor:
    """utils/system_utils.py:39-47 restated (x: [B, C, H, W])."""
    b, h, w = x.size(0), x.size(2), x.size(3)
    count_h = x[:, :, 1:, :].numel() // b
    count_w = x[:, :, :, 1:].numel() // b
    h_tv = torch.pow(x[:, :, 1:, :] - x[:, :, : h - 1, :], 2).sum()
    w_tv = torch.pow(x[:, :, :, 1:] - x[:, :, :, : w - 1], 2).sum()
    return 2 * (h_tv / count_h + w_tv / count_w) / b


def disp_on_device(depth_alpha: torch.Tensor, fovx: float):
    """scene_gaussian.py:1023-1032 with the masked minimum formed without the boolean-mask gather (no host read): the minimum
    over alpha <= 0.1, or over everything when no pixel qualifies -- what the reference's try / except amounts to."""
    depth, alpha = torch.chunk(depth_alpha, 2)
    focal = 1 / (2 * math.tan(fovx / 2))
    disp = focal / (depth + (alpha * 10) + 1e-5)
    masked = torch.where(alpha <= 0.1, disp, torch.full_like(disp, float("inf"))).amin()
    min_d = torch.where(torch.isinf(masked), disp.amin(), masked)
    return torch.clamp((disp - min_d) / (disp.max() - min_d), 0.0, 1.0), alpha


def build(P: int, H: int, W: int, V: int = 4, K: int = 16, D: int = 3, dev=None, init_opacity: bool = False, seed: int = 0):
    """-> {leg name: step function(i)}, each over its own copy of the same parameters."""
    from dreamscene_amd import densify, render_api, scene, synth
    from dreamscene_amd.optim import FusedAdam
    from dreamscene_amd.rasterizer import GaussianRasterizationSettings, RasterContext
    from dreamscene_amd.views import GaussianRasterizerViews
    dev = dev or torch.device("cuda:0")
    g = synth.g_object(P, seed=seed, K=K, init_opacity=init_opacity)
    rng = np.random.default_rng(11)
    cams = [synth.orbit_camera(float(rng.uniform(5.2, 5.5)), float(rng.uniform(60.0, 90.0)), 360.0 * i / 64.0,
                               float(rng.uniform(0.32, 0.60)), H, W) for i in range(64)]
    op = np.clip(g["opacities"], 1e-4, 1 - 1e-4)
    raw0 = dict(_xyz=g["means3D"], _features_dc=g["shs"][:, :1], _features_rest=g["shs"][:, 1:],
                _opacity=np.log(op / (1 - op)).reshape(P, 1), _scaling=np.log(g["scales"]), _rotation=g["rotations"])
    targets = torch.rand((V, 3, H, W), device=dev)
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32), device=dev)
    cam_t = [(c, t(c.world_view_transform), t(c.full_proj_transform), t(c.camera_center)) for c in cams]
    white, black = t([1.0, 1.0, 1.0]), t([0.0, 0.0, 0.0])

    def make():
        leaves = {k: torch.tensor(np.ascontiguousarray(v, dtype=np.float32), device=dev, requires_grad=True)
                  for k, v in raw0.items()}
        return leaves, [{"params": [leaves[k]], "lr": LRS[k], "name": k} for k in LRS]

    def loss_of(images, depths, scales):
        images, depths = torch.stack(images, dim=0), torch.stack(depths, dim=0)
        guidance = LAMBDA_GUIDANCE * ((images - targets) ** 2).mean()            # stand-in for guidance.train_step
        loss_scale = torch.mean(torch.stack(scales, dim=0), dim=-1).mean()
        return guidance + LAMBDA_TV * (tv_loss(images) + tv_loss(depths)) + LAMBDA_SCALE * loss_scale

    def view_settings(pyrng, i):
        """The V cameras of step i and object_render's per-view random decisions (same order: SH degree, background)."""
        out = []
        for j in range(V):
            c, vm, pm, cp = cam_t[(V * i + j) % 64]
            sh = 0 if pyrng.random() < SH_DEG_AUG else D
            bg = white
            if pyrng.random() < BG_AUG:
                bg = torch.rand(3, device=dev) if pyrng.random() < 0.5 else black
            out.append((c, GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=bg,
                                                         scale_modifier=1.0, viewmatrix=vm, projmatrix=pm, sh_degree=sh,
                                                         campos=cp, prefiltered=False, score_flag=False)))
        return out

    legs = {}

    # ---------------------------------------------------------------- as imported
    lv_a, groups_a = make()
    params_a = render_api.GaussianParams(lv_a["_xyz"], lv_a["_scaling"], lv_a["_rotation"], lv_a["_opacity"], lv_a["_features_dc"],
                                         lv_a["_features_rest"], D)
    opt_a = torch.optim.Adam(groups_a, lr=0.0, eps=1e-15)
    max_radii2D = torch.zeros(P, device=dev)
    grad_accum, denom = torch.zeros((P, 1), device=dev), torch.zeros((P, 1), device=dev)
    rng_a = random.Random(5)

    def step_as_imported(i):
        images, depths, scales = [], [], []
        for j in range(V):
            cam = cams[(V * i + j) % 64]
            out = render_api.object_render(params_a, cam, white, test=False, sh_deg_aug_ratio=SH_DEG_AUG, bg_aug_ratio=BG_AUG,
                                           shs_aug_ratio=SHS_AUG, scale_aug_ratio=SCALE_AUG, rng=rng_a)
            images.append(out["image"]); depths.append(out["depth"]); scales.append(out["scales"])
        loss_of(images, depths, scales).backward()
        vis, radii, vsp = out["visibility_filter"], out["radii"], out["viewspace_points"]
        max_radii2D[vis] = torch.max(max_radii2D[vis], radii[vis].float())
        grad_accum[vis] += torch.norm(vsp.grad[vis, :2], dim=-1, keepdim=True)
        denom[vis] += 1
        opt_a.step()
        opt_a.zero_grad(set_to_none=True)
    legs["as_imported"] = step_as_imported

    # ---------------------------------------------------------------- views + fused epilogue
    lv_b, groups_b = make()
    opt_b = FusedAdam(groups_b, lr=0.0, eps=1e-15)
    stats_b = densify.DensifyStats(P, dev)
    rc_b = RasterContext()
    rng_b = random.Random(5)

    def step_views_fused(i):
        vs = view_settings(rng_b, i)
        scales = torch.exp(lv_b["_scaling"])
        rots = torch.nn.functional.normalize(lv_b["_rotation"])
        opac = torch.sigmoid(lv_b["_opacity"])
        shs = torch.cat((lv_b["_features_dc"], lv_b["_features_rest"]), dim=1)
        vsp = torch.zeros((V, P, 3), device=dev, requires_grad=True)
        sc = torch.clamp(scales[None] + torch.randn((V, P, 3), device=dev) * ((0.2 ** 0.5) * scales[None] / 4), 0.0)
        with stats_b.collect(rc_b):                      # the LAST view's statistics count, like the reference's trainers
            outs = GaussianRasterizerViews([s for _, s in vs], context=rc_b)(
                means3D=lv_b["_xyz"], means2D=vsp, shs=shs, opacities=opac, scales=sc, rotations=rots)
        depths = [disp_on_device(o[2], c.FoVx)[0] for (c, _), o in zip(vs, outs)]
        loss_of([o[0] for o in outs], depths, list(sc)).backward()
        opt_b.step(set_to_none=True)
    legs["views_fused"] = step_views_fused

    # ---------------------------------------------------------------- raw leaves straight into the kernels
    lv_c, groups_c = make()
    opt_c = FusedAdam(groups_c, lr=0.0, eps=1e-15)
    stats_c = densify.DensifyStats(P, dev)
    model_c = tuple(lv_c[k] for k in LEAF_ORDER)
    rc_c = scene.SceneContext()
    rng_c = random.Random(5)

    def step_raw_leaves(i):
        vs = view_settings(rng_c, i)
        vsp = torch.zeros((V, P, 3), device=dev, requires_grad=True)
        with stats_c.collect(rc_c):
            outs = scene.rasterize_models_views([s for _, s in vs], [model_c], vsp,
                                                scale_noise=torch.randn((V, P, 3), device=dev), context=rc_c)
        depths = [disp_on_device(o[2], c.FoVx)[0] for (c, _), o in zip(vs, outs)]
        loss_of([o[0] for o in outs], depths, [o[3] for o in outs]).backward()
        opt_c.step(set_to_none=True)
    legs["raw_leaves"] = step_raw_leaves
    return legs


def measure(P: int, H: int, W: int, V: int = 4, K: int = 16, D: int = 3, dev=None, init_opacity: bool = False,
            seconds: float = 1.5, warmup: int = 5, legs=("as_imported", "views_fused", "raw_leaves")) -> dict:
    """ms per step of every leg (the same loop bench.py's other `*_seconds` entries use: warm-up, then whole chunks of steps for
    >= `seconds`, one device synchronisation at the end)."""
    dev = dev or torch.device("cuda:0")
    fns = build(P, H, W, V, K, D, dev, init_opacity)
    res = {}
    for name in legs:
        fn = fns[name]
        try:
            for i in range(warmup):
                fn(i)
            torch.cuda.synchronize(dev)
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < seconds:
                for _ in range(4):
                    fn(warmup + n)
                    n += 1
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / n
            res[name] = {"ms_per_step": round(dt * 1e3, 3), "views_per_s": round(V / dt, 1), "steps": n}
        except Exception as e:           # (a leg that fails must not take the bench line with it)
            res[name] = {"error": repr(e)[:300]}
            torch.cuda.synchronize(dev)
    return res
