"""Gather-free multi-model rendering with the activations fused into the rasterizer.

`SceneGaussian.scene_render` (scene_gaussian.py:673-893) builds the rasterizer inputs of a view by running the
activations of every visible GaussianModel (gs_renderer.py:464-488: exp / normalize / sigmoid / cat(f_dc, f_rest)),
`torch.cat`-ing the results of all models (:753-843) and, when training, adding noise to the SH coefficients and the
scales (:844-852) -- about forty elementwise / copy kernels and ~1 GB of HBM traffic per view at 2.3 M Gaussians,
as much as the rasterizer itself. Here the RAW leaf tensors go to the HIP kernels as a table of models
(include/gsrast.h, GsrScene): K1 applies the activations while it reads, culled Gaussians never touch their SH
rows, and K8 writes the gradients of the raw leaves directly (no cat / split / activation backward kernels).

    image, radii, depth_alpha, scales = rasterize_models(settings, models, means2D, scale_noise=None, sh_noise=None)

`models` is a list of objects with `_xyz, _scaling, _rotation, _opacity, _features_dc, _features_rest` (GaussianModel's
own attribute names) or of 6-tuples in that order. Index i of every per-Gaussian result (radii, means2D.grad, scales)
is the index torch.cat over the models would give. `scene_render` below restates the reference's glue on top of it.
"""
from __future__ import annotations

import dataclasses
import math
import random
from typing import List, Optional, Sequence

import torch

from . import rasterizer as R

LEAVES = ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")


@dataclasses.dataclass
class SceneContext(R.RasterContext):
    """RasterContext + model_grad_buffers: optional per-model tuples of 6 tensors the backward ADDS this call's gradients
    to (device-side accumulation over the views of one optimizer step, the scene path's counterpart of grad_arena); the
    autograd outputs are then None for the leaves."""
    model_grad_buffers: Optional[List[tuple]] = None

    def snapshot(self) -> "SceneContext":
        return dataclasses.replace(self)


def _scene_rc(context) -> SceneContext:
    if context is None:
        return SceneContext()
    if isinstance(context, SceneContext):
        return context.snapshot()
    return SceneContext(**{f.name: getattr(context, f.name) for f in dataclasses.fields(R.RasterContext)})


def _leaves(model) -> tuple:
    if isinstance(model, (tuple, list)):
        if len(model) != 6:
            raise ValueError("a model tuple is (xyz, scaling, rotation, opacity, features_dc, features_rest)")
        return tuple(model)
    return tuple(getattr(model, n) for n in LEAVES)


class _RasterizeModels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, settings, rc, scale_noise, sh_noise, means2D, *leaves):
        models = [tuple(leaves[6 * m:6 * m + 6]) for m in range(len(leaves) // 6)]
        out, st = R.rasterize_forward_raw(settings, None, None, None, None, None, None, None, want_aux=False,
                                          scene=dict(models=models, scale_noise=scale_noise, sh_noise=sh_noise), rc=rc)
        ctx.st, ctx.rc = st, rc
        ctx.n_leaves = len(leaves)
        ctx.set_materialize_grads(False)       # no zero tensors for outputs nobody differentiates (radii is [P] int32)
        ctx.mark_non_differentiable(out["radii"])
        if settings.score_flag:
            ctx.mark_non_differentiable(out["score"])
            return out["score"], out["color"], out["radii"], out["depth_alpha"], out["act_scales"]
        return out["color"], out["radii"], out["depth_alpha"], out["act_scales"]

    @staticmethod
    def backward(ctx, *grads):
        st = ctx.st
        g_color, _, g_da, g_scales = grads[-4:]
        H, W = st.view.image_height, st.view.image_width
        if g_color is None:
            g_color = torch.zeros((3, H, W), dtype=torch.float32, device=st.dev)
        if g_da is None:
            g_da = torch.zeros((2, H, W), dtype=torch.float32, device=st.dev)
        rc = ctx.rc
        bufs = rc.model_grad_buffers
        o = R.rasterize_backward_raw(st, g_color, g_da, model_grads=bufs, accumulate=bufs is not None,
                                     dL_dscales_out=g_scales, stats=rc.densify_stats, profile=rc.profile)
        flat = []
        for row in o["model_grads"]:
            flat.extend([None] * 6 if bufs is not None else row)
        return (None, None, None, None, o["dL_dmeans2D"], *flat)


def rasterize_models(settings, models: Sequence, means2D: torch.Tensor, scale_noise: Optional[torch.Tensor] = None,
                     sh_noise: Optional[torch.Tensor] = None, context=None):
    """One view of several GaussianModels through the fused path. Returns what GaussianRasterizer returns, plus the
    activated (and augmented) scales [P,3] (differentiable: the trainers put a loss on them).
    context: a RasterContext / SceneContext (optional)."""
    flat = []
    for m in models:
        flat.extend(_leaves(m))
    return _RasterizeModels.apply(settings, _scene_rc(context), scale_noise, sh_noise, means2D, *flat)


class _RasterizeModelsViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, settings_list, rc, scale_noise, sh_noise, means2D, *leaves):
        from .views import rasterize_views_forward_raw
        V = len(settings_list)
        models = [tuple(leaves[6 * m:6 * m + 6]) for m in range(len(leaves) // 6)]
        scenes = [dict(models=models, scale_noise=None if scale_noise is None else scale_noise[k],
                       sh_noise=None if sh_noise is None else sh_noise[k]) for k in range(V)]
        res = rasterize_views_forward_raw(settings_list, None, None, None, None, None, None, None, scenes=scenes, rc=rc)
        ctx.states, ctx.rc = [st for _, st in res], rc
        ctx.set_materialize_grads(False)
        outs = []
        for o, _ in res:
            ctx.mark_non_differentiable(o["radii"])
            outs += [o["color"], o["radii"], o["depth_alpha"], o["act_scales"]]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        sts = ctx.states
        V = len(sts)
        H, W, dev = sts[0].view.image_height, sts[0].view.image_width, sts[0].dev
        z = lambda c: torch.zeros((c, H, W), dtype=torch.float32, device=dev)
        gcs = [grads[4 * k] if grads[4 * k] is not None else z(3) for k in range(V)]
        gdas = [grads[4 * k + 2] if grads[4 * k + 2] is not None else z(2) for k in range(V)]
        gss = [grads[4 * k + 3] for k in range(V)]
        rc = ctx.rc
        bufs = rc.model_grad_buffers
        o = R.rasterize_backward_views_scene_raw(sts, gcs, gdas, model_grads=bufs, accumulate=bufs is not None,
                                                 dL_dscales_outs=gss if any(g is not None for g in gss) else None,
                                                 stats=rc.densify_stats, stats_views=rc.stats_views, profile=rc.profile)
        flat = []
        for row in o["model_grads"]:
            flat.extend([None] * 6 if bufs is not None else row)
        return (None, None, None, None, o["dL_dmeans2D"], *flat)


def rasterize_models_views(settings_list, models: Sequence, means2D: torch.Tensor,
                           scale_noise: Optional[torch.Tensor] = None, sh_noise: Optional[torch.Tensor] = None,
                           context=None):
    """The views of one optimizer step of several GaussianModels in one call (raw leaves, activations and per-view noise
    fused: scale_noise [V,P,3], sh_noise [V,P,K,3] N(0,1) samples or None). means2D: [V,P,3] zeros. Returns a list of
    (image, radii, depth_alpha, scales) per view; the parameter gradients are the sums over the views."""
    flat = []
    for m in models:
        flat.extend(_leaves(m))
    V = len(settings_list)
    from .views import _uniform
    if not _uniform(list(settings_list)):
        raise ValueError("rasterize_models_views: all views of a call must have the same image size and scale_modifier")
    out = _RasterizeModelsViews.apply(tuple(settings_list), _scene_rc(context), scale_noise, sh_noise, means2D, *flat)
    return [tuple(out[4 * k:4 * k + 4]) for k in range(V)]


def scene_render(models: Sequence, camera, bg_color: torch.Tensor, active_sh_degree: int,
                 scaling_modifier: float = 1.0, black_video: bool = False, sh_deg_aug_ratio: float = 0.1,
                 bg_aug_ratio: float = 0.3, shs_aug_ratio: float = 1.0, scale_aug_ratio: float = 1.0,
                 test: bool = False, no_grad: bool = False, rng: random.Random = random):
    """SceneGaussian.scene_render (scene_gaussian.py:673-893) over the fused path: same random augmentation decisions
    in the same order, same output dict. The noise samples are drawn with torch.randn in the concatenated index space
    (the reference draws them with randn_like on the concatenated tensors)."""
    from .rasterizer import GaussianRasterizationSettings
    first = _leaves(models[0])[0]
    dev = first.device
    P = sum(int(_leaves(m)[0].shape[0]) for m in models)
    K = 1 + int(_leaves(models[0])[5].shape[1])
    screenspace_points = torch.zeros((P, 3), dtype=first.dtype, requires_grad=True, device=dev) + 0
    if not no_grad:
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
    if black_video:
        bg_color = torch.zeros_like(bg_color)
    act_SH = 0 if (rng.random() < sh_deg_aug_ratio and not test) else active_sh_degree
    if rng.random() < bg_aug_ratio and not test:
        bg_color = torch.rand_like(bg_color) if rng.random() < 0.5 else torch.zeros_like(bg_color)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
    settings = GaussianRasterizationSettings(
        image_height=int(camera.image_height), image_width=int(camera.image_width),
        tanfovx=math.tan(camera.FoVx * 0.5), tanfovy=math.tan(camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=t(camera.world_view_transform),
        projmatrix=t(camera.full_proj_transform), sh_degree=act_SH, campos=t(camera.camera_center),
        prefiltered=False, score_flag=False)
    sh_noise = torch.randn((P, K, 3), dtype=torch.float32, device=dev) if (rng.random() < shs_aug_ratio and not test) else None
    scale_noise = torch.randn((P, 3), dtype=torch.float32, device=dev) if (rng.random() < scale_aug_ratio and not test) else None
    rendered_image, radii, depth_alpha, scales = rasterize_models(settings, models, screenspace_points, scale_noise,
                                                                  sh_noise)
    depth, alpha = torch.chunk(depth_alpha, 2)
    focal = 1 / (2 * math.tan(camera.FoVx / 2))
    disp = focal / (depth + (alpha * 10) + 1e-5)
    try:
        min_d = disp[alpha <= 0.1].min()
    except Exception:
        min_d = disp.min()
    disp = torch.clamp((disp - min_d) / (disp.max() - min_d), 0.0, 1.0)
    return {"image": rendered_image, "depth": disp, "alpha": alpha, "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0, "radii": radii, "scales": scales}
