"""Several views of the SAME Gaussians in one call (the C_batch_size views of an optimizer step,
training/object_trainer.py:302-382; new functionality, no reference counterpart).

The reference renders its 4 views per step one after the other. Everything between K1 and the render of a view is a
chain of ~20 small, launch-latency-bound kernels (depth sort, column counts); `gsr_forward_project_batch` pushes the
chains of all views through each launch together (blockIdx.y = view). The per-view results are exactly those of
`GaussianRasterizer` (same kernels, same order of operations inside a view); the backward runs per view and sums the
parameter gradients on the device (K8 accumulate mode).

    rast = GaussianRasterizerViews([settings_0, ..., settings_3])
    outs = rast(means3D, means2D, opacities, shs=shs, scales=scales, rotations=rotations)   # means2D: [V,P,3] zeros
    # scales: [P,3] shared, or [V,P,3] = every view its own (the trainers add fresh noise to the scales of every view)
    (image_k, radii_k, depth_alpha_k) = outs[k]
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import torch

from . import _lib as L
from . import rasterizer as R

MAX_VIEWS = 16


def _uniform(settings_list) -> bool:
    s0 = settings_list[0]
    return all(int(s.image_height) == int(s0.image_height) and int(s.image_width) == int(s0.image_width) and
               float(s.scale_modifier) == float(s0.scale_modifier) for s in settings_list)


def rasterize_views_forward_raw(settings_list: Sequence, means3D, opacities, shs, colors_precomp, scales, rotations,
                                cov3D_precomp, want_aux: bool = False, scenes=None, rc=None, score_sum=None):
    """Forward of V views. Returns [(outputs, state)] like rasterize_forward_raw per view.
    scales: [P,3] shared by the views, or [V,P,3] (every view its own, e.g. with the trainers' per-view scale noise).
    scenes: instead of the tensors, one `scene` dict per view (rasterize_forward_raw): the same models' raw leaves,
    per-view noise samples.
    score_flag views (all of them or none): every view's outputs carry its own `score` [P]; with score_sum = a [P] fp32
    tensor the scores of ALL views are ADDED to it instead (K6's per-splat atomics of every view land in the one buffer:
    the `imp_list` of the reference's prune_list, scene_gaussian.py:1063-1079) and the views' `score` entries alias it."""
    lib = L.load()
    rc = rc or R.DEFAULT_CONTEXT
    V = len(settings_list)
    if scenes is not None:
        return _views_forward_scene(lib, settings_list, scenes, want_aux, rc)
    per_view = scales is not None and scales.dim() == 3
    if per_view and scales.shape[0] != V:
        raise ValueError(f"per-view scales must be [V,P,3] with V = {V}")
    if per_view:
        scales = scales.contiguous()
    sc = (lambda k: scales[k]) if per_view else (lambda k: scales)
    s0 = settings_list[0]
    dev = means3D.device
    P, H, W = int(means3D.shape[0]), int(s0.image_height), int(s0.image_width)
    same = _uniform(settings_list)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ws = R._workspace(dev, stream)
    n_score = sum(bool(s.score_flag) for s in settings_list)
    if n_score not in (0, V):
        raise ValueError("the views of one call must all have score_flag set, or none of them")
    if score_sum is not None and (n_score != V or score_sum.dtype != torch.float32 or tuple(score_sum.shape) != (P,)
                                  or not score_sum.is_contiguous() or score_sum.device != dev):
        raise ValueError("score_sum must be a contiguous fp32 [P] tensor on the Gaussians' device, with score_flag views")
    if score_sum is not None and int(rc.score_mode) == 0:
        raise ValueError("score_sum with score_mode 0: use views.importance_scores (it sums raw pixel counts, score_mode 2, "
                         "and applies the opacity once)")
    batched = (1 < V <= MAX_VIEWS and same and P > 0 and rc.forward_mode == "auto" and ws.hint.get((P, H, W)) is not None
               and (W + 15) // 16 <= 256 and (H + 15) // 16 <= 256 and P < (1 << 24))

    # one segment length for all views of the call (their backward is one launch): from the capacity the batch will use --
    # or, before any capacity is known, whatever the first view picks for its exact size
    seg = rc.seg_len or (R.pick_seg_len(int(ws.hint[(P, H, W)] * 1.5), V) if ws.hint.get((P, H, W)) else None)

    def one_by_one():
        res, seg_k = [], seg
        for k, s in enumerate(settings_list):
            res.append(R.rasterize_forward_raw(s, means3D, opacities, shs, colors_precomp, sc(k), rotations, cov3D_precomp,
                                               want_aux=want_aux, rc=rc, seg_len=seg_k))
            seg_k = seg_k or int(res[-1][1].binning.seg_len)
        if score_sum is not None:
            for o, _ in res:
                if int(rc.score_mode) == 2:       # raw pixel counts: u32 bit patterns in the float32 tensors
                    score_sum.view(torch.int32).add_(o["score"].view(torch.int32))
                else:
                    score_sum.add_(o["score"])
                o["score"] = score_sum
        return res
    if not batched:
        return one_by_one()
    prof = rc.profile.handle if rc.profile is not None else None
    stride = R._align(int(lib.gsr_project_scratch_bytes(P)), 256)
    with torch.cuda.device(dev):
        big = ws.scratch("proj_scratch_batch", stride * V)
        sort_of = _sort_slices(ws, V)
        synced = [False]          # the first generator to resume waits for the projection; the others find it done
        if ws.batch_pinned is None:
            ws.batch_pinned = torch.zeros(MAX_VIEWS, dtype=torch.int64).pin_memory()
            ws.batch_pinned_np = ws.batch_pinned.numpy()
        dirty = [False]
        held = score_sum.clone() if score_sum is not None else None      # (restored if a view outgrows its capacity)
        gens = [R._forward_steps(s, means3D, opacities, shs, colors_precomp, sc(k), rotations, cov3D_precomp, False,
                                 want_aux, None, None,
                                 dict(scratch=big[k * stride:(k + 1) * stride], pinned=ws.batch_pinned, index=k,
                                      pinned_np=ws.batch_pinned_np, n_views=V,
                                      event=ws.event, sort=sort_of(k), synced=synced, score=score_sum, score_dirty=dirty,
                                      seg_len=seg, stream=stream, device_set=True),
                                 rc)
                for k, s in enumerate(settings_list)]
        results = _drive_batch(lib, ws, gens, V, dev, stream, prof)
        if dirty[0]:
            # a view's pair count exceeded the speculated capacity: its clamped lists already added to the shared buffer.
            # Rare (the capacity follows the recent maximum with 1.5x headroom): put the buffer back and take the views one
            # at a time (exact sizes)
            score_sum.copy_(held)
            return one_by_one()
    return results


def _sort_slices(ws, V):
    """k -> (nbytes -> slice k of one tensor holding V equally spaced sort scratch buffers)."""
    def provider(k):
        def get(nbytes):
            stride = R._align(int(nbytes), 256)
            big = ws.scratch("sort_scratch_batch", stride * V)
            return big[k * stride:(k + 1) * stride]
        return get
    return provider


def _drive_batch(lib, ws, gens, V, dev, stream, prof):
    """Common tail of the batched forwards: project all views, render all views, then let every generator finish."""
    heads = [next(g) for g in gens]                       # (view, geom, gaussians, binning, images, cap)
    views = (L.GsrView * V)(*[h[0] for h in heads])
    geoms = (L.GsrGeom * V)(*[h[1] for h in heads])
    gauss = (L.GsrGaussians * V)(*[h[2] for h in heads])
    L.check(lib.gsr_forward_project_batch(V, views, gauss, geoms, ws.batch_pinned.data_ptr(), stream, prof),
            "gsr_forward_project_batch")
    for k, h in enumerate(heads):
        h[1].sorted_idx = geoms[k].sorted_idx
    ws.event.record(torch.cuda.current_stream(dev))
    caps = {h[5] for h in heads}
    assert len(caps) == 1, caps
    bins = (L.GsrBinning * V)(*[h[3] for h in heads])
    imgs = (L.GsrImages * V)(*[h[4] for h in heads])
    L.check(lib.gsr_forward_render_batch(V, views, geoms, caps.pop(), bins, imgs, stream, prof),
            "gsr_forward_render_batch")
    results = []
    for g in gens:
        try:
            next(g)
            raise RuntimeError("forward generator did not finish")
        except StopIteration as e:
            results.append(e.value)
    return results


def _views_forward_scene(lib, settings_list, scenes, want_aux, rc):
    V = len(settings_list)
    s0 = settings_list[0]
    models = scenes[0]["models"]
    dev = models[0][0].device
    P = sum(int(m[0].shape[0]) for m in models)
    H, W = int(s0.image_height), int(s0.image_width)
    same = _uniform(settings_list)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ws = R._workspace(dev, stream)
    K = 1 + (int(models[0][5].shape[1]) if models[0][5] is not None and models[0][5].numel() > 0 else 0)
    batched = (1 < V <= MAX_VIEWS and same and P > 0 and rc.forward_mode == "auto" and ws.hint.get((P, H, W)) is not None
               and (W + 15) // 16 <= 256 and (H + 15) // 16 <= 256 and P < (1 << 24) and K in (1, 4, 9, 16)
               and not any(s.score_flag for s in settings_list))
    seg = rc.seg_len or (R.pick_seg_len(int(ws.hint[(P, H, W)] * 1.5), V) if ws.hint.get((P, H, W)) else None)   # one per call
    if not batched:
        res = []
        for s, sc in zip(settings_list, scenes):
            res.append(R.rasterize_forward_raw(s, None, None, None, None, None, None, None, want_aux=want_aux, scene=sc,
                                               rc=rc, seg_len=seg))
            seg = seg or int(res[-1][1].binning.seg_len)
        return res
    prof = rc.profile.handle if rc.profile is not None else None
    stride = R._align(int(lib.gsr_project_scratch_bytes(P)), 256)
    with torch.cuda.device(dev):
        big = ws.scratch("proj_scratch_batch", stride * V)
        sort_of = _sort_slices(ws, V)
        synced = [False]          # the first generator to resume waits for the projection; the others find it done
        if ws.batch_pinned is None:
            ws.batch_pinned = torch.zeros(MAX_VIEWS, dtype=torch.int64).pin_memory()
            ws.batch_pinned_np = ws.batch_pinned.numpy()
        gens = [R._forward_steps(s, None, None, None, None, None, None, None, False, want_aux, None, scenes[k],
                                 dict(scratch=big[k * stride:(k + 1) * stride], pinned=ws.batch_pinned, index=k,
                                      pinned_np=ws.batch_pinned_np, n_views=V,
                                      event=ws.event, sort=sort_of(k), synced=synced, seg_len=seg, stream=stream, device_set=True), rc)
                for k, s in enumerate(settings_list)]
        return _drive_batch(lib, ws, gens, V, dev, stream, prof)


class _RasterizeViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, settings_list, rc):
        res = rasterize_views_forward_raw(settings_list, means3D, opacities, shs, colors_precomp, scales, rotations,
                                          cov3D_precomp, rc=rc)
        ctx.states, ctx.rc = [st for _, st in res], rc
        ctx.opac_shape = opacities.shape
        ctx.has_means2D = means2D is not None
        ctx.per_view_scales = scales is not None and scales.dim() == 3      # [V,P,3]: every view its own scales
        ctx.set_materialize_grads(False)       # no zero tensors for outputs nobody differentiates (radii is [P] int32)
        outs = []
        for o, _ in res:
            ctx.mark_non_differentiable(o["radii"])
            outs += [o["color"], o["radii"], o["depth_alpha"]]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        sts, rc = ctx.states, ctx.rc
        V = len(sts)
        gcs, gdas = [], []
        for k, st in enumerate(sts):
            H, W, dev = st.view.image_height, st.view.image_width, st.dev
            g_color, g_da = grads[3 * k], grads[3 * k + 2]
            gcs.append(g_color if g_color is not None else torch.zeros((3, H, W), dtype=torch.float32, device=dev))
            gdas.append(g_da if g_da is not None else torch.zeros((2, H, W), dtype=torch.float32, device=dev))
        o = R.rasterize_backward_views_raw(sts, gcs, gdas, arena=rc.grad_arena, accumulate=rc.accumulate,
                                           stats=rc.densify_stats, stats_views=rc.stats_views,
                                           per_view_scales=ctx.per_view_scales, profile=rc.profile)
        if not ctx.has_means2D:
            o["dL_dmeans2D"] = None
        if rc.grad_arena is not None:      # the parameter gradients live in the arena, not in .grad (RasterContext)
            return (None, o["dL_dmeans2D"], None, o["dL_dcolors"], None,
                    o["dL_dscales"] if ctx.per_view_scales else None, None, o["dL_dcov3D"], None, None)
        return (o["dL_dmeans3D"], o["dL_dmeans2D"], o["dL_dshs"], o["dL_dcolors"],
                o["dL_dopacities"].reshape(ctx.opac_shape), o["dL_dscales"], o["dL_drotations"], o["dL_dcov3D"], None, None)


class GaussianRasterizerViews(torch.nn.Module):
    """The V views of one optimizer step through one call. All views must have the same image size and scale_modifier
    (the batched backward is one pass over all views); per view may differ: camera, background, active SH degree, and
    the scales ([V,P,3]). context: see rasterizer.RasterContext (densify_stats count for the LAST view unless
    context.stats_views says otherwise -- what the reference's trainers do, object_trainer.py:386-390)."""

    def __init__(self, raster_settings_list, context=None):
        super().__init__()
        self.raster_settings_list = list(raster_settings_list)
        self.context = context
        if not self.raster_settings_list:
            raise ValueError("at least one view")
        if not _uniform(self.raster_settings_list):
            raise ValueError("GaussianRasterizerViews: all views of a call must have the same image_height, image_width "
                             "and scale_modifier (render differently sized views with separate GaussianRasterizer calls)")

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None) -> List[tuple]:
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        V = len(self.raster_settings_list)
        if means2D is not None and means2D.shape[0] != V:      # (None: like GaussianRasterizer, nobody wants its gradient)
            raise ValueError(f"means2D must be [V,P,3] with V = {V} views")
        if any(s.score_flag for s in self.raster_settings_list):
            # forward-only, like the reference's score_render (scene_gaussian.py:546-671): per view the 4-tuple
            # (important_score, image, radii, depth_alpha) of GaussianRasterizer with score_flag
            if not all(s.score_flag for s in self.raster_settings_list):
                raise ValueError("the views of one call must all have score_flag set, or none of them")
            with torch.no_grad():
                rc_s = (self.context or R.DEFAULT_CONTEXT).snapshot()
                rc_s._forward_only = True
                res = rasterize_views_forward_raw(self.raster_settings_list, means3D, opacities, shs, colors_precomp, scales,
                                                  rotations, cov3D_precomp, rc=rc_s)
            return [(o["score"], o["color"], o["radii"], o["depth_alpha"]) for o, _ in res]
        rc = (self.context or R.DEFAULT_CONTEXT).snapshot()
        rc._forward_only = not (torch.is_grad_enabled() and any(
            t is not None and t.requires_grad for t in (means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                                        cov3D_precomp)))
        flat = _RasterizeViews.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                     tuple(self.raster_settings_list), rc)
        return [tuple(flat[3 * k:3 * k + 3]) for k in range(V)]


def importance_scores(settings_list: Sequence, means3D, opacities, shs=None, colors_precomp=None, scales=None,
                      rotations=None, cov3D_precomp=None, context=None, chunk: int = MAX_VIEWS, out=None):
    """Sum over the given cameras of the per-Gaussian importance score -- `prune_list` of the reference
    (scene_gaussian.py:1063-1079: 48 sphere cameras, one `score_render` each, `imp_list += important_score`). The cameras
    go through the batched forward `chunk` at a time: K1 once per chunk, the binning of all views through shared launches,
    K6 (score variant) of all views in ONE launch, every view's per-splat sums added to the one [P] buffer by the kernel's
    own atomics. settings_list: GaussianRasterizationSettings with score_flag=True (same image size / scale_modifier).
    Forward only. Returns the [P] fp32 sum (`out` if given: added to)."""
    settings_list = list(settings_list)
    if not settings_list or not all(s.score_flag for s in settings_list):
        raise ValueError("importance_scores needs at least one view, all with score_flag=True")
    rc = (context or R.DEFAULT_CONTEXT).snapshot()
    rc._forward_only = True
    P = int(means3D.shape[0])
    dev = means3D.device
    chunk = max(1, min(int(chunk), MAX_VIEWS))
    # weight 0 (opacity per contributing pixel): the kernels COUNT the pixels with integer atomics (score_mode 2: raw counts
    # in the buffer) over all cameras and the opacity is multiplied in once, in float64 -- exact up to the final rounding
    # (a float sum of ~10^4 equal increments is only good to ~1e-4). weight 1 (alpha * T): float atomics into the one buffer.
    counts_mode = int(rc.score_mode) == 0
    if counts_mode:
        rc.score_mode = 2
        acc = torch.zeros(P, dtype=torch.int32, device=dev).view(torch.float32)
    else:
        acc = torch.zeros(P, dtype=torch.float32, device=dev)
    with torch.no_grad():
        for i in range(0, len(settings_list), chunk):
            rasterize_views_forward_raw(settings_list[i:i + chunk], means3D, opacities, shs, colors_precomp, scales, rotations,
                                        cov3D_precomp, rc=rc, score_sum=acc)
        if counts_mode:
            acc = (acc.view(torch.int32).to(torch.float64) * opacities.reshape(-1).to(torch.float64)).to(torch.float32)
    if out is not None:
        out.add_(acc)
        return out
    return acc
