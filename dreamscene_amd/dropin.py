"""The reference's OWN interface at speed: one `GaussianRasterizer(...)` call per view, replayed from captured graphs.

The unmodified trainers render the C_batch_size views of a step one call after the other, keep every view's outputs until
the loss, then run ONE backward over all of them (training/object_trainer.py:302-382, scene_gaussian.py:966-1021). Issued
eagerly a view is 24 kernel launches: ~0.3 ms of host work against 0.15-0.37 ms on the GPU -- the drop-in path was bound
by launch overhead (round 3: 2 314 views/s at 500 k @1024^2 where the kernels allow 2 680, 4 224 at 100 k @512^2 where they
allow ~8 000). This module puts the launches of a view behind two graph launches without changing what the caller sees:

  * a RING of captured single-view pipelines (`graph.CapturedViews` with V = 1) per (device, stream, P, K, H, W,
    scale_modifier, prefiltered). A slot is LEASED from the forward of a call until its backward has run (or until the
    autograd graph of the call is dropped): the views of a step sit in different slots, so every view's forward state
    (lists, checkpoints, final_T, the rendered image K7 re-reads) survives until its backward, whatever the order.
  * the OUTPUTS handed to the caller are fresh tensors (image, radii, depth_alpha copied out of the slot: 22 MB at
    1024^2) -- the caller may keep them as long as it likes, exactly like the eager path's.
  * slots are keyed on the ADDRESSES of the parameter tensors (zero-copy); the trainers' activations (`get_features` is a
    torch.cat, `get_opacity` a sigmoid, `get_scaling` an exp: gs_renderer.py:464-488) come back at the same addresses step
    after step because the caching allocator hands the same blocks out again -- view j of step s + 1 finds the slot
    view j of step s captured. A slot that keeps missing stages its inputs instead (one fused copy; CapturedViews).
  * pair counts: checked against the slot's capacity after every replay (pinned words, polled while the graph still
    runs); an overflow makes that call run eagerly with exact sizes, like every call before a slot is warm.
  * forward-only calls (torch.no_grad(): video_inference, object_trainer.py:81-118) use the same slots without a lease.
  * the GRADIENTS a backward returns are the slot's static tensors (as with CapturedViews), every one of them an object the
    slot's result dict holds as well: autograd consumes them in stream order (the next node's kernels are enqueued before
    the slot can be replayed again) and CLONES a tensor somebody else holds before it keeps it as `.grad` or adds into it,
    `retain_grad()` clones. Only the result of `torch.autograd.grad(...)` aliases a slot; it stays valid until that slot's
    next backward -- a full step when the views' forwards of a step precede their backwards (the trainers), but only until the
    NEXT view's backward when every view runs forward + backward before the next one starts (the freed slot is the first
    one the next call finds): clone what you keep. INTEGRATION.md section 5b'.

Not eligible (the eager path runs, as before): colors_precomp / cov3D_precomp inputs, score_flag, camera gradients, a
RasterContext with an arena / profile / densify_stats, non-fp32 or non-contiguous inputs, P = 0, grids beyond 256 x 256
tiles. OPT-IN: `GSR_DROPIN_GRAPHS=1` in the environment (unmodified trainers construct the module without a context) or
`RasterContext(dropin_graphs=True)` -- see the measurements at ENABLED below for when it pays.
"""
from __future__ import annotations

import os
import threading
from typing import Optional

import torch

from . import rasterizer as R

MAX_RINGS = 16          # (device, stream, P, K, H, W, ...) keys alive: the module's internal streams each have their own
MAX_SLOTS = 8           # per key: views of a step in flight (C_batch_size = 4) + the ones whose graph is still alive
# Opt-in (GSR_DROPIN_GRAPHS=1, or RasterContext(dropin_graphs=True) per module). Measured on MI355X / ROCm 7.2, round 4
# (gpurun_out/r4d, profiles/r04_dropin_graphs.txt): a graph launch of the 17 forward nodes costs the host 10 us instead of
# ~85 us of launches + ~60 us of Python, but the GPU then runs the 21 dependent kernels of a view with 5-8 us between
# nodes, and the copy-out adds launches: 100 k @512^2, forward + backward of a view right after each other 3 270 -> 3 900
# views/s; the trainers' pattern (four forwards, then four backwards) 3 440 -> 3 210, and at 500 k @1024^2 2 390 -> 1 940
# (eight slots' state cycling through the caches). Neither pattern reaches what ONE batched call gives (10 600 / 4 900).
ENABLED = os.environ.get("GSR_DROPIN_GRAPHS", "0") == "1"

_RINGS = {}
_LOCK = threading.Lock()


class _Slot:
    __slots__ = ("cv", "busy", "stamp")

    def __init__(self, rc):
        from .graph import CapturedViews
        self.cv = CapturedViews(context=rc)
        self.busy = False
        self.stamp = 0


class _Lease:
    """Held by the autograd context of a call: the slot is free again when the backward has run or the graph died."""
    __slots__ = ("slot",)

    def __init__(self, slot):
        self.slot = slot
        slot.busy = True

    def release(self):
        s, self.slot = self.slot, None
        if s is not None:
            s.busy = False

    def __del__(self):
        self.release()


class _Ring:
    def __init__(self):
        self.slots = []
        self.clock = 0
        self.stats = dict(calls=0, replays=0, eager=0, no_slot=0)

    def acquire(self, ptr_sig, rc) -> Optional[_Slot]:
        """A free slot: one whose capture was made over these very input addresses, else one without a capture yet, else the
        least recently used free one (its CapturedViews re-captures or stages), else a new one; None when all are leased."""
        self.clock += 1
        free = [s for s in self.slots if not s.busy]
        best = None
        for s in free:
            cap = s.cv._cap
            if cap is not None and getattr(cap, "ptr_sig", None) == ptr_sig:
                best = s
                break
        if best is None:
            for s in free:
                if s.cv._cap is None and (best is None or s.stamp < best.stamp):
                    best = s
        if best is None and len(self.slots) < MAX_SLOTS:
            best = _Slot(rc)
            if self.slots:       # what the ring has learnt (pair counts, forward variant, warm-up) carries over
                o = self.slots[0].cv
                best.cv._peak_n, best.cv._fwd_mode, best.cv._warm = o._peak_n, o._fwd_mode, o._warm
            self.slots.append(best)
        if best is None and free:
            best = min(free, key=lambda s: s.stamp)
        if best is not None:
            best.stamp = self.clock
        return best


def eligible(s, means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, context,
             wanted: Optional[bool] = None) -> bool:
    """wanted: the caller has already decided that this accelerator is the selected one (rasterizer.per_view_accel)."""
    if wanted is None:
        want = getattr(context, "dropin_graphs", None) if context is not None else None
        wanted = ENABLED if want is None else bool(want)
    if not wanted:
        return False
    if shs is None or colors_precomp is not None or cov3D_precomp is not None:
        return False
    if scales is None or rotations is None or s.score_flag:
        return False
    if context is not None and (context.grad_arena is not None or context.profile is not None or
                                context.densify_stats is not None or context.forward_mode != "auto"):
        return False
    if means3D.device.type != "cuda" or means3D.shape[0] == 0 or means3D.shape[0] >= (1 << 24):
        return False
    if (int(s.image_width) + 15) // 16 > 256 or (int(s.image_height) + 15) // 16 > 256:
        return False
    if scales.dim() != 2 or shs.dim() != 3 or shs.shape[1] not in (1, 4, 9, 16):
        return False
    for t in (means3D, opacities, shs, scales, rotations):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != means3D.device:
            return False
    if any(getattr(t, "requires_grad", False) for t in (s.viewmatrix, s.projmatrix, s.campos)):
        return False
    if torch.cuda.is_current_stream_capturing():
        return False
    return True


def _ring_for(s, means3D, shs) -> _Ring:
    dev = means3D.device
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, int(means3D.shape[0]), int(shs.shape[1]),
           int(s.image_height), int(s.image_width), float(s.scale_modifier), bool(s.prefiltered))
    with _LOCK:
        r = _RINGS.get(key)
        if r is None:
            if len(_RINGS) >= MAX_RINGS:  # P changes with every densification: drop the rings of the old sizes
                _RINGS.pop(next(iter(_RINGS)))
            r = _RINGS[key] = _Ring()
    return r


def _copy_out(outs):
    img, radii, da = outs
    o_img, o_da, o_r = torch.empty_like(img), torch.empty_like(da), torch.empty_like(radii)
    torch._foreach_copy_([o_img, o_da], [img, da])
    o_r.copy_(radii)
    return o_img, o_r, o_da


class _DropinFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, opacities, scales, rotations, ring, slot, settings, rc):
        cv = slot.cv
        outs, cap_state, eager_states = cv._forward((settings,), means3D, opacities, shs, scales, rotations, rc)
        img, radii, da = outs[0]
        if cap_state is not None:
            cap_state.ptr_sig = ring_sig(means3D, opacities, shs, scales, rotations) if not cv._staged else None
            img, radii, da = _copy_out(outs[0])
            ring.stats["replays"] += 1
        else:
            ring.stats["eager"] += 1
        ctx.cv, ctx.rc, ctx.cap_state, ctx.eager_states = cv, rc, cap_state, eager_states
        ctx.generation = cap_state.generation if cap_state is not None else -1
        # the slot's static state belongs to this call until its backward has run (captured), or not at all (eager state)
        ctx.lease = _Lease(slot) if cap_state is not None else None
        ctx.opac_shape = opacities.shape
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii)
        return img, radii, da

    @staticmethod
    def backward(ctx, g_img, _g_radii, g_da):
        cap = ctx.cap_state
        if cap is not None and cap.generation != ctx.generation:
            raise RuntimeError("GaussianRasterizer (captured drop-in path): second backward of a call whose slot has been "
                               "reused; pass RasterContext(dropin_graphs=False) for calls differentiated more than once")
        try:
            o = ctx.cv._backward(cap, ctx.eager_states, (g_img, None, g_da), ctx.rc, False)
        finally:
            if ctx.lease is not None:
                ctx.lease.release()
        # Everything handed to autograd must be a tensor OBJECT somebody else holds too (the slot's result dict): autograd
        # steals a gradient whose use count is 1 -- AccumulateGrad makes it the leaf's `.grad`, the engine's input buffers add
        # into it in place -- and a fresh view object of the slot's static storage (m2d[0], a reshape) has use count 1: `.grad`
        # would alias the slot, and the slot's next backward would overwrite it. The views are made once per slot
        # and kept in the dict; a held tensor is cloned by autograd before it keeps or modifies it.
        vw = o.get("_dropin_views")
        if vw is None or vw[2] != tuple(ctx.opac_shape):
            m2d = o["dL_dmeans2D"]
            vw = o["_dropin_views"] = (m2d[0] if m2d.dim() == 3 else m2d, o["dL_dopacities"].reshape(ctx.opac_shape),
                                       tuple(ctx.opac_shape))
        return (o["dL_dmeans3D"], vw[0], o["dL_dshs"], vw[1], o["dL_dscales"], o["dL_drotations"], None, None, None, None)


def ring_sig(means3D, opacities, shs, scales, rotations):
    return (means3D.data_ptr(), opacities.data_ptr(), shs.data_ptr(), scales.data_ptr(), rotations.data_ptr())


def rasterize(s, means3D, means2D, opacities, shs, scales, rotations, context):
    """The captured drop-in call (the caller has checked `eligible`). Returns (image, radii, depth_alpha), or None when no
    slot is free (more than MAX_SLOTS views in flight): the caller then takes the eager path."""
    ring = _ring_for(s, means3D, shs)
    rc = (context or R.DEFAULT_CONTEXT).snapshot()
    slot = ring.acquire(ring_sig(means3D, opacities, shs, scales, rotations), rc)
    ring.stats["calls"] += 1
    if slot is None:
        ring.stats["no_slot"] += 1
        return None
    needs_grad = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (means3D, means2D, opacities, shs, scales, rotations))
    if not needs_grad:
        # forward only: nothing of the slot is needed once the outputs are copied out
        with torch.no_grad():
            outs, cap_state, _ = slot.cv._forward((s,), means3D, opacities, shs, scales, rotations, rc)
            if cap_state is None:
                ring.stats["eager"] += 1
                return outs[0]
            cap_state.ptr_sig = ring_sig(means3D, opacities, shs, scales, rotations) if not slot.cv._staged else None
            ring.stats["replays"] += 1
            return _copy_out(outs[0])
    if means2D is None:
        means2D = torch.zeros_like(means3D)
    return _DropinFn.apply(means3D, means2D, shs, opacities, scales, rotations, ring, slot, s, rc)


def stats():
    """Per ring: calls / replays / eager / no_slot counters and the number of slots (diagnostics, bench.py)."""
    return {str(k): dict(r.stats, slots=len(r.slots)) for k, r in _RINGS.items()}


def reset():
    """Drop every ring (tests; also frees the slots' device memory)."""
    with _LOCK:
        _RINGS.clear()
