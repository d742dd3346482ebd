"""The V views of an optimizer step with their launches replayed from captured graphs (hipGraph through
torch.cuda.CUDAGraph) -- same kernels, same results as `GaussianRasterizerViews`, a fraction of the host work.

Why: a 4-view step is ~45 kernel launches issued from Python through ctypes; the host needs 0.93 ms to enqueue what the
GPU executes in 0.93 ms at 500 k Gaussians @1024^2, and smaller workloads are outright host-bound (100 k @512^2: 7 100
views/s where the kernels alone allow 11 400 -- round-1 measurements). In capacity mode every launch parameter of the
path depends only on (P, K, H, W, V, capacity): the kernels take the data-dependent pair count N from device memory. So
the launches of a step are three fixed sequences:
    graph F   K1 of all views, depth sorts, column counts (-> N of every view lands in pinned host words), pair emission,
              tile sort, ranges, work lists, K6
    graph C   work lists, K7 of all views, K8                        (backward; also captured over the caller's gradient
              addresses when those repeat, so that nothing is copied)
The host polls the pinned words while F still runs and checks N against the capacity (the same protocol as the eager
path: one wait per step, the GPU stays busy); if a view outgrew the capacity the step is redone eagerly with exact sizes
and the graphs are re-captured with more room. (Measured on ROCm 7.2: every graph boundary costs ~10-15 us of GPU idle
time, which is why the forward is ONE graph; and a graph whose ROOT node is a memset node started before the work
enqueued ahead of it had finished -- libgsrast clears memory with kernels only.)

What changes from step to step lives in device memory the graphs only point to:
  * the parameters: the caller's own tensors (persistent leaves updated in place by the optimizer); a new tensor
    (densification changes P) means a new capture;
  * the cameras: bg / viewmatrix / projmatrix / campos AND tanfov / active SH degree of every view sit in one packed
    block (`gsr_pack_views`, GsrView.dynamic) rewritten by one small launch per step from the step's settings;
  * per-view scales [V,P,3] (the trainers' scale noise) and the upstream gradients are copied into static buffers.
The outputs are static tensors too: they are overwritten by the next step's replay (consume them within the step, as
with any captured graph).

    rast = CapturedViews(context=RasterContext(grad_arena=arena))
    outs = rast(settings_list, means3D=..., means2D=m2d, opacities=..., shs=..., scales=..., rotations=...)
"""
from __future__ import annotations

import ctypes as C
import time
import weakref
from typing import List, Optional, Sequence

import torch

from . import _lib as L
from . import rasterizer as R
from . import views as VW

WARM_CALLS = 2          # eager calls (they learn the pair counts) before the first capture
MAX_DIRECT_GRAPHS = 4   # backward graphs captured over callers' gradient addresses (beyond that: copy + the static one)
PTR_MISSES_TO_STAGE = 2  # consecutive captures invalidated by nothing but new input ADDRESSES before the inputs are staged
PTR_REPEATS_TO_UNSTAGE = 8   # consecutive calls with identical input addresses before staged inputs go back to zero-copy


class _Captured:
    """Static buffers + the three graphs of one (inputs, geometry, capacity) signature."""
    pass


def _sig(settings_list, tensors, rc, per_view, by_ptr=True):
    """What a capture is valid for. by_ptr: the graphs read the caller's parameter tensors in place (persistent leaves:
    zero copies), so their addresses are part of the key; otherwise (staged inputs) only their shapes are. Everything that
    is baked into the captured structs belongs here: `prefiltered` and the forward variant included."""
    s0 = settings_list[0]
    return (len(settings_list), int(s0.image_height), int(s0.image_width), float(s0.scale_modifier), per_view,
            tuple(((t.data_ptr() if by_ptr else 0), tuple(t.shape)) if t is not None else None for t in tensors),
            tuple(bool(s.prefiltered) for s in settings_list), rc.fwd_variant,
            id(rc.grad_arena), bool(rc.accumulate), rc.score_mode,
            tuple(t.data_ptr() for t in rc.densify_stats) if rc.densify_stats is not None else None,
            tuple(rc.stats_views) if isinstance(rc.stats_views, (list, tuple)) else rc.stats_views)


class _Pending:
    """Lives as long as the autograd graph of a differentiable captured forward whose backward has not run: while one is alive,
    forward-only calls of the same CapturedViews (an eval render between a step's forward and its backward) take the eager
    path and leave the capture's static state and outputs alone."""
    __slots__ = ("done", "__weakref__")

    def __init__(self):
        self.done = False


class _CapturedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, opacities, scales, rotations, owner, settings_list, rc, differentiable):
        forward_only_beside = (not differentiable) and owner._backward_pending()
        if forward_only_beside:
            outs, cap_state, eager_states = owner._eager_forward(settings_list, means3D, opacities, shs, scales, rotations, rc)
        else:
            outs, cap_state, eager_states = owner._forward(settings_list, means3D, opacities, shs, scales, rotations, rc)
        ctx.owner, ctx.rc, ctx.cap_state, ctx.eager_states = owner, rc, cap_state, eager_states
        ctx.generation = cap_state.generation if cap_state is not None else -1
        ctx.pending = None
        if differentiable and cap_state is not None:
            ctx.pending = _Pending()
            owner._pending = weakref.ref(ctx.pending)
        ctx.opac_shape = opacities.shape
        ctx.per_view_scales = scales.dim() == 3
        ctx.set_materialize_grads(False)
        flat = []
        for (img, radii, da) in outs:
            ctx.mark_non_differentiable(radii)
            flat += [img, radii, da]
        return tuple(flat)

    @staticmethod
    def backward(ctx, *grads):
        cap = ctx.cap_state
        if cap is not None and cap.generation != ctx.generation:
            # The static forward state AND the outputs of a capture belong to its LATEST replay: a backward of an earlier
            # call (gradient accumulation over two forwards, retain_graph, an eval render in between) would use the later
            # step's state. Re-running the forward eagerly could restore the state, but not the OUTPUT tensors the caller's
            # loss saved for its own backward (they are the capture's static tensors, overwritten as well): the upstream
            # gradient would be computed from the wrong image. So: refuse. Callers with this pattern use the module whose
            # outputs are the caller's own -- GaussianRasterizerViews, or GaussianRasterizer with GSR_DROPIN_GRAPHS=1
            # (dropin.py: outputs copied out, state leased until the backward) -- INTEGRATION.md section 5b.
            # (A FORWARD-ONLY call in between -- torch.no_grad(), an eval render -- does not get here: while a differentiable
            #  forward waits for its backward such calls run eagerly and leave the capture alone, _Pending.)
            raise RuntimeError(
                "CapturedViews: backward of a forward whose captured state has been overwritten by a later forward "
                f"(replay {ctx.generation}, now {cap.generation}); run backward before the next forward of the same "
                "CapturedViews, or use GaussianRasterizerViews for calls whose graphs must stay alive")
        o = ctx.owner._backward(ctx.cap_state, ctx.eager_states, grads, ctx.rc, ctx.per_view_scales)
        if ctx.pending is not None:
            ctx.pending.done = True
        if ctx.rc.grad_arena is not None:
            return (None, o["dL_dmeans2D"], None, None, o["dL_dscales"] if ctx.per_view_scales else None, None,
                    None, None, None, None)
        return (o["dL_dmeans3D"], o["dL_dmeans2D"], o["dL_dshs"], o["dL_dopacities"].reshape(ctx.opac_shape),
                o["dL_dscales"], o["dL_drotations"], None, None, None, None)


class CapturedViews(torch.nn.Module):
    """Like views.GaussianRasterizerViews, with the settings of the step passed to forward() (cameras change every step;
    image size and scale_modifier must stay the same) and shs + scales + rotations as inputs (the trainers' case)."""

    def __init__(self, context: Optional[R.RasterContext] = None, headroom: float = 1.5):
        super().__init__()
        self.context = context
        self.headroom = float(headroom)
        self._cap: Optional[_Captured] = None
        self._warm = 0
        self._staged = False        # inputs copied into static buffers owned by the capture (see _forward)
        self._ptr_misses = 0
        self._ptr_repeats, self._last_ptr_sig = 0, None
        self._peak_n = 0
        self._fwd_mode = 0
        self._pending = None        # weak reference to the _Pending of the latest differentiable captured forward
        self.stats = dict(captures=0, replays=0, eager_steps=0, overflows=0, staged_inputs=False)

    def _backward_pending(self) -> bool:
        p = self._pending() if self._pending is not None else None
        return p is not None and not p.done

    # ------------------------------------------------------------------------------------------------ public
    def forward(self, raster_settings_list: Sequence, means3D, means2D, opacities, shs, scales, rotations) -> List[tuple]:
        settings_list = tuple(raster_settings_list)
        V = len(settings_list)
        if V < 1 or V > VW.MAX_VIEWS:
            raise ValueError(f"1..{VW.MAX_VIEWS} views per call")
        if not VW._uniform(settings_list):
            raise ValueError("all views of a call must have the same image_height, image_width and scale_modifier")
        if any(s.score_flag for s in settings_list):
            raise ValueError("score_flag views are forward-only: GaussianRasterizerViews / views.importance_scores render them batched")
        if means2D.shape[0] != V:
            raise ValueError(f"means2D must be [V,P,3] with V = {V} views")
        rc = (self.context or R.DEFAULT_CONTEXT).snapshot()
        differentiable = torch.is_grad_enabled() and any(
            t is not None and t.requires_grad for t in (means3D, means2D, shs, opacities, scales, rotations))
        flat = _CapturedFn.apply(means3D, means2D, shs, opacities, scales, rotations, self, settings_list, rc, differentiable)
        return [tuple(flat[3 * k:3 * k + 3]) for k in range(V)]

    # ------------------------------------------------------------------------------------------------ internals
    def _eager_forward(self, settings_list, means3D, opacities, shs, scales, rotations, rc):
        res = VW.rasterize_views_forward_raw(settings_list, means3D, opacities, shs, None, scales, rotations, None, rc=rc)
        self._peak_n = max([self._peak_n] + [int(o["N"]) for o, _ in res])
        self._fwd_mode = int(res[-1][1].binning.fwd_mode)      # the compositing variant the eager path settled on
        self.stats["eager_steps"] += 1
        return [(o["color"], o["radii"], o["depth_alpha"]) for o, _ in res], None, [st for _, st in res]

    def _forward(self, settings_list, means3D, opacities, shs, scales, rotations, rc):
        dev = means3D.device
        if dev.type != "cuda":
            raise L.GsrError("the HIP rasterizer needs tensors on a cuda (ROCm) device; there is no CPU fallback")
        per_view = scales.dim() == 3
        persistent = (means3D, opacities, shs, rotations) + (() if per_view else (scales,))
        for t in persistent + ((scales,) if per_view else ()):
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError("CapturedViews inputs must be contiguous fp32 tensors")
        # A capture reads the caller's parameter tensors IN PLACE (persistent leaves, updated in place by the optimizer:
        # nothing is copied) and is therefore keyed on their addresses. Callers that hand over fresh tensors every step
        # -- the reference's trainers pass activations: get_features is a torch.cat, get_opacity a sigmoid, get_scaling an
        # exp (gs_renderer.py:464-488) -- can never repeat an address the capture itself keeps alive, and every miss
        # would be a full re-capture. After PTR_MISSES_TO_STAGE captures lost to nothing but new addresses the inputs are
        # STAGED instead: copied into static buffers owned by the capture (one fused copy, 236 B per Gaussian at K = 16)
        # and the capture is keyed on shapes only.
        shape_sig = _sig(settings_list, persistent, rc, per_view, by_ptr=False) + (tuple(scales.shape),)
        ptr_sig = _sig(settings_list, persistent, rc, per_view) + (tuple(scales.shape),)
        if self._staged:
            # staging is not for ever: a caller that has settled on persistent tensors (the same addresses for
            # PTR_REPEATS_TO_UNSTAGE calls in a row) goes back to the zero-copy capture keyed on addresses
            self._ptr_repeats = self._ptr_repeats + 1 if ptr_sig == self._last_ptr_sig else 0
            self._last_ptr_sig = ptr_sig
            if self._ptr_repeats >= PTR_REPEATS_TO_UNSTAGE:
                self._staged, self._ptr_misses, self._ptr_repeats = False, 0, 0
                self.stats["staged_inputs"] = False
        sig = shape_sig if self._staged else ptr_sig
        cap = self._cap
        if cap is None or cap.sig != sig:
            if cap is not None and not self._staged and cap.shape_sig == shape_sig:
                self._ptr_misses += 1
                if self._ptr_misses >= PTR_MISSES_TO_STAGE:
                    self._staged, sig = True, shape_sig
                    self.stats["staged_inputs"] = True
            self._cap = cap = None
            if self._warm < WARM_CALLS or int(means3D.shape[0]) == 0:
                self._warm += 1
                return self._eager_forward(settings_list, means3D, opacities, shs, scales, rotations, rc)
            self._cap = cap = self._capture(sig, settings_list, means3D, opacities, shs, scales, rotations, rc, per_view)
            cap.shape_sig = shape_sig
        elif not self._staged:
            self._ptr_misses = 0
        lib = L.load()
        V = len(settings_list)
        stream = torch.cuda.current_stream(dev)
        with torch.cuda.device(dev):
            # this step's cameras -> the packed block the captured views point into (one small launch)
            structs = (L.GsrView * V)(*[self._user_view(s, cap.P, cap.K, rc) for s in settings_list])
            L.check(lib.gsr_pack_views(V, structs, cap.packed.data_ptr(), stream.cuda_stream), "gsr_pack_views")
            if cap.staged is not None:
                torch._foreach_copy_(cap.staged, [means3D, opacities, shs, rotations, scales])
            elif per_view:
                cap.scales.copy_(scales)
            cap.pinned_np[:] = -1
            cap.generation += 1
            cap.gF.replay()
            cap.evF.record(stream)
            # The pair counts land in the pinned words when the column-count kernel of the projection phase has run -- a
            # third of the way into graph F. The host polls them (page-locked, device-visible memory: nothing else tells the
            # host that a kernel INSIDE a graph has finished) so that the capacity check is over while the compositing
            # still runs; if the words only become visible when the graph is done, the event ends the wait.
            t_wait = time.perf_counter()
            spins = 0
            while True:
                ns = cap.pinned_np[:V]
                if int(ns.min()) >= 0:
                    break
                spins += 1
                if (spins & 63) == 0 and cap.evF.query():
                    break
            if rc.host_stats is not None:
                rc.host_stats.wait_s += time.perf_counter() - t_wait
        ns = [int(x) for x in cap.pinned_np[:V]]
        self._peak_n = max([self._peak_n] + ns)
        self.stats["replays"] += 1
        if max(ns) > cap.cap or min(ns) < 0:
            # a view outgrew the capacity (its lists were clamped): redo this step exactly, capture again next step
            self.stats["overflows"] += 1
            self._cap = None
            return self._eager_forward(settings_list, means3D, opacities, shs, scales, rotations, rc)
        return cap.outs, cap, None

    def _user_view(self, s, P, K, rc):
        if (int(s.sh_degree) + 1) ** 2 > K:     # the eager path answers GSR_EINVAL; a replay must not clamp silently
            raise ValueError(f"sh_degree {int(s.sh_degree)} needs {(int(s.sh_degree) + 1) ** 2} SH coefficients, shs holds {K}")
        dev = s.viewmatrix.device
        f = lambda t, n: R._prep(t.reshape(-1), n, dev, align=4)
        bg, vm, pm, cp = f(s.bg, "bg"), f(s.viewmatrix, "viewmatrix"), f(s.projmatrix, "projmatrix"), f(s.campos, "campos")
        v = R._view_struct(s, P, K, bg, vm, pm, cp, rc.score_mode)
        v._keep = (bg, vm, pm, cp)          # until gsr_pack_views has been enqueued (stream order covers the rest)
        return v

    def _capture(self, sig, settings_list, means3D, opacities, shs, scales, rotations, rc, per_view) -> _Captured:
        lib = L.load()
        dev = means3D.device
        V = len(settings_list)
        s0 = settings_list[0]
        P, K, H, W = int(means3D.shape[0]), int(shs.shape[1]), int(s0.image_height), int(s0.image_width)
        cap = _Captured()
        cap.sig, cap.P, cap.K, cap.V = sig, P, K, V
        want = int(self._peak_n * self.headroom) + 65536
        q = max(65536, 1 << max(0, want.bit_length() - 4))
        cap.cap = (want + q - 1) // q * q
        f32 = torch.float32
        with torch.cuda.device(dev):
            cap.packed = torch.zeros((V, L.GSR_PACKED_VIEW_FLOATS), dtype=f32, device=dev)
            cap.pinned = torch.zeros(VW.MAX_VIEWS, dtype=torch.int64).pin_memory()
            cap.generation = 0
            cap.staged = None
            if self._staged:
                # static copies of every input; the captured structs point at them
                cap.staged = [torch.empty_like(t) for t in (means3D, opacities, shs, rotations, scales)]
                torch._foreach_copy_(cap.staged, [means3D, opacities, shs, rotations, scales])
                means3D, opacities, shs, rotations, scales = cap.staged
            cap.scales = scales if (self._staged or not per_view) else torch.empty_like(scales)
            # the captured views: their camera tensors are slices of the packed block, tanfov / SH degree come from it too
            views = []
            for k, s in enumerate(settings_list):
                row = cap.packed[k]
                views.append(s._replace(bg=row[0:3], viewmatrix=row[4:20].view(4, 4), projmatrix=row[20:36].view(4, 4),
                                        campos=row[36:39]))
            stride = R._align(int(lib.gsr_project_scratch_bytes(P)), 256)
            cap.proj_scratch = torch.empty(stride * V + 4096, dtype=torch.uint8, device=dev)
            sort_bytes = R._align(int(lib.gsr_sort_scratch_bytes(cap.cap, lib.gsr_num_tiles(H, W))), 256)
            cap.sort_scratch = torch.empty(sort_bytes * V + 4096, dtype=torch.uint8, device=dev)
            gens = [R._forward_steps(
                views[k], means3D, opacities, shs, None, cap.scales[k] if per_view else scales, rotations, None, False,
                False, "auto", None,
                dict(scratch=cap.proj_scratch[k * stride:(k + 1) * stride], pinned=cap.pinned, index=k, event=None,
                     sort=(lambda nbytes, k=k: cap.sort_scratch[k * sort_bytes:(k + 1) * sort_bytes]),
                     capture=dict(cap=cap.cap, fwd_mode=int(rc.fwd_variant if rc.fwd_variant is not None else self._fwd_mode)),
                     dynamic=cap.packed[k].data_ptr() + 40 * 4, seg_len=rc.seg_len or R.pick_seg_len(cap.cap, V)), rc) for k in range(V)]
            heads = [next(g) for g in gens]
            cap.views = (L.GsrView * V)(*[h[0] for h in heads])
            cap.geoms = (L.GsrGeom * V)(*[h[1] for h in heads])
            cap.gauss = (L.GsrGaussians * V)(*[h[2] for h in heads])
            cap.bins = (L.GsrBinning * V)(*[h[3] for h in heads])
            cap.imgs = (L.GsrImages * V)(*[h[4] for h in heads])
            # a first, un-captured run fills the packed block and proves the arguments before anything is recorded
            structs = (L.GsrView * V)(*[self._user_view(s, P, K, rc) for s in settings_list])
            cur = torch.cuda.current_stream(dev)
            L.check(lib.gsr_pack_views(V, structs, cap.packed.data_ptr(), cur.cuda_stream), "gsr_pack_views")
            if per_view and cap.staged is None:
                cap.scales.copy_(scales)
            torch.cuda.synchronize(dev)
            cap.gF, cap.gC = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(cap.gF, capture_error_mode="thread_local"):
                st = torch.cuda.current_stream(dev).cuda_stream
                L.check(lib.gsr_forward_project_batch(V, cap.views, cap.gauss, cap.geoms, cap.pinned.data_ptr(), st, None),
                        "gsr_forward_project_batch (capture)")
                for k in range(V):
                    heads[k][1].sorted_idx = cap.geoms[k].sorted_idx
                L.check(lib.gsr_forward_render_batch(V, cap.views, cap.geoms, cap.cap, cap.bins, cap.imgs, st, None),
                        "gsr_forward_render_batch (capture)")
            res = []
            for g in gens:
                try:
                    next(g)
                    raise RuntimeError("forward generator did not finish")
                except StopIteration as e:
                    res.append(e.value)
            cap.states = [st_ for _, st_ in res]
            cap.outs = [(o["color"], o["radii"], o["depth_alpha"]) for o, _ in res]
            # ---- backward: static upstream-gradient buffers, static results
            cap.g_color = torch.zeros((V, 3, H, W), dtype=f32, device=dev)
            cap.g_da = torch.zeros((V, 2, H, W), dtype=f32, device=dev)
            cap.bwd = None
            cap.evF = torch.cuda.Event()
            cap.pinned_np = cap.pinned.numpy()       # the same page-locked words, readable without a torch call
            cap.rc = rc
            cap.per_view = per_view
        self.stats["captures"] += 1
        return cap

    def _capture_backward(self, cap: _Captured, rc, per_view):
        """Graph C over the static upstream-gradient buffers (the fallback every call can use after one copy)."""
        dev = cap.g_color.device
        with torch.cuda.device(dev):
            # run it once eagerly: allocates the result tensors (kept as the static ones) and validates the arguments
            # (persistent: the result tensors stay this capture's and nothing else writes them between replays -- K8 then clears
            #  only the rows the previous replay reached, GsrGrads.zero_outside; _trusted checks the premise before every replay and
            #  a replay it does not hold for takes the graph captured without it)
            o = R.rasterize_backward_views_raw(cap.states, list(cap.g_color), list(cap.g_da), arena=rc.grad_arena,
                                               accumulate=rc.accumulate, stats=None, per_view_scales=per_view,
                                               private_scratch=True, persistent=True)
            torch.cuda.synchronize(dev)
            with torch.cuda.graph(cap.gC, pool=cap.gF.pool(), capture_error_mode="thread_local"):
                oc = R.rasterize_backward_views_raw(cap.states, list(cap.g_color), list(cap.g_da), arena=rc.grad_arena,
                                                    accumulate=rc.accumulate, stats=rc.densify_stats,
                                                    stats_views=rc.stats_views, per_view_scales=per_view, reuse=o,
                                                    trust_zeros=True)
            cap.bwd = o
            cap.bwd_zo = int(oc["_zero_outside"])       # what graph C's K8 takes for granted about the result tensors
            cap.gC0 = None                              # graph C without that premise (captured when first needed)
            cap.bwd_versions = self._result_versions(cap, rc)
            self.stats["bwd_zero_outside"] = cap.bwd_zo
            cap.direct = {}          # (upstream-gradient addresses, premise) -> (graph C reading the caller's tensors, its zero_outside)

    _RESULTS = ("dL_dmeans3D", "dL_dopacities", "dL_dshs", "dL_dscales", "dL_drotations")

    @classmethod
    def _result_versions(cls, cap: _Captured, rc):
        """Version counters of the result tensors this capture owns: the per-view rows, and -- without an arena -- the summed
        gradients (an arena answers for itself: GradArena.zero_outside_ok)."""
        o = cap.bwd
        own = [o["_m2d"]] + ([o["dL_dscales"]] if cap.per_view and o.get("dL_dscales") is not None else [])
        if rc.grad_arena is None:
            own += [o[k] for k in cls._RESULTS if o.get(k) is not None]
        return tuple(t._version for t in own)

    def _trusted(self, cap: _Captured, rc, zo: int) -> bool:
        """Does what a K8 captured with GsrGrads.zero_outside = zo takes for granted hold right now? Normally yes: the result
        tensors hold what the previous replay wrote. Not after the arena went through a dense exchange or another writer
        (GradArena.zero_outside_ok), another module left its bitmap in the arena, or the caller edited a returned gradient in place
        (version counters)."""
        if not zo:
            return True
        arena = rc.grad_arena
        if self._result_versions(cap, rc) != cap.bwd_versions:
            return False
        if arena is not None:
            if (zo & 1) and not arena.zero_outside_ok(cap.bwd["_regions"]):
                return False
            if (zo & 2) and getattr(arena, "_mask_owner", None) is not cap.bwd["_token"]:
                return False
        return True

    def _static_graph(self, cap: _Captured, rc, per_view, trusted: bool):
        if trusted or not cap.bwd_zo:
            return cap.gC
        if cap.gC0 is None:
            cap.gC0 = torch.cuda.CUDAGraph()
            torch.cuda.synchronize(cap.g_color.device)
            with torch.cuda.graph(cap.gC0, pool=cap.gF.pool(), capture_error_mode="thread_local"):
                R.rasterize_backward_views_raw(cap.states, list(cap.g_color), list(cap.g_da), arena=rc.grad_arena,
                                               accumulate=rc.accumulate, stats=rc.densify_stats, stats_views=rc.stats_views,
                                               per_view_scales=per_view, reuse=cap.bwd, trust_zeros=False)
        return cap.gC0

    def _replayed_backward(self, cap: _Captured, rc) -> None:
        """Bookkeeping behind a replay of graph C (the launches ran without rasterize_backward_views_raw): the arena holds K8's
        rows and bitmap again, and the result tensors what K8 wrote."""
        if rc.grad_arena is not None:
            R._arena_written(rc.grad_arena, bool(rc.accumulate), cap.bwd["_token"], cap.bwd["_regions"])
        cap.bwd_versions = self._result_versions(cap, rc)

    def _direct_graph(self, cap: _Captured, gcs, gdas, rc, per_view, trusted: bool):
        """A loss written in torch produces its gradients at the same addresses step after step (the caching allocator
        returns the blocks it was just given back), so graph C is also captured over the CALLER'S gradient tensors: when
        the addresses repeat nothing is copied (80 MB per 4-view step at 1024^2). The tensors are only read while the
        backward that received them runs (stream order), exactly like the eager path reads them."""
        key = tuple(t.data_ptr() for t in gcs + gdas) + (bool(trusted or not cap.bwd_zo),)
        g = cap.direct.get(key)
        if g is None and len(cap.direct) < MAX_DIRECT_GRAPHS:
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize(cap.g_color.device)
            with torch.cuda.graph(g, pool=cap.gF.pool(), capture_error_mode="thread_local"):
                oc = R.rasterize_backward_views_raw(cap.states, gcs, gdas, arena=rc.grad_arena, accumulate=rc.accumulate,
                                                    stats=rc.densify_stats, stats_views=rc.stats_views,
                                                    per_view_scales=per_view, reuse=cap.bwd, trust_zeros=trusted)
            g = (g, int(oc["_zero_outside"]))
            cap.direct[key] = g
            self.stats["captures_bwd_direct"] = self.stats.get("captures_bwd_direct", 0) + 1
        return g

    def _backward(self, cap: Optional[_Captured], eager_states, grads, rc, per_view):
        V = len(grads) // 3
        if cap is None:            # a warm-up / overflow step: the eager backward on the eager states
            st0 = eager_states[0]
            H, W, dev = st0.view.image_height, st0.view.image_width, st0.dev
            z = lambda c: torch.zeros((c, H, W), dtype=torch.float32, device=dev)
            gcs = [grads[3 * k] if grads[3 * k] is not None else z(3) for k in range(V)]
            gdas = [grads[3 * k + 2] if grads[3 * k + 2] is not None else z(2) for k in range(V)]
            return R.rasterize_backward_views_raw(eager_states, gcs, gdas, arena=rc.grad_arena, accumulate=rc.accumulate,
                                                  stats=rc.densify_stats, stats_views=rc.stats_views,
                                                  per_view_scales=per_view, profile=rc.profile)
        dev = cap.g_color.device
        with torch.cuda.device(dev):
            if cap.bwd is None:
                self._capture_backward(cap, rc, per_view)
            gcs = [grads[3 * k] for k in range(V)]
            gdas = [grads[3 * k + 2] for k in range(V)]
            usable = all(g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.data_ptr() % 4 == 0
                         for g in gcs + gdas)
            trusted = self._trusted(cap, rc, cap.bwd_zo)
            g_direct = self._direct_graph(cap, gcs, gdas, rc, per_view, trusted) if usable else None
            if g_direct is not None and self._trusted(cap, rc, g_direct[1]):
                g_direct[0].replay()
                self._replayed_backward(cap, rc)
                return cap.bwd
            dst, src = [], []
            for k in range(V):
                for buf, g in ((cap.g_color[k], gcs[k]), (cap.g_da[k], gdas[k])):
                    if g is None:
                        buf.zero_()
                    elif g.data_ptr() != buf.data_ptr():
                        dst.append(buf)
                        src.append(g)
            if dst:
                torch._foreach_copy_(dst, src)
            self._static_graph(cap, rc, per_view, trusted).replay()
            self._replayed_backward(cap, rc)
        return cap.bwd
