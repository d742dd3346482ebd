"""Host-side mirror of the reference's rasterizer interface, bound to the HIP library through the C ABI.

Same names, argument meaning and error behaviour as the package DreamScene imports
(`from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`,
scene_gaussian.py:11-12):
  * GaussianRasterizationSettings -- the 12 keyword fields of scene_gaussian.py:951-964 (with `score_flag`,
    without upstream's `debug`);
  * GaussianRasterizer(raster_settings=...)(means3D=, means2D=, shs=, colors_precomp=, opacities=, scales=,
    rotations=, cov3D_precomp=) -> (image [3,H,W], radii [P] int32, depth_alpha [2,H,W]), with a leading
    important_score [P] when raster_settings.score_flag (scene_gaussian.py:637, 1012);
  * gradients flow through image AND depth_alpha (scene_gaussian.py:1023-1032) to means3D, means2D (the dummy
    screen-space tensor, gs_renderer.py:1061-1065), opacities, shs / colors_precomp, scales, rotations /
    cov3D_precomp.
PyTorch is used for device memory, streams and autograd plumbing only; all arithmetic is in libgsrast.so.
"""
from __future__ import annotations

import os

import collections
import contextlib
import ctypes as C
import dataclasses
import threading
import time
import weakref
from typing import NamedTuple, Optional, Sequence

import torch

from . import _lib as L


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    score_flag: bool = False


@dataclasses.dataclass
class RasterContext:
    """Everything that steers a rasterizer call beyond the reference's 12 settings fields. One object per
    GaussianRasterizer (or per call of the raw functions); the autograd Functions take a SNAPSHOT of it at forward time
    and the backward -- which autograd runs on its own thread -- reads only that snapshot: nothing here is a module
    global, two rasterizers with different contexts can run forward / backward concurrently.

    score_mode     important_score weight (the fork's exact definition is unpinned, SEMANTICS.md section 4):
                   0 = opacity per contributing (pixel, splat) pair [default, LightGaussian's hit term], 1 = alpha * T
    profile        optional GsrProfile handle (bench.py: per-kernel HIP-event timings)
    grad_arena     optional multiview.GradArena: the backward writes the parameter gradients of the view straight into
                   the arena's flat buffer (zero-copy hand-off to the exchange) and returns NO parameter gradients to
                   autograd (param.grad is left alone: the arena is where they live; returning views of it would make
                   loss.backward() add the arena to .grad a second time)
    accumulate     with an arena: False = this call's gradients overwrite the arena, True = they are ADDED on the
                   device (sum over the views of one optimizer step before a single exchange)
    densify_stats  optional (max_radii2D, xyz_gradient_accum, denom) fp32 [P] tensors updated inside K8 for the visible
                   Gaussians of the call's view (object_trainer.py:386-390); several views per call: `stats_views`
    stats_views    indices of the views of a multi-view call whose statistics count; None = the LAST view only, which is
                   what the reference's trainers do (the loop's last viewspace_points / visibility_filter / radii)
    forward_mode   "auto": speculate the pair capacity from previous calls and enqueue the whole forward without draining
                   the GPU (exact re-run if it was too small); "sync": always the exact two-phase forward
    fwd_variant    forward compositing variant (GsrBinning.fwd_mode): None = per call from the previous view's statistics
    dropin_graphs  GaussianRasterizer only: True = replay the call's launches from captured graphs (dropin.py) when the call is
                   eligible (SH + scales + rotations inputs, no arena / profile / statistics); False = always eager; None =
                   what the environment says (GSR_DROPIN_GRAPHS=1: on; default off -- dropin.py has the measurements)
    host_stats     optional HostStats: seconds the calls made with this context spent blocked on the projection's pair
                   counts (bench.py reports it per step)
    seg_len        entries per forward checkpoint / backward work item (GsrBinning.seg_len): None = per launch from its size
                   (pick_seg_len); 256, 128 or 64 = that for every call made with this context
    side_streams   GaussianRasterizer only: consecutive calls rotate over this many internal HIP streams (see `_SideStreams`):
                   None = what the environment says (GSR_SIDE_STREAMS=n; default SIDE_STREAMS_DEFAULT), 0 / 1 = the caller's stream
    per_view_accel GaussianRasterizer only: the ONE opt-in accelerator of the per-view call -- "off", "streams" (internal
                   streams, `side_streams` of them, 2 when unset) or "graphs" (captured ring, dropin.py). None = what the
                   environment says (GSR_PER_VIEW_ACCEL=off|streams|graphs), else derived from the two fields above / their
                   environment variables. The two are mutually exclusive BY CONSTRUCTION: asking
                   for both (fields or environment) raises -- they are slower together than either alone (HISTORY round 5:
                   2 420 vs 2 497 / 2 583 views/s) and serve different call patterns.
    """
    score_mode: int = 0
    profile: Optional[L.Profile] = None
    grad_arena: Optional[object] = None
    accumulate: bool = False
    densify_stats: Optional[tuple] = None
    stats_views: Optional[Sequence[int]] = None
    forward_mode: str = "auto"
    fwd_variant: Optional[int] = None
    dropin_graphs: Optional[bool] = None
    host_stats: Optional["HostStats"] = None
    seg_len: Optional[int] = None
    side_streams: Optional[int] = None
    per_view_accel: Optional[str] = None

    def snapshot(self) -> "RasterContext":
        # (a plain field-for-field copy: dataclasses.replace() re-runs __init__ through a keyword dict, ~3 us per call)
        c = object.__new__(RasterContext)
        c.__dict__.update(self.__dict__)
        return c


class HostStats:
    """Seconds spent blocked on the projection's pair counts by the calls that carry this object in their RasterContext (a
    statistic, not a switch: how much of a step the host is idle, i.e. how far the path is from being bound by host-side
    enqueueing). Shared by reference between a context and its per-call snapshots."""
    __slots__ = ("wait_s",)

    def __init__(self):
        self.wait_s = 0.0


DEFAULT_CONTEXT = RasterContext()


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _prep(t: Optional[torch.Tensor], name: str, dev: torch.device, align: int = 16) -> Optional[torch.Tensor]:
    """fp32, contiguous, on `dev`, base aligned to `align` bytes (the ABI wants 16 for shs / rotations / splat rows; the
    other inputs are read element-wise, e.g. the [k] slices of per-view scales [V,P,3] with P % 4 != 0)."""
    if t is None:
        return None
    if t.device != dev:
        raise ValueError(f"{name} is on {t.device}, expected {dev}")
    if t.dtype != torch.float32:
        t = t.float()
    t = t.contiguous()
    if t.data_ptr() % align:
        t = t.clone()
    return t


def _view_struct(s: GaussianRasterizationSettings, P: int, K: int, bg, vm, pm, cp, score_mode: int = 0) -> L.GsrView:
    v = L.GsrView()
    v.P, v.sh_stride, v.sh_degree = P, K, int(s.sh_degree)
    v.image_height, v.image_width = int(s.image_height), int(s.image_width)
    v.tanfovx, v.tanfovy, v.scale_modifier = float(s.tanfovx), float(s.tanfovy), float(s.scale_modifier)
    v.prefiltered, v.score_mode = int(bool(s.prefiltered)), int(score_mode)
    v.bg, v.viewmatrix, v.projmatrix, v.campos = bg.data_ptr(), vm.data_ptr(), pm.data_ptr(), cp.data_ptr()
    return v


class _State:
    """Everything the backward needs; tensors are kept alive here (the C side owns nothing)."""
    __slots__ = ("view", "gauss", "geom", "binning", "images", "keep", "P", "K", "N", "dev", "cam_grads", "scene",
                 "versions")


class _Workspace:
    """Per (device, stream) reusable scratch + the pair-count speculation state. Scratch is only touched by work
    enqueued on that stream, so reuse across calls is ordered by the stream itself."""

    def __init__(self, dev):
        self.dev = dev
        self.n_pinned = torch.zeros(1, dtype=torch.int64).pin_memory()
        self.n_pinned_np = self.n_pinned.numpy()       # the same page-locked word, readable without a torch call
        self.stats_pinned = torch.zeros(2, dtype=torch.int32).pin_memory()   # [0]: non-empty tiles of the last view
        self.stats_pinned_np = self.stats_pinned.numpy()
        self.event = torch.cuda.Event()
        self.batch_pinned = None     # pinned int64 [GSR_MAX_BATCH_VIEWS]: pair counts of a batched projection
        self.batch_pinned_np = None
        self.proj_scratch = None
        self.proj_scratch_batch = None
        self.sort_scratch_batch = None
        self.sort_scratch = None
        self.hint = {}           # (P, H, W) -> decaying max of recent pair counts
        self.last_stats = {}     # (P, H, W) -> (non-empty tiles or None = read the pinned word, pair count)

    def scratch(self, which: str, nbytes: int) -> torch.Tensor:
        t = getattr(self, which)
        if t is None or t.numel() < nbytes:
            t = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=self.dev)
            setattr(self, which, t)
        return t


_NULL_CTX = contextlib.nullcontext()
_PAIR_COUNT_SPINS = 4096     # polls of the pair-count word before _wait_pair_counts falls back to event.synchronize()
_WORKSPACES = {}
_STATE_LAYOUTS = {}     # (sizes of a call) -> (offsets of the regions of its state buffer, total bytes)


def _workspace(dev, stream) -> _Workspace:
    key = (dev.index, stream)
    ws = _WORKSPACES.get(key)
    if ws is None:
        ws = _WORKSPACES[key] = _Workspace(dev)
    return ws


def _align(n: int, a: int = 256) -> int:
    return (n + a - 1) // a * a


def _wait_pair_counts(words, n: int, event) -> None:
    """Blocks until the n page-locked pair-count words (numpy int64 view) hold counts. The library stores a count EARLY -- the
    first workgroup of the depth sort's first pass writes it, ~40 us after K1 started (include/gsrast.h, GSR_N_PENDING = -1
    until then) -- so the host polls the words instead of waiting ~120 us for the event behind the whole projection; when the
    event has completed everything in front of it has run and the words are final whatever path wrote them."""
    spins = 0
    while True:
        if (words[0] != -1) if n == 1 else (int(words[:n].min()) != -1):
            return
        spins += 1
        if (spins & 31) == 0:
            if event.query():
                return
            if spins >= _PAIR_COUNT_SPINS:
                # the early store has not become visible for ~a millisecond of polling (page-locked memory that is not
                # fine-grained -- HIP_HOST_COHERENT=0 -- shows it at kernel end only): stop burning a core under the GIL and
                # block on the event; the words are final once it has completed
                event.synchronize()
                return
            time.sleep(0)            # let other Python threads (data loaders, other ranks' helper threads) take the GIL


def rasterize_forward_raw(s: GaussianRasterizationSettings, means3D, opacities, shs, colors_precomp, scales,
                          rotations, cov3D_precomp, want_keys: bool = False, want_aux: bool = True,
                          mode: Optional[str] = None, scene: Optional[dict] = None,
                          rc: Optional[RasterContext] = None, seg_len: Optional[int] = None):
    gen = _forward_steps(s, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, want_keys,
                         want_aux, mode, scene, None, rc or DEFAULT_CONTEXT, seg_len=seg_len)
    try:
        next(gen)
    except StopIteration as e:
        return e.value
    raise RuntimeError("unreachable: a non-batched forward does not yield")


# GsrBinning.seg_len (include/gsrast.h): list entries per forward checkpoint = per work item of the backward. A backward item
# is a serial recurrence over its entries (~0.3 us each on MI355X), so no launch ends before its longest item has: 65-90 us
# with 256 entries, whatever the size of the launch. Launches with little total work (one view of the reference's per-view
# interface, small scenes) finish sooner with 128-entry items; large ones (the 4-view step at 500 k Gaussians) are bound by
# their total work and only pay for the extra checkpoints (K6) and item prologues (K7). Measured, round 4 (K7 / K6 / GPU time
# per step in us, seg 256 -> 128 -> 64; profiles/HISTORY.md): one view 500 k @1024^2 92 / 51 / 370 -> 83 / 54 / 364 -> 78 / 61 /
# 370; one view 100 k @512^2 65 / 28 / 227 -> 44 / 31 / 206 -> 36 / 35 / 207; 4 views 100 k @512^2 103 / 50 / 330 -> 88 / 54 / 321 ->
# 83 / 58 / 323; 4 views 500 k @1024^2 222 / 168 / 807 -> 239 / 180 / 833. 64 never wins (the library supports it; the policy
# does not pick it). The threshold is the pair CAPACITY (1.5x the recent pair count) summed over the views of the launch.
SEG_LEN_256_FROM = 8_000_000


def pick_seg_len(cap, n_views: int = 1) -> int:
    env = os.environ.get("GSR_SEG_LEN")       # (tools / experiments: 64, 128 or 256 for every call)
    if env:
        if env.strip() not in ("64", "128", "256"):
            raise ValueError(f"GSR_SEG_LEN must be 64, 128 or 256, got {env!r}")
        return int(env)
    if not cap:
        return 256
    return 256 if int(cap) * int(n_views) >= SEG_LEN_256_FROM else 128


def _forward_steps(s: GaussianRasterizationSettings, means3D, opacities, shs, colors_precomp, scales,
                   rotations, cov3D_precomp, want_keys, want_aux, mode, scene, batch, rc: RasterContext, seg_len=None):
    """Generator behind rasterize_forward_raw. With batch = dict(scratch=<this view's slice of the batch's projection
    scratch>, pinned=<pinned int64 [V]>, index=k, event=<Event>) it allocates and binds, YIELDS (view struct, geom struct)
    for the caller to run gsr_forward_project_batch / gsr_forward_render_batch over all views, and continues when resumed.
    Returns (outputs, state) through StopIteration.value."""
    """Forward through the C ABI. Returns (outputs dict, _State). Used by the autograd Function and by tests.
    scene (SURVEY.md 8f rank 2): {"models": [(xyz, scaling, rotation, opacity, features_dc, features_rest), ...] raw
    leaf tensors, "scale_noise": [P,3] | None, "sh_noise": [P,K,3] | None, "want_act": bool}; the per-Gaussian
    tensor arguments must then be None."""
    lib = L.load()
    dev = (scene["models"][0][0] if scene is not None else means3D).device
    if dev.type != "cuda":
        raise L.GsrError("the HIP rasterizer needs tensors on a cuda (ROCm) device; there is no CPU fallback")
    H, W = int(s.image_height), int(s.image_width)
    sc_struct, sc_keep, sc_out = None, None, {}
    if scene is not None:
        if any(t is not None for t in (means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp)):
            raise ValueError("with scene= the per-Gaussian tensor arguments must be None")
        models = scene["models"]
        if not 1 <= len(models) <= L.GSR_MAX_MODELS:
            raise ValueError(f"1..{L.GSR_MAX_MODELS} models per call, got {len(models)}")
        sc_struct = L.GsrScene()
        sc_struct.n_models = len(models)
        sc_keep = []
        P, K_scene = 0, None
        for m, md in enumerate(models):
            xyz, scaling, rotation, opacity, f_dc, f_rest = [_prep(t, "model tensor", dev) for t in md]
            n = int(xyz.shape[0])
            k = 1 + (int(f_rest.shape[1]) if f_rest is not None and f_rest.numel() > 0 else 0)
            if n > 0:
                if K_scene is not None and k != K_scene:
                    raise ValueError("all models of a scene must have the same number of SH coefficients")
                K_scene = k
            ms = sc_struct.models[m]
            ms.count = n
            ms.xyz, ms.scaling, ms.rotation, ms.opacity = _ptr(xyz), _ptr(scaling), _ptr(rotation), _ptr(opacity)
            ms.features_dc, ms.features_rest = _ptr(f_dc), (_ptr(f_rest) if k > 1 else None)
            sc_keep.append((xyz, scaling, rotation, opacity, f_dc, f_rest))
            P += n
        K_scene = K_scene or 1
        sn, hn = _prep(scene.get("scale_noise"), "scale_noise", dev), _prep(scene.get("sh_noise"), "sh_noise", dev)
        if sn is not None and tuple(sn.shape) != (P, 3):
            raise ValueError(f"scale_noise must be [P,3], got {tuple(sn.shape)}")
        if hn is not None and tuple(hn.shape) != (P, K_scene, 3):
            raise ValueError(f"sh_noise must be [P,K,3], got {tuple(hn.shape)}")
        sc_struct.scale_noise, sc_struct.sh_noise = _ptr(sn), _ptr(hn)
        sc_out["scales"] = torch.empty((P, 3), dtype=torch.float32, device=dev)
        sc_struct.scales_out = sc_out["scales"].data_ptr()
        if scene.get("want_act"):
            sc_out["rotations"] = torch.empty((P, 4), dtype=torch.float32, device=dev)
            sc_out["opacities"] = torch.empty((P,), dtype=torch.float32, device=dev)
            sc_struct.rotations_out, sc_struct.opacities_out = sc_out["rotations"].data_ptr(), sc_out["opacities"].data_ptr()
        sc_keep.append((sn, hn))
    else:
        P = int(means3D.shape[0])
    means3D = _prep(means3D, "means3D", dev)
    opacities = _prep(opacities, "opacities", dev)
    shs, colors_precomp = _prep(shs, "shs", dev), _prep(colors_precomp, "colors_precomp", dev)
    scales, rotations = _prep(scales, "scales", dev, align=4), _prep(rotations, "rotations", dev)
    cov3D_precomp = _prep(cov3D_precomp, "cov3D_precomp", dev)
    # (contiguous [4,4] / [3] tensors are read through their data pointers as they are: no reshape, which is a view op)
    bg = _prep(s.bg, "bg", dev, align=4)
    vm = _prep(s.viewmatrix, "viewmatrix", dev, align=4)
    pm = _prep(s.projmatrix, "projmatrix", dev, align=4)
    cp = _prep(s.campos, "campos", dev, align=4)
    if bg.numel() != 3 or vm.numel() != 16 or pm.numel() != 16 or cp.numel() != 3:
        raise ValueError("raster settings: bg [3], viewmatrix [4,4], projmatrix [4,4], campos [3] expected")
    K = int(shs.shape[1]) if shs is not None else (K_scene if scene is not None else 0)
    if shs is not None and (shs.dim() != 3 or shs.shape[0] != P or shs.shape[2] != 3):
        raise ValueError(f"shs must be [P,K,3], got {tuple(shs.shape)}")

    st = _State()
    st.P, st.K, st.dev = P, K, dev
    st.view = _view_struct(s, P, K, bg, vm, pm, cp, rc.score_mode)
    if batch is not None and batch.get("dynamic") is not None:
        st.view.dynamic = int(batch["dynamic"])     # device f32[4]: tanfov / SH degree read when the kernels run
    g = L.GsrGaussians()
    g.means3D, g.opacities, g.shs, g.colors_precomp = _ptr(means3D), _ptr(opacities), _ptr(shs), _ptr(colors_precomp)
    g.scales, g.rotations, g.cov3D_precomp = _ptr(scales), _ptr(rotations), _ptr(cov3D_precomp)
    if sc_struct is not None:
        g.scene = C.pointer(sc_struct)
    st.gauss = g
    st.scene = (sc_struct, sc_keep) if sc_struct is not None else None

    # (a batched call looks the stream up once and sets the device once for all its views: torch.cuda.current_stream() and the
    #  device context manager cost ~6 us each from Python, six of them per view)
    stream = batch["stream"] if (batch is not None and batch.get("stream") is not None) else \
        torch.cuda.current_stream(dev).cuda_stream
    ws = _workspace(dev, stream)
    prof = rc.profile.handle if rc.profile is not None else None
    i32 = torch.int32
    Pm = max(P, 1)
    nb = lib.gsr_num_blocks(P)
    tiles = lib.gsr_num_tiles(H, W)
    mode = mode or rc.forward_mode
    hint = ws.hint.get((P, H, W)) if mode == "auto" else None
    capture = batch is not None and batch.get("capture") is not None
    if capture:           # graph capture (graph.py): fixed capacity, no host reads at all, nothing learned from this call
        hint = int(batch["capture"]["cap"])
    if batch is not None and hint is None:
        raise RuntimeError("batched forward needs a capacity hint (render the views once unbatched first)")

    with (_NULL_CTX if (batch is not None and batch.get("device_set")) else torch.cuda.device(dev)):
        radii = torch.empty(Pm, dtype=i32, device=dev)
        color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        depth_alpha = torch.empty((2, H, W), dtype=torch.float32, device=dev)
        if s.score_flag and batch is not None and batch.get("score") is not None:
            score = batch["score"]       # the batch's views add into the caller's [P] buffer (views.importance_scores)
        else:
            score = torch.zeros(P, dtype=torch.float32, device=dev) if s.score_flag else None
        proj_bytes = int(lib.gsr_project_scratch_bytes(P))
        proj_scratch = batch["scratch"] if batch is not None else ws.scratch("proj_scratch", proj_bytes)

        # forward only (the caller knows no backward can follow: torch.no_grad() / nothing requires a gradient): K6 writes no
        # checkpoints (GsrImages.ckpt = NULL) and their region of the state buffer -- the largest -- is not allocated
        fwd_only = bool(getattr(rc, "_forward_only", False)) and not capture
        seg_req = (batch.get("seg_len") if batch is not None else None) or seg_len or rc.seg_len
        if seg_req and int(seg_req) not in (64, 128, 256):
            raise ValueError(f"seg_len must be 64, 128 or 256, got {seg_req}")

        def seg_of(cap):
            return int(seg_req) if seg_req else pick_seg_len(cap, 1)

        def alloc_state(cap):
            """One allocation for everything the backward re-reads (splat, tile counts, offsets, lists, ranges,
            final_T, n_contrib); carved by offsets, no per-tensor allocations."""
            seg = seg_of(cap)
            lkey = (Pm, nb, tiles, H, W, cap, seg, bool(want_keys), fwd_only)
            lay = _STATE_LAYOUTS.get(lkey)
            if lay is None:
                sizes = dict(splat=Pm * 48, tiles_touched=Pm * 4, block_offsets=(nb + 8) * 4, point_list=max(cap, 1) * 4,
                             ranges=tiles * 8, tile_work=(2 * tiles + 2 + 2 * (cap // seg + tiles)) * 4 + 64,
                             tile_depth=tiles * 4, ckpt=(0 if fwd_only else (cap // seg + 1) * 6 * 256 * 4), final_T=H * W * 4,
                             n_contrib=H * W * 4,
                             keys_sorted=(max(cap, 1) * 8 if want_keys else 0))
                offs, tot = {}, 0
                for k, sz in sizes.items():
                    offs[k] = tot
                    tot += _align(sz)
                if len(_STATE_LAYOUTS) > 64:
                    _STATE_LAYOUTS.clear()
                lay = _STATE_LAYOUTS[lkey] = (offs, tot)
            offs, tot = lay
            buf = torch.empty(tot, dtype=torch.uint8, device=dev)
            base = buf.data_ptr()
            return buf, {k: base + o for k, o in offs.items()}, offs

        geom = L.GsrGeom()
        b = L.GsrBinning()
        im = L.GsrImages()
        im.color, im.depth_alpha, im.important_score = color.data_ptr(), depth_alpha.data_ptr(), _ptr(score)

        def bind(ptrs, cap, on_device):
            geom.splat, geom.radii, geom.tiles_touched = ptrs["splat"], radii.data_ptr(), ptrs["tiles_touched"]
            geom.block_offsets = ptrs["block_offsets"]
            geom.scratch, geom.scratch_bytes = proj_scratch.data_ptr(), proj_scratch.numel()
            sort_bytes = int(lib.gsr_sort_scratch_bytes(cap, tiles))
            # batched: every view its own, equally spaced slice (emission and the ty pass of all views share launches)
            sort_scratch = batch["sort"](sort_bytes) if (batch is not None and on_device) else \
                ws.scratch("sort_scratch", sort_bytes)
            b.point_list, b.ranges, b.tile_work = ptrs["point_list"], ptrs["ranges"], ptrs["tile_work"]
            b.seg_len = seg_of(cap)
            b.bwd_items_cap = cap // b.seg_len + tiles
            b.keys_sorted = ptrs["keys_sorted"] if want_keys else None
            b.scratch, b.scratch_bytes, b.count_on_device = sort_scratch.data_ptr(), sort_scratch.numel(), int(on_device)
            # forward variant from the PREVIOUS view's statistics (complete by now; a wrong guess only costs speed):
            # whole-tile items when thousands of shallow tiles saturate the machine, quarter items otherwise
            if capture:
                b.fwd_mode = int(batch["capture"].get("fwd_mode", rc.fwd_variant or 0))
            else:
                last_active, last_n = ws.last_stats.get((P, H, W), (0, 0))
                act = int(ws.stats_pinned_np[0]) if last_active is None else last_active
                b.fwd_mode = int(rc.fwd_variant if rc.fwd_variant is not None else
                                 (act >= 2048 and last_n > 0 and last_n / max(act, 1) < 1024))
            if b.fwd_mode == 1:
                b.seg_len = 256      # the whole-tile forward checkpoints every 256 entries (the buffers above are large enough)
            b.stats_host = ws.stats_pinned.data_ptr()
            im.final_T, im.n_contrib, im.tile_depth = ptrs["final_T"], ptrs["n_contrib"], ptrs["tile_depth"]
            im.ckpt = None if fwd_only else ptrs["ckpt"]     # (forward only: no checkpoints written, none allocated)

        NSIZED = ("point_list", "ranges", "tile_work", "tile_depth", "ckpt", "final_T", "n_contrib", "keys_sorted")
        n_pairs = C.c_uint64(0)
        if hint is None:
            # exact two-phase forward: project (+ one host sync for N), then allocate exactly, then render
            buf, ptrs, offs = alloc_state(0)
            bind(ptrs, 0, False)
            L.check(lib.gsr_forward_project(C.byref(st.view), C.byref(g), C.byref(geom), C.byref(n_pairs), stream,
                                            prof), "gsr_forward_project")
            N = int(n_pairs.value)
            cap = N
            buf2, ptrs2, offs2 = alloc_state(cap)
            # keep the projected state (splat / tile counts / offsets) from the first buffer: re-point only the
            # N-sized and per-pixel regions into the second one
            for k in NSIZED:
                ptrs[k] = ptrs2[k]
            bind(ptrs, cap, False)
            L.check(lib.gsr_forward_render(C.byref(st.view), C.byref(geom), N, C.byref(b), C.byref(im), stream, prof),
                    "gsr_forward_render")
            keep_bufs = (buf, buf2)
            view_src = {k: (buf2, offs2[k]) if k in NSIZED else (buf, offs[k]) for k in offs}
        else:
            # capacity = 1.5x the recent maximum, rounded UP to a coarse grid (1/8 octave) so that consecutive calls
            # ask the caching allocator for identical block sizes (jittering sizes fragment it and end in
            # device-synchronising hipMalloc/hipFree calls: measured 3.5x slowdown after a few hundred views)
            want = int(hint * 1.5) + 65536
            q = max(65536, 1 << max(0, want.bit_length() - 4))
            cap = (want + q - 1) // q * q
            if capture:
                cap = hint                       # the caller chose the capacity
            buf, ptrs, offs = alloc_state(cap)
            bind(ptrs, cap, True)
            if batch is not None:
                # the caller projects (gsr_forward_project_batch) and renders (gsr_forward_render_batch) all views of
                # the batch in one go, then resumes this generator for the bookkeeping
                yield st.view, geom, g, b, im, cap
                pinned, pidx, event = batch["pinned"], batch["index"], batch["event"]
                pinned_np, n_words = batch.get("pinned_np"), int(batch.get("n_views", 1))
            else:
                pinned, pidx, event = ws.n_pinned, 0, ws.event
                pinned_np, n_words = ws.n_pinned_np, 1
                L.check(lib.gsr_forward_project_async(C.byref(st.view), C.byref(g), C.byref(geom), pinned.data_ptr(),
                                                      stream, prof), "gsr_forward_project_async")
                event.record(torch.cuda.current_stream(dev))
                L.check(lib.gsr_forward_render(C.byref(st.view), C.byref(geom), cap, C.byref(b), C.byref(im), stream,
                                               prof), "gsr_forward_render")
            if capture:
                # nothing may touch the host while the stream is being captured: the pair counts land in the pinned words
                # when the graph RUNS; the owner of the graph compares them with `cap` after every replay (graph.py)
                st.N = None
                st.geom, st.binning, st.images = geom, b, im
                color_alias, da_alias = color.detach(), depth_alpha.detach()
                st.keep = (means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, bg, vm, pm, cp, radii,
                           (buf,), color_alias, da_alias, sc_keep, sc_out)
                st.versions = []
                return dict(color=color, radii=(radii if P == Pm else radii[:P]), depth_alpha=depth_alpha, score=score, N=None, cap=cap,
                            **{"act_" + k: v for k, v in sc_out.items()}), st
            if batch is None or not batch.get("synced", [False])[0]:
                t_wait = time.perf_counter()
                # (the render is already enqueued behind the projection: the GPU stays busy)
                if pinned_np is not None:
                    _wait_pair_counts(pinned_np, n_words, event)
                else:
                    event.synchronize()
                if rc.host_stats is not None:
                    rc.host_stats.wait_s += time.perf_counter() - t_wait
                if batch is not None and "synced" in batch:
                    batch["synced"][0] = True     # one wait covers the pair counts of all views of the batch
            N = (int(pinned_np[pidx]) if pinned_np is not None else int(pinned[pidx].item())) if P > 0 else 0
            keep_bufs = (buf,)
            view_src = {k: (buf, offs[k]) for k in offs}
            if N >= (1 << 32):
                L.check(-2, "gsr_forward_render")
            if N > cap:
                # speculation too small: redo binning + render exactly (projection results are still valid)
                buf2, ptrs2, offs2 = alloc_state(N)
                for k in NSIZED:
                    ptrs[k] = ptrs2[k]
                    view_src[k] = (buf2, offs2[k])
                bind(ptrs, N, False)
                if score is not None and batch is not None and batch.get("score") is not None:
                    batch["score_dirty"][0] = True     # shared accumulator: the caller recomputes the whole batch
                elif score is not None:
                    score.zero_()
                L.check(lib.gsr_forward_render(C.byref(st.view), C.byref(geom), N, C.byref(b), C.byref(im), stream,
                                               prof), "gsr_forward_render")
                keep_bufs = (buf, buf2)
                cap = N
        st.N = N
        ws.last_stats[(P, H, W)] = (None, N)     # the tile count lands in stats_pinned asynchronously
        if mode == "auto":
            ws.hint[(P, H, W)] = max(N, int(ws.hint.get((P, H, W), 0) * 0.9))
    st.geom, st.binning, st.images = geom, b, im
    # scratch is dead after the forward; everything else is kept for backward
    b.scratch, b.scratch_bytes, b.keys_sorted = None, 0, None
    geom.scratch, geom.scratch_bytes, geom.sorted_idx = None, 0, None
    color_alias, da_alias = color.detach(), depth_alpha.detach()
    st.keep = (means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, bg, vm, pm, cp, radii,
               keep_bufs, color_alias, da_alias)
    _track(st, [("means3D", means3D), ("opacities", opacities), ("shs", shs), ("colors_precomp", colors_precomp),
                ("scales", scales), ("rotations", rotations), ("cov3D_precomp", cov3D_precomp), ("bg", bg),
                ("viewmatrix", vm), ("projmatrix", pm), ("campos", cp), ("rendered image", color_alias),
                ("depth_alpha", da_alias)] +
           ([(f"model {m} leaf {j}", t) for m, row in enumerate(sc_keep[:-1]) for j, t in enumerate(row)]
            if sc_keep is not None else []))
    # ^ backward re-reads the output image (suffix sums from checkpoints). DETACHED aliases on purpose: the objects
    #   returned to autograd acquire grad_fn -> ctx -> this state; keeping them here would close a reference cycle
    #   and defer every free to Python's cyclic GC (measured: memory bloat and 5x slowdown after ~500 views).
    out = dict(color=color, radii=(radii if P == Pm else radii[:P]), depth_alpha=depth_alpha, score=score, N=N,
               **{"act_" + k: v for k, v in sc_out.items()})
    if want_aux:
        def view(name, dtype, count, shape=None):
            bufk, off = view_src[name]
            esz = torch.empty(0, dtype=dtype).element_size()
            t = bufk[off:off + count * esz].view(dtype)
            return t.view(shape) if shape is not None else t
        out.update(splat=view("splat", torch.float32, P * 12, (P, 12)),
                   tiles_touched=view("tiles_touched", i32, P),
                   point_list=view("point_list", i32, N), ranges=view("ranges", i32, tiles * 2, (tiles, 2)),
                   final_T=view("final_T", torch.float32, H * W, (H, W)), n_contrib=view("n_contrib", i32, H * W, (H, W)),
                   keys_sorted=view("keys_sorted", torch.int64, N) if want_keys else None)
    return out, st


def _bind_stats(gr, stats, P: int, dev) -> None:
    """stats: None or the (max_radii2D, xyz_gradient_accum, denom) tensors K8 updates for this view's visible Gaussians."""
    if stats is None:
        return
    ptrs = []
    for t in stats:
        if t.numel() != P or t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
            raise ValueError("densify_stats tensors must be contiguous fp32 with P elements on the view's device")
        ptrs.append(t.data_ptr())
    gr.stat_max_radii2D, gr.stat_xyz_gradient_accum, gr.stat_denom = ptrs


def _stat_views(V: int, stats, stats_views) -> set:
    """The views of a V-view call whose densification statistics count: the last one unless told otherwise (the
    reference's trainers use the last view of the step only, object_trainer.py:386-390)."""
    if stats is None:
        return set()
    if stats_views is None:
        return {V - 1}
    if isinstance(stats_views, str):
        if stats_views != "all":
            raise ValueError("stats_views is None (last view), 'all' or a sequence of view indices")
        return set(range(V))
    sv = {int(k) % V for k in stats_views}
    return sv


def _check_versions(st: _State) -> None:
    """K7 / K8 re-read the forward's inputs and outputs through raw pointers (no save_for_backward), so autograd's own
    version check does not see them: an in-place edit between forward and backward would silently corrupt gradients."""
    for name, t, ver in st.versions:
        if t._version != ver:
            raise RuntimeError(f"rasterizer backward: `{name}` was modified in place after the forward (version {ver} -> "
                               f"{t._version}); its memory is re-read by the backward. Clone it before editing.")


def _track(st: _State, named) -> None:
    st.versions = [(n, t, t._version) for n, t in named if t is not None]


def _model_grad_rows(sc_struct, leaves_of, model_grads, accumulate, K, dev):
    """Per model: the 6 gradient tensors the kernels write (given ones are validated, missing ones allocated)."""
    f32 = torch.float32
    outs = []
    for m in range(sc_struct.n_models):
        leaves = leaves_of[m]
        given = model_grads[m] if model_grads is not None else (None,) * 6
        row = []
        for t, gt in zip(leaves, given):
            if t is None or (t.numel() == 0 and gt is None):
                row.append(None if t is None else torch.zeros_like(t))
                continue
            if gt is None:
                if accumulate:
                    raise ValueError("accumulate=True needs the gradient tensors to add to")
                gt = torch.empty_like(t)
            elif gt.shape != t.shape or gt.dtype != f32 or not gt.is_contiguous() or gt.device != dev:
                raise ValueError("model gradient tensors must match the raw leaves (shape, fp32, contiguous, device)")
            row.append(gt)
        outs.append(tuple(row))
    return outs



class _Scratch:
    """K7 -> K8 scratch of V views of P Gaussians, kept between calls under GsrGrads.scratch_clean: `partials` [V,P,32] (the
    per-Gaussian sums K7 adds to) and `reach` [V, (P+63)//64] int64 (one bit per Gaussian K7 marked) are ALL ZERO whenever no backward is
    in flight -- K8 zeroes what it consumed -- so a step launches no clear (96 MB of stores + as many of loads at 500 k
    Gaussians x 4 views). `dirty` guards the invariant: set while a call is being enqueued, cleared when it returned OK."""
    __slots__ = ("partials", "reach", "dirty", "lock")

    def __init__(self, V, P, dev):
        self.partials = torch.zeros((V, max(P, 1), L.GSR_PARTIAL_WORDS), dtype=torch.float32, device=dev)
        self.reach = torch.zeros((V, (max(P, 1) + 63) // 64), dtype=torch.int64, device=dev)
        self.dirty = False
        # held from _scratch_acquire until the K7 / K8 pair that uses the buffers has been ENQUEUED (_scratch_release): two
        # host threads on one stream cannot interleave K7(A), K7(B), K8(A) on the same sums
        self.lock = threading.Lock()

    def begin(self):
        """Exclusive use for one enqueue; re-zeroed first if a previous call failed half-way (dirty)."""
        self.lock.acquire()
        try:
            if self.dirty:
                self.partials.zero_()
                self.reach.zero_()
            self.dirty = True
        except BaseException:        # (a device error / OOM inside zero_(): the next backward on this stream must not deadlock)
            self.lock.release()
            raise
        return self

    def end(self, ok: bool):
        if ok:
            self.dirty = False
        self.lock.release()


_SCRATCH: "collections.OrderedDict" = collections.OrderedDict()
_SCRATCH_LOCK = threading.Lock()
_SCRATCH_MAX = 12      # (one per internal stream of the module + the batched / captured ones)


def _scratch_acquire(dev, V: int, P: int, private: bool = False) -> _Scratch:
    """The scratch of (device, current stream, V, P): calls on one stream are ordered, so they can share it; another stream
    gets its own. Allocated (zeroed) on first use. The caller brackets its enqueue with scratch.begin() (exclusive use;
    re-zeroes after a call that failed half-way) ... scratch.end(ok). private: a scratch of the caller's own, not from the cache (captured graphs keep theirs:
    a replay does not pass through here, and a graph may be replayed on another stream than the one it was made on).
    Memory that stays pinned: 128 B per Gaussian and view per cached key (up to _SCRATCH_MAX keys), e.g. 256 MB for 500 k x 4."""
    if private:
        return _Scratch(V, P, dev)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream, V, P)
    with _SCRATCH_LOCK:
        s = _SCRATCH.get(key)
        if s is None:
            # P changes with every densification: the buffers of this (device, stream, V) for another P are dead weight
            # (128 B per Gaussian and view: 256 MB at 500 k x 4) -- drop them before allocating the new one
            for old in [k for k in _SCRATCH if k[:3] == key[:3] and k[3] != P]:
                del _SCRATCH[old]
            s = _SCRATCH[key] = _Scratch(V, P, dev)
            while len(_SCRATCH) > _SCRATCH_MAX:
                _SCRATCH.popitem(last=False)
        else:
            _SCRATCH.move_to_end(key)
    return s


def _bind_scratch(gr, sc: _Scratch, k: int = 0):
    gr.partials = sc.partials[k].data_ptr()
    gr.reach = sc.reach[k].data_ptr()
    gr.scratch_clean = 1

def _backward_scene(st: _State, dL_dcolor, dL_ddepth_alpha, cam_grads: bool, model_grads, accumulate: bool,
                    dL_dscales_out=None, stats=None, profile=None) -> dict:
    """Backward of a scene forward: the parameter gradients are written (or, with accumulate, ADDED) straight into
    per-model tensors shaped like the raw leaves. model_grads: list of 6-tuples (None entries are allocated here)."""
    lib = L.load()
    dev, P, K = st.dev, st.P, st.K
    f32 = torch.float32
    sc_struct, sc_keep = st.scene
    sg = L.GsrSceneGrads()
    outs = _model_grad_rows(sc_struct, sc_keep, model_grads, accumulate, K, dev)
    for m, row in enumerate(outs):
        mg = sg.models[m]
        mg.xyz, mg.scaling, mg.rotation, mg.opacity = _ptr(row[0]), _ptr(row[1]), _ptr(row[2]), _ptr(row[3])
        mg.features_dc, mg.features_rest = _ptr(row[4]), (_ptr(row[5]) if K > 1 else None)
    o = dict(dL_dmeans2D=torch.empty((P, 3), dtype=f32, device=dev), model_grads=outs,
             dL_dview=torch.zeros(16, dtype=f32, device=dev) if cam_grads else None,
             dL_dproj=torch.zeros(16, dtype=f32, device=dev) if cam_grads else None,
             dL_dcampos=torch.zeros(3, dtype=f32, device=dev) if cam_grads else None)
    scratch = _scratch_acquire(dev, 1, P)
    gr = L.GsrGrads()
    gr.dL_dmeans2D = o["dL_dmeans2D"].data_ptr()
    gr.dL_dview, gr.dL_dproj, gr.dL_dcampos = _ptr(o["dL_dview"]), _ptr(o["dL_dproj"]), _ptr(o["dL_dcampos"])
    _bind_scratch(gr, scratch)
    gr.accumulate = int(bool(accumulate))
    dL_dscales_out = _prep(dL_dscales_out, "dL_dscales_out", dev)
    sg.dL_dscales_out = _ptr(dL_dscales_out)
    _bind_stats(gr, stats, P, dev)
    gr.scene = C.pointer(sg)
    ig = L.GsrImageGrads()
    ig.dL_dcolor, ig.dL_ddepth_alpha = dL_dcolor.data_ptr(), dL_ddepth_alpha.data_ptr()
    stream = torch.cuda.current_stream(dev).cuda_stream
    prof = profile.handle if profile is not None else None
    ok = False
    scratch.begin()
    try:
        with torch.cuda.device(dev):
            L.check(lib.gsr_backward(C.byref(st.view), C.byref(st.gauss), C.byref(st.geom), C.byref(st.binning),
                                     C.byref(st.images), C.byref(ig), C.byref(gr), stream, prof), "gsr_backward")
        ok = True
    finally:
        scratch.end(ok)
    return o


def _arena_regions(g, per_view_scales: bool = False):
    """The regions of a GradArena a backward over the Gaussians `g` (GsrGaussians) writes."""
    r = {"means3D", "opacities"}
    if g.shs:
        r.add("shs")
    if g.scales and not per_view_scales:
        r.add("scales")
    if g.rotations:
        r.add("rotations")
    return frozenset(r)


def _arena_zero_outside(arena, accumulate: bool, regions=None) -> int:
    """GsrGrads.zero_outside bit 0 for a backward that OVERWRITES `regions` of the arena (default all): their rows outside the
    reached bitmap are known to be zero (GradArena.zero_outside_ok) -- K8 clears what the bitmap names and nothing else."""
    if arena is None or accumulate or getattr(arena, "reached", None) is None:
        return 0
    ok = getattr(arena, "zero_outside_ok", None)
    return 1 if (ok is not None and ok(regions)) else 0


def _arena_written(arena, accumulate: bool, token=None, regions=None) -> None:
    """Bookkeeping behind a HIP backward into the arena: K8 left the bitmap of the rows it reached (OR-ed into it when
    accumulating) and, when it overwrote, zeros everywhere else IN THE REGIONS IT WRITES (default all; the others now sit under a
    bitmap that does not describe them). token: the persistent result dict whose per-view rows the new bitmap describes as well
    (an OR-ed bitmap stays a superset of whatever it described)."""
    if arena is None or getattr(arena, "reached", None) is None:
        return
    arena.reached_valid = True
    if not accumulate:
        arena.zero_outside_reached = True
        arena._maintained = frozenset(arena.views) if regions is None else frozenset(regions)
        arena._k8_version = arena.flat._version
        arena._mask_owner = token
    elif arena.flat._version != getattr(arena, "_k8_version", None):
        arena.zero_outside_reached = False        # (K8 added to rows a torch op had written)


def rasterize_backward_raw(st: _State, dL_dcolor, dL_ddepth_alpha, cam_grads: bool = False, arena=None,
                           accumulate: bool = False, model_grads=None, dL_dscales_out=None, stats=None,
                           profile=None) -> dict:
    lib = L.load()
    dev, P, K = st.dev, st.P, st.K
    _check_versions(st)
    dL_dcolor = _prep(dL_dcolor, "dL_dcolor", dev, align=4)           # (read pixel by pixel: no 16-byte requirement)
    dL_ddepth_alpha = _prep(dL_ddepth_alpha, "dL_ddepth_alpha", dev, align=4)
    if st.scene is not None:
        return _backward_scene(st, dL_dcolor, dL_ddepth_alpha, cam_grads, model_grads, accumulate, dL_dscales_out,
                               stats, profile)
    g = st.gauss
    f32 = torch.float32
    av = {}
    if arena is not None:
        if arena.P != P or (g.shs and arena.K != K) or arena.flat.device != dev:
            raise ValueError("GradArena does not match this view's (P, K, device)")
        av = arena.views

    def new(*shape, name=None):
        t = av.get(name)
        return t if t is not None else torch.empty(shape, dtype=f32, device=dev)
    o = dict(dL_dmeans3D=new(P, 3, name="means3D"), dL_dmeans2D=new(P, 3), dL_dopacities=new(P, 1, name="opacities"),
             dL_dshs=new(P, K, 3, name="shs") if g.shs else None, dL_dcolors=new(P, 3) if g.colors_precomp else None,
             dL_dscales=new(P, 3, name="scales") if g.scales else None,
             dL_drotations=new(P, 4, name="rotations") if g.rotations else None,
             dL_dcov3D=new(P, 6) if g.cov3D_precomp else None,
             dL_dview=torch.zeros(16, dtype=f32, device=dev) if cam_grads else None,
             dL_dproj=torch.zeros(16, dtype=f32, device=dev) if cam_grads else None,
             dL_dcampos=torch.zeros(3, dtype=f32, device=dev) if cam_grads else None)
    scratch = _scratch_acquire(dev, 1, P)
    gr = L.GsrGrads()
    for k, t in o.items():
        setattr(gr, k, _ptr(t))
    _bind_scratch(gr, scratch)
    gr.accumulate = int(bool(accumulate and arena is not None))
    capturing = False
    if arena is not None and getattr(arena, "reached", None) is not None:
        gr.reached_mask = arena.reached.data_ptr()      # K8 marks the rows an exchange has to move (GradArena.reached_rows)
        # (not under a capture: what is known now need not hold when the graph replays)
        capturing = torch.cuda.is_current_stream_capturing()
        gr.zero_outside = 0 if capturing else _arena_zero_outside(arena, gr.accumulate != 0, _arena_regions(g))
    _bind_stats(gr, stats, P, dev)
    ig = L.GsrImageGrads()
    ig.dL_dcolor, ig.dL_ddepth_alpha = dL_dcolor.data_ptr(), dL_ddepth_alpha.data_ptr()
    stream = torch.cuda.current_stream(dev).cuda_stream
    prof = profile.handle if profile is not None else None
    ok = False
    scratch.begin()
    try:
        with torch.cuda.device(dev):
            L.check(lib.gsr_backward(C.byref(st.view), C.byref(g), C.byref(st.geom), C.byref(st.binning),
                                     C.byref(st.images), C.byref(ig), C.byref(gr), stream, prof), "gsr_backward")
        ok = True
    finally:
        scratch.end(ok)
        if not ok and arena is not None:
            arena.touch()
    if capturing and arena is not None:
        arena.touch()          # (nothing ran yet; the graph's replays are the capturer's business: nothing is known about the arena)
    else:
        _arena_written(arena, gr.accumulate != 0, None, _arena_regions(g))
    return o


def rasterize_backward_views_scene_raw(states, dL_dcolors, dL_ddepth_alphas, model_grads=None, accumulate: bool = False,
                                       dL_dscales_outs=None, stats=None, stats_views=None, profile=None) -> dict:
    """Backward of several views of the same SCENE (raw leaves, rasterize_forward_raw(scene=...) per view or batched):
    K7 per view, one K8 pass over all views; the gradients of the raw leaves are the sums over the views, written to (or,
    with accumulate, added to) one tensor per leaf. dL_dscales_outs: per view, the gradient arriving through the returned
    scales (or None). stats / stats_views: see RasterContext."""
    lib = L.load()
    V = len(states)
    st0 = states[0]
    dev, P, K = st0.dev, st0.P, st0.K
    f32 = torch.float32
    for st in states:
        _check_versions(st)
    sc0, keep0 = st0.scene
    outs = _model_grad_rows(sc0, keep0, model_grads, accumulate, K, dev)
    m2d = torch.empty((V, max(P, 1), 3), dtype=f32, device=dev)
    scratch = _scratch_acquire(dev, V, P)
    views = (L.GsrView * V)(*[st.view for st in states])
    gauss = (L.GsrGaussians * V)(*[st.gauss for st in states])
    geoms = (L.GsrGeom * V)(*[st.geom for st in states])
    bins = (L.GsrBinning * V)(*[st.binning for st in states])
    imgs = (L.GsrImages * V)(*[st.images for st in states])
    igs = (L.GsrImageGrads * V)()
    grs = (L.GsrGrads * V)()
    sgs = [L.GsrSceneGrads() for _ in range(V)]
    keep = []
    counted = _stat_views(V, stats, stats_views)
    for k in range(V):
        gc, gda = _prep(dL_dcolors[k], "dL_dcolor", dev, align=4), _prep(dL_ddepth_alphas[k], "dL_ddepth_alpha", dev, align=4)
        gso = _prep(dL_dscales_outs[k], "dL_dscales_out", dev) if dL_dscales_outs is not None else None
        keep += [gc, gda, gso]
        igs[k].dL_dcolor, igs[k].dL_ddepth_alpha = gc.data_ptr(), gda.data_ptr()
        for m, row in enumerate(outs):
            mg = sgs[k].models[m]
            mg.xyz, mg.scaling, mg.rotation, mg.opacity = _ptr(row[0]), _ptr(row[1]), _ptr(row[2]), _ptr(row[3])
            mg.features_dc, mg.features_rest = _ptr(row[4]), (_ptr(row[5]) if K > 1 else None)
        sgs[k].dL_dscales_out = _ptr(gso)
        grs[k].scene = C.pointer(sgs[k])
        grs[k].dL_dmeans2D = m2d[k].data_ptr()
        _bind_scratch(grs[k], scratch, k)
        grs[k].accumulate = int(bool(accumulate))
        if k in counted:
            _bind_stats(grs[k], stats, P, dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    prof = profile.handle if profile is not None else None
    ok = False
    scratch.begin()
    try:
        with torch.cuda.device(dev):
            L.check(lib.gsr_backward_views(V, views, gauss, geoms, bins, imgs, igs, grs, stream, prof), "gsr_backward_views")
        ok = True
    finally:
        scratch.end(ok)
    return dict(dL_dmeans2D=m2d[:, :P], model_grads=outs)


def rasterize_backward_views_raw(states, dL_dcolors, dL_ddepth_alphas, arena=None, accumulate: bool = False,
                                 stats=None, stats_views=None, per_view_scales: Optional[bool] = None,
                                 profile=None, reuse: Optional[dict] = None, private_scratch: bool = False,
                                 persistent: bool = False, trust_zeros: Optional[bool] = None) -> dict:
    """Backward of several views of the same Gaussians through gsr_backward_views: K7 per view, one K8 pass over all
    views. Returns the SUMMED parameter gradients (written to / added to the arena's views when given) and the per-view
    means2D gradients [V,P,3]. per_view_scales (default: whether the views' scales are different tensors): every view has
    its own scales tensor, `dL_dscales` is then [V,P,3]. reuse: the dict a previous call with the same arguments returned --
    its tensors receive the results again and nothing is allocated (graph capture, graph.py).
    persistent: the caller keeps the returned dict and hands it back as `reuse`, writing NOTHING into its tensors in between (or
    zeroing them all if something did): the dict then carries a reached-row bitmap of its own (without an arena) and calls with
    `reuse` run with GsrGrads.zero_outside -- K8 clears the rows the previous call reached and this one does not, instead of
    everything nothing reached (118 + 24 MB of zeros per 4-view step at C3). trust_zeros=False: never (the call clears every row
    nothing reached, whatever is known about the tensors); None (default): unless the stream is being captured -- what is known now
    need not hold when somebody else's graph replays (graph.py, which checks before every replay, says True explicitly)."""
    lib = L.load()
    V = len(states)
    st0 = states[0]
    dev, P, K = st0.dev, st0.P, st0.K
    g = st0.gauss
    f32 = torch.float32
    for st in states:
        _check_versions(st)
    av = {}
    if arena is not None:
        if arena.P != P or (g.shs and arena.K != K) or arena.flat.device != dev:
            raise ValueError("GradArena does not match this view's (P, K, device)")
        av = arena.views

    def new(*shape, name=None):
        t = av.get(name)
        return t if t is not None else torch.empty(shape, dtype=f32, device=dev)
    names = ("dL_dmeans3D", "dL_dopacities", "dL_dshs", "dL_dcolors", "dL_dscales", "dL_drotations", "dL_dcov3D")
    if per_view_scales is None:
        per_view_scales = any(st.gauss.scales != g.scales for st in states)
    own_mask = None
    if reuse is not None:
        o = {k: reuse[k] for k in names}
        m2d, scratch = reuse["_m2d"], reuse["_scratch"]
        own_mask = reuse.get("_reached")
    else:
        o = dict(dL_dmeans3D=new(P, 3, name="means3D"), dL_dopacities=new(P, 1, name="opacities"),
                 dL_dshs=new(P, K, 3, name="shs") if g.shs else None, dL_dcolors=new(P, 3) if g.colors_precomp else None,
                 dL_dscales=new(P, 3, name="scales") if g.scales else None,
                 dL_drotations=new(P, 4, name="rotations") if g.rotations else None,
                 dL_dcov3D=new(P, 6) if g.cov3D_precomp else None)
        if per_view_scales:        # every view has its own scales tensor -> its own scale gradient
            o["dL_dscales"] = torch.empty((V, P, 3), dtype=f32, device=dev)
        m2d = torch.empty((V, max(P, 1), 3), dtype=f32, device=dev)
        # (a reused dict -- graph capture -- keeps the scratch of the call that made it: static addresses, and the
        #  captured K7 / K8 pair maintains the all-zero invariant like an eager one; private_scratch: that call asks for a
        #  scratch of its own instead of the cached one eager calls share)
        scratch = _scratch_acquire(dev, V, P, private=private_scratch)
        if persistent and arena is None:
            own_mask = torch.zeros((P + 63) // 64, dtype=torch.int64, device=dev)
    # GsrGrads.zero_outside: bit 0 the summed gradients (the arena, or a persistent dict's own tensors), bit 1 the per-view rows
    # (only a persistent dict keeps those)
    acc = bool(accumulate and arena is not None)
    token = reuse.get("_token") if reuse is not None else (object() if persistent else None)
    has_mask = (arena is not None and getattr(arena, "reached", None) is not None) or own_mask is not None
    # (with an arena the bitmap is the ARENA's: it describes this dict's per-view rows only if this dict's call wrote it last)
    keeps = reuse is not None and token is not None and not acc and has_mask and \
        (arena is None or getattr(arena, "_mask_owner", None) is token)
    regions = _arena_regions(g, per_view_scales)
    zo = (_arena_zero_outside(arena, acc, regions) if arena is not None else (1 if keeps else 0)) | (2 if keeps else 0)
    if per_view_scales and V == 1 and not keeps:
        # ONE view with "per-view" scales: the library sees an ordinary single view, for which dL_dscales is one of the summed
        # outputs bit 0 speaks for -- but the buffer is this call's own fresh [1,P,3] tensor, not the arena's region (found by
        # tools/fuzz_views.py, seeds 90 / 160: uninitialised rows in dL/dscales)
        zo &= ~1
    foreign_capture = trust_zeros is None and torch.cuda.is_current_stream_capturing()
    if trust_zeros is None:
        trust_zeros = not foreign_capture
    if not trust_zeros:
        zo = 0
    views = (L.GsrView * V)(*[st.view for st in states])
    gauss = (L.GsrGaussians * V)(*[st.gauss for st in states])
    geoms = (L.GsrGeom * V)(*[st.geom for st in states])
    bins = (L.GsrBinning * V)(*[st.binning for st in states])
    imgs = (L.GsrImages * V)(*[st.images for st in states])
    igs = (L.GsrImageGrads * V)()
    grs = (L.GsrGrads * V)()
    keep = []
    counted = _stat_views(V, stats, stats_views)
    for k in range(V):
        gc, gda = _prep(dL_dcolors[k], "dL_dcolor", dev, align=4), _prep(dL_ddepth_alphas[k], "dL_ddepth_alpha", dev, align=4)
        keep += [gc, gda]
        igs[k].dL_dcolor, igs[k].dL_ddepth_alpha = gc.data_ptr(), gda.data_ptr()
        for name, t in o.items():
            if name.startswith("_"):
                continue
            setattr(grs[k], name, _ptr(t))
        if per_view_scales:
            grs[k].dL_dscales = o["dL_dscales"][k].data_ptr()
        grs[k].dL_dmeans2D = m2d[k].data_ptr()
        _bind_scratch(grs[k], scratch, k)
        grs[k].accumulate = int(acc)
        if arena is not None and getattr(arena, "reached", None) is not None:
            grs[k].reached_mask = arena.reached.data_ptr()
            grs[k].zero_outside = zo
        elif own_mask is not None:
            grs[k].reached_mask = own_mask.data_ptr()
            grs[k].zero_outside = zo
        if k in counted:
            _bind_stats(grs[k], stats, P, dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    prof = profile.handle if profile is not None else None
    ok = False
    scratch.begin()        # (a reused dict's scratch as well: same exclusivity, same re-zeroing after a failed call)
    try:
        with torch.cuda.device(dev):
            L.check(lib.gsr_backward_views(V, views, gauss, geoms, bins, imgs, igs, grs, stream, prof),
                    "gsr_backward_views")
        ok = True
    finally:
        scratch.end(ok)
        if not ok and arena is not None:
            arena.touch()
    if foreign_capture and arena is not None:
        arena.touch()          # (nothing ran; when and how often the graph will run is the capturer's business: nothing is known)
    else:
        _arena_written(arena, acc, token, regions)
    o["dL_dmeans2D"] = m2d[:, :P]
    o["_m2d"], o["_scratch"] = m2d, scratch
    o["_reached"], o["_token"], o["_zero_outside"], o["_regions"] = own_mask, token, zo, regions
    return o


# ---- consecutive per-view calls on internal streams ("next round" 1a) -----------------------------------------
# One GaussianRasterizer call is a chain of ~17 dependent launches of which the ~140 us binning part leaves the GPU idle
# (launch / visibility latency on 12 MB of keys). The unmodified trainers make one call per view; when several calls follow
# each other without the caller touching their results in between, view j + 1's K1 + binning can run under view j's K6 -- IF
# the calls are not all on one stream. So the module rotates its calls over n internal streams:
#   fork   the internal stream waits for an event recorded on the CALLER'S stream: everything the caller enqueued before the
#          call (the producers of the inputs) is finished before K1 reads them;
#   join   the caller's stream waits for the event recorded behind K6 on the internal stream before the call returns: whatever
#          the caller enqueues next sees finished outputs. Nothing is deferred: stream semantics are exactly those of a call
#          on the caller's own stream.
# The join alone would serialise consecutive calls again: the NEXT call's fork event would sit behind this call's join on
# the caller's stream. It does not have to: an input tensor that is the same live tensor OBJECT, at the same autograd version
# counter, as at an earlier call of this module on this caller stream has not been written since (every in-place torch op
# bumps the counter; raw-pointer writers -- FusedAdam -- bump it explicitly), so it was complete at THAT call's fork event,
# which precedes the previous call's join on the caller's stream. When that holds for EVERY input of a call (the parameter
# tensors and the camera tensors of the settings), the call forks from the older event and starts while the previous views
# are still running; one new tensor (fresh activations, a freshly built camera) and the call forks from "now". Sound by
# construction, and only as effective as the caller lets it be: the reference's trainers re-run the activations before every
# call and read the outputs right behind it on the host (`disp[alpha <= 0.1]`, scene_gaussian.py:1027), so THEIR forwards
# serialise whatever this module does; a loop over cameras against persistent tensors (bench.py's `dropin_views_per_s`, the
# importance-score loop, video_inference) overlaps.
# Forward only: the internal stream is made current INSIDE the autograd Function's forward and left again before it returns,
# so autograd records the caller's stream for the node and the backward (K7 / K8) runs on the caller's stream as always.
# (Round 5 measured the other way first -- the whole call under the internal stream, autograd running every backward node on
# it with its own cross-stream syncs: forward-only 4 480 -> 6 120 views/s at C3, but four forwards + four backwards 2 715 ->
# 2 580: every backward paid two cross-stream hand-offs and won nothing, the next view's upstream gradient sits behind this
# view's results on the caller's stream anyway.)
# Memory: tensors allocated inside the forward belong to the internal stream's allocator pool -- which is what makes the early
# fork safe (a block of the CALLER'S pool may still be read by kernels the caller enqueued after the old fork event) -- and
# everything the caller's stream will touch (the outputs, the state the backward re-reads) is record_stream()ed on it; the
# inputs are record_stream()ed on the internal stream.
SIDE_STREAMS_DEFAULT = 0
SIDE_STREAMS_MAX = 8


PER_VIEW_ACCELS = ("off", "streams", "graphs")


def per_view_accel(context) -> str:
    """The one accelerator of the per-view call this context (or, with None fields, the environment) selects: "off", "streams"
    or "graphs". Both at once is an error, never a silent precedence."""
    mode = getattr(context, "per_view_accel", None) if context is not None else None
    if mode is None:
        mode = os.environ.get("GSR_PER_VIEW_ACCEL") or None
    if mode is not None:
        if mode not in PER_VIEW_ACCELS:
            raise ValueError(f"per_view_accel is one of {PER_VIEW_ACCELS}, not {mode!r}")
        return mode
    from . import dropin
    g = getattr(context, "dropin_graphs", None) if context is not None else None
    graphs = dropin.ENABLED if g is None else bool(g)
    streams = _side_streams_wanted(context) > 1
    if graphs and streams:
        raise ValueError("internal streams (side_streams / GSR_SIDE_STREAMS) and the captured ring (dropin_graphs / "
                         "GSR_DROPIN_GRAPHS) are mutually exclusive: choose one (RasterContext.per_view_accel)")
    return "graphs" if graphs else ("streams" if streams else "off")


def _side_streams_wanted(context) -> int:
    n = context.side_streams if (context is not None and context.side_streams is not None) else None
    if n is None:
        env = os.environ.get("GSR_SIDE_STREAMS")
        n = int(env) if env else SIDE_STREAMS_DEFAULT
    return max(0, min(int(n), SIDE_STREAMS_MAX))


class _SideStreams:
    """Per (device, caller stream): the internal streams, the latest fork event on the caller's stream and the tensors known
    to have been complete at (or before) it."""
    __slots__ = ("streams", "joins", "turn", "fork", "known", "stats")

    def __init__(self, dev, n):
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
        self.joins = [torch.cuda.Event() for _ in range(n)]
        self.turn = 0
        self.fork = None
        self.known = {}          # id(tensor) -> (weakref to it, its version when it was seen complete)
        self.stats = dict(calls=0, forks=0, reused_forks=0)

    def proves_complete(self, inputs) -> bool:
        """Every input is the same live tensor object, at the same version counter, as when it was last seen complete."""
        known = self.known
        try:
            for t in inputs:
                e = known.get(id(t))
                if e is None or e[0]() is not t or e[1] != t._version or e[2] != t.data_ptr():
                    return False
        except RuntimeError:             # (inference-mode tensors have no version counter: never provably unchanged)
            return False
        return True

    def remember(self, inputs) -> None:
        known = self.known
        if len(known) > 512:                     # (ids of dead tensors accumulate: start over, one serialised call)
            known.clear()
        try:
            for t in inputs:
                known[id(t)] = (weakref.ref(t), t._version, t.data_ptr())
        except RuntimeError:             # inference-mode tensor: forget everything, the next call forks from "now"
            known.clear()

    def fork_event(self, cur, inputs):
        if self.fork is not None and self.proves_complete(inputs):
            self.stats["reused_forks"] += 1
        else:
            ev = torch.cuda.Event()
            ev.record(cur)
            self.fork = ev
            self.stats["forks"] += 1
        # every input of this call is complete at self.fork (proved above, or because the event was recorded just now behind
        # everything the caller enqueued); entries of earlier calls stay valid: their events precede this one on the stream
        self.remember(inputs)
        return self.fork


_SIDE = {}
_SIDE_LOCK = threading.Lock()


def _side_for(dev, cur, n) -> _SideStreams:
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), cur.cuda_stream, n)
    sd = _SIDE.get(key)
    if sd is None:
        with _SIDE_LOCK:
            sd = _SIDE.get(key)
            if sd is None:
                sd = _SIDE[key] = _SideStreams(dev, n)
    return sd


def side_stream_stats() -> dict:
    return {f"{k[0]}:{k[1]:#x}:{k[2]}": dict(v.stats) for k, v in _SIDE.items()}


def _call_on_side_stream(n, call, tensors, settings):
    """call() with an internal stream current; see the comment block above. tensors: the per-Gaussian inputs (None entries
    skipped); the camera tensors come from `settings`."""
    dev = tensors[0].device
    cur = torch.cuda.current_stream(dev)
    sd = _side_for(dev, cur, n)
    inputs = [t for t in tensors if t is not None] + [settings.bg, settings.viewmatrix, settings.projmatrix, settings.campos]
    k = sd.turn
    sd.turn = (k + 1) % n
    side = sd.streams[k]
    sd.stats["calls"] += 1
    side.wait_event(sd.fork_event(cur, inputs))
    for t in inputs:
        if t.is_cuda:
            t.record_stream(side)         # (their memory must not be handed out again while the internal stream reads it)
    out = call(side)                      # (the Function's forward makes `side` current around its launches and allocations)
    j = sd.joins[k]
    j.record(side)
    cur.wait_event(j)
    return out


class _RasterizeGaussians(torch.autograd.Function):
    """viewmatrix / projmatrix / campos are the settings' own tensors, passed once more as explicit inputs so that autograd
    can deliver dL/dviewmatrix (+ projmatrix, campos) to callers that optimise the camera (BASELINE.json north_star); the
    reference's call sites never ask for them (camera tensors without requires_grad, utils/cam_utils.py:196-210)."""

    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix,
                projmatrix, campos, settings, rc):
        side = getattr(rc, "_side", None)
        if side is None:
            out, st = rasterize_forward_raw(settings, means3D, opacities, shs, colors_precomp, scales, rotations,
                                            cov3D_precomp, want_aux=False, rc=rc)
        else:
            # an internal stream of the module (_SideStreams): current for the launches and allocations of the forward only
            cur = torch.cuda.current_stream(means3D.device)
            with torch.cuda.stream(side):
                out, st = rasterize_forward_raw(settings, means3D, opacities, shs, colors_precomp, scales, rotations,
                                                cov3D_precomp, want_aux=False, rc=rc)
            # Everything the caller's stream will read that may have been ALLOCATED inside this forward belongs to the internal
            # stream's pool: the outputs, the state buffers, and every copy _prep() made of an input -- .contiguous() / .float()
            # of bg / viewmatrix / projmatrix / campos (the reference builds its cameras as RT.transpose(0, 1).cuda(),
            # utils/cam_utils.py:197: non-contiguous, so a fresh copy per call) or of a parameter tensor. K7 / K8 read them
            # through raw pointers on the caller's stream; without record_stream the block would return to the internal pool
            # when ctx dies and the next forward on that stream could overwrite it under a backward still in flight.
            # (record_stream on a tensor of the caller's own pool -- an input passed through untouched -- is harmless.)
            def _mark(x):
                if isinstance(x, torch.Tensor):
                    if x.is_cuda:
                        x.record_stream(cur)
                elif isinstance(x, (tuple, list)):
                    for y in x:
                        _mark(y)
            _mark((out["color"], out["radii"], out["depth_alpha"], out["score"]))
            _mark(st.keep)
        ctx.st, ctx.rc = st, rc
        ctx.opac_shape = opacities.shape
        ctx.cam_shapes = (viewmatrix.shape, projmatrix.shape, campos.shape)
        ctx.cam_grads = any(ctx.needs_input_grad[8:11])
        ctx.set_materialize_grads(False)       # no zero tensors for outputs nobody differentiates (radii is [P] int32)
        ctx.mark_non_differentiable(out["radii"])
        if settings.score_flag:
            ctx.mark_non_differentiable(out["score"])
            return out["score"], out["color"], out["radii"], out["depth_alpha"]
        return out["color"], out["radii"], out["depth_alpha"]

    @staticmethod
    def backward(ctx, *grads):
        st, rc = ctx.st, ctx.rc
        if len(grads) == 4:
            _, g_color, _, g_da = grads
        else:
            g_color, _, g_da = grads
        H, W = st.view.image_height, st.view.image_width
        if g_color is None:
            g_color = torch.zeros((3, H, W), dtype=torch.float32, device=st.dev)
        if g_da is None:
            g_da = torch.zeros((2, H, W), dtype=torch.float32, device=st.dev)
        o = rasterize_backward_raw(st, g_color, g_da, cam_grads=ctx.cam_grads, arena=rc.grad_arena,
                                   accumulate=rc.accumulate, stats=rc.densify_stats, profile=rc.profile)
        cam = (None, None, None)
        if ctx.cam_grads:
            cam = tuple(o[k].reshape(sh) for k, sh in zip(("dL_dview", "dL_dproj", "dL_dcampos"), ctx.cam_shapes))
        if rc.grad_arena is not None:      # the parameter gradients live in the arena, not in .grad (RasterContext)
            return (None, o["dL_dmeans2D"], None, o["dL_dcolors"], None, None, None, o["dL_dcov3D"], *cam, None, None)
        return (o["dL_dmeans3D"], o["dL_dmeans2D"], o["dL_dshs"], o["dL_dcolors"],
                o["dL_dopacities"].reshape(ctx.opac_shape), o["dL_dscales"], o["dL_drotations"], o["dL_dcov3D"],
                *cam, None, None)


class GaussianRasterizer(torch.nn.Module):
    """Drop-in for diff_gaussian_rasterization.GaussianRasterizer (scene_gaussian.py:966, 1012-1021). `context` (optional,
    not part of the reference's interface) carries the extras of this repo -- see RasterContext."""

    def __init__(self, raster_settings: GaussianRasterizationSettings, context: Optional[RasterContext] = None):
        super().__init__()
        self.raster_settings = raster_settings
        self.context = context

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        s = self.raster_settings
        ctx = self.context
        accel = per_view_accel(ctx)
        if accel == "graphs":
            from . import dropin
            if dropin.eligible(s, means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, ctx,
                               wanted=True):
                out = dropin.rasterize(s, means3D, means2D, opacities, shs, scales, rotations, ctx)
                if out is not None:
                    return out

        # no backward can follow (torch.no_grad(), or nothing that requires a gradient): the forward writes no checkpoints
        forward_only = not (torch.is_grad_enabled() and any(
            t is not None and t.requires_grad for t in (means3D, means2D, opacities, shs, colors_precomp, scales, rotations,
                                                        cov3D_precomp, s.viewmatrix, s.projmatrix, s.campos)))

        def call(side=None):
            rc = (ctx or DEFAULT_CONTEXT).snapshot()
            rc._side = side
            rc._forward_only = forward_only
            return _RasterizeGaussians.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                             cov3D_precomp, s.viewmatrix, s.projmatrix, s.campos, s, rc)
        n = max(_side_streams_wanted(ctx), 2) if accel == "streams" else 0
        # (a profile times stages with events on ONE stream; a stream being captured stays as it is; the captured ring above has
        #  its own way of cutting the host's share and is never combined with the internal streams: per_view_accel. An arena /
        #  densification statistics are written by the BACKWARD, which stays on the caller's stream: no restriction.)
        if n > 1 and means3D.is_cuda and (ctx is None or ctx.profile is None) \
                and not torch.cuda.is_current_stream_capturing():
            return _call_on_side_stream(n, call, (means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp), s)
        return call()
