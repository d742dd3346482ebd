#!/usr/bin/env python
"""bench.py -- fwd+bwd views/s of the MI355X-native Gaussian rasterizer on BASELINE.json's metric config.

A "step" = one pass of the hot path over one batch of views: every GPU renders `--views-per-step` views
(default 4 = the reference's C_batch_size, configs/objects/sample.yaml:60, training/object_trainer.py:302-382:
4 views are rendered and their gradients accumulated before each optimizer step), forward (K1-K6) + backward (K7-K8)
from fixed upstream gradients on image and depth_alpha. The per-view parameter gradients are summed on the device and,
with N GPUs, the sums are combined by ONE exchange per step (weak scaling: per-GPU work is fixed).

TWO figures, both in the line:
  * `value`: the V views of a step through ONE batched call (`GaussianRasterizerViews`, this repo's extension of the
    reference's interface: the views of a step share K1 / K8 and the sort launches) -- `config.batched_call` = true;
  * `dropin_views_per_s`: the same V views through the reference's own interface, one `GaussianRasterizer(...)` call per
    view, exactly what the unmodified trainers do (scene_gaussian.py:966-1021) -- the drop-in boundary's number.
`--unbatched` makes the drop-in path the headline instead. Inputs are synthetic (dreamscene_amd/synth.py, SURVEY.md
8d), resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` for the dominant kernel (HIP events on
the launch stream, via the library's GsrProfile; VALU issue fraction from the committed SQ counters) and `cpu_baseline`
(the C port of the same algorithm on ALL host cores, the same on one thread, and the PyTorch-CPU oracle on all cores at
C1 / C2; N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
SIMDS = 1024            # 256 CUs x 4 SIMDs
# Issue time one wave64 instruction costs its SIMD, MEASURED on this part at 8 waves per SIMD (tools/probe/valu_rate*.hip,
# profiles/r03_valu_rate.txt), ns: plain fp32 / integer add-mul-fma-mov 1.22 (v_pk_fma_f32: 2.36 -- packed fp32 does NOT
# raise the fp32 rate on gfx950, 107 TFLOP/s either way); min / max / med3 / cmp / cndmask / cvt / ldexp / rndne / shifts /
# every DPP or SGPR-operand form 1.85; v_exp / v_rcp / v_permlane*_swap 3.5; an SALU instruction 0.8 (it shares the issue port)
ISSUE_NS = {"plain": 1.22, "other": 1.85, "trans": 3.5, "salu": 0.8}
# static mix of the VALU instructions of the compositing loops (by class, from the ISA of render.hip's hot loops)
# (K7 per (8x8 block, splat): ~86 instructions, of which 18 DPP adds, 3 compares, min, rndne, cvt, ldexp = "other", the
#  reciprocal and 2 permlane swaps = "trans"; K6 per step: ~49, of which 6 DPP, 4 compares, 4 selects, min, rndne, cvt, ldexp)
VALU_MIX = {"render_bwd": {"plain": 0.665, "other": 0.30, "trans": 0.035},
            "render_fwd": {"plain": 0.61, "other": 0.39, "trans": 0.0}}
# stage -> the kernel whose launches the stage timer brackets (for the counters in profiles/traffic.json)
STAGE_KERNEL = {"preprocess": "k_preprocess", "preprocess_bwd": "k_preprocess_bwd", "render_fwd": "k_render_fwd",
                "render_bwd": "k_render_bwd", "duplicate": "k_emit"}


def algorithmic_bytes(stage: str, P: int, N: int, HW: int, K: int, D: int, views: int = 1) -> float:
    """Algorithmic HBM bytes of ONE LAUNCH of each stage covering `views` views of the same Gaussians (SURVEY.md section 8d;
    DESIGN.md 'bytes per unit'). The batched K1 / K8 read the parameter rows (and K8 writes the summed parameter
    gradients) once per launch whatever the number of views; everything else is per view."""
    S = 12 * (D + 1) ** 2
    V = views
    return {
        "preprocess": P * (44 + S) + V * P * 48,
        "scan": V * (P * 4 / 256 * 2),
        "duplicate": V * (P * 20 + N * 12),
        "sort": V * (N * 24),                 # lower bound: one read + one write of (u64 key, u32 value)
        "ranges": V * (N * 8),
        "render_fwd": V * (N * 44 + HW * 28),
        "render_bwd": V * (N * 44 + HW * 28 + P * 40),
        "preprocess_bwd": P * (44 + S) + V * P * 40 + V * P * 12 + P * (44 + 12 * K),
    }[stage]


def timed_kernel_name(stage: str, K: int, seg: int, batched: bool):
    """Name (as tools/profile_digest.py shortens it) of the template instance a stage's timed launch runs."""
    if stage == "render_bwd":
        return f"k_render_bwd<{seg}>"
    if stage == "render_fwd":
        return f"k_render_fwd<false, {seg}>"
    if stage == "preprocess":
        return f"k_preprocess_views<{K}>" if batched else f"k_preprocess<{K},"
    if stage == "preprocess_bwd":
        return f"k_preprocess_bwd_views<{K},"
    return None


def pick_kernel(names, want, prefix, stage):
    """The counters' entry of the kernel that was timed: the exact instance; an instance-name PREFIX (K8: the trailing template
    arguments depend on the launch) when it is unambiguous; the stage's kernel prefix only when ONE kernel carries it. Never a
    max over several instances (round 4 printed k_render_bwd<128>'s 485 MB for the timed k_render_bwd<256>'s 383 MB)."""
    if want:
        if want in names:
            return want
        c = [n for n in names if n.startswith(want)]
        if len(c) == 1:
            return c[0]
    if prefix:
        c = [n for n in names if n.startswith(prefix) and (stage != "preprocess" or "bwd" not in n)]
        if len(c) == 1:
            return c[0]
    return None


def self_launch_cmd(n_gpus: int, argv, port: int):
    """The command `python bench.py --gpus N ...` turns itself into when no launcher set up the ranks: the same line the
    task contract says the driver uses (one process per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_gpus)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def self_launch(n_gpus: int, argv) -> None:
    import socket
    share = os.environ.get("GSR_BENCH_SHARE_GPU", "0") == "1"
    n_dev = torch.cuda.device_count()
    if not share and n_dev < n_gpus:
        raise SystemExit(f"bench.py --gpus {n_gpus}: this node exposes {n_dev} GPU(s); one rank per GPU is the only "
                         "measurement configuration (GSR_BENCH_BACKEND=gloo GSR_BENCH_SHARE_GPU=1 runs the ranks on one "
                         "device for functional tests)")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.stdout.flush()
    os.execve(sys.executable, self_launch_cmd(n_gpus, argv, port), env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gaussians", type=int, default=500_000)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--scene", choices=["object", "indoor"], default="object")
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--views-per-step", type=int, default=4)
    ap.add_argument("--init-opacity", action="store_true",
                    help="object scene in the reference's initial state: every Gaussian at opacity 0.1 "
                         "(gs_renderer.py:598; ~87 layers blend before T < 1e-4 stops a pixel)")
    ap.add_argument("--exchange", choices=["measure", "dense", "auto", "rows", "sparse_rs", "direct"], default="measure",
                    help="wire format of the multi-GPU gradient exchange (multiview.GradExchange); dense = one in-place "
                         "all-reduce of the active columns, the only format that needs no host read; measure (default) = "
                         "after the warm-up time the formats of --exchange-candidates for a few steps each and keep the fastest "
                         "(every rank takes the same decision); with one rank there is nothing to exchange")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true",
                    help="skip the drop-in measurement after the timed region (profiling runs: keeps the per-kernel "
                         "statistics of the batched launches free of single-view launches)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="after the --steps region: keep stepping for at least this long and report `sustained_views_per_s` "
                         "(0 = skip)")
    ap.add_argument("--rotate-seconds", type=float, default=2.0,
                    help="then: a run that renders 4 NEW cameras (of 64 sampled like the reference's random cameras) every "
                         "step and applies a fused Adam update in between, for at least this long -> `rotating_cameras` "
                         "(0 = skip)")
    ap.add_argument("--train-seconds", type=float, default=1.2,
                    help="then: `training_like` (what object_render(test=False) hands the rasterizer: fresh scale noise per "
                         "view = scales [V,P,3], SH degree 0 with probability 0.1, random / black background with probability "
                         "0.5; scene_gaussian.py:938-947, 1004-1008, config.py:20-23), `init_state` (every opacity 0.1, "
                         "gs_renderer.py:598) and `forward_only` (video_inference, object_trainer.py:81-118), each for at "
                         "least this long (0 = skip)")
    ap.add_argument("--fwd-mode", type=int, default=None, help="force the forward compositing variant (0 / 1)")
    ap.add_argument("--capture", choices=["auto", "on", "off"], default="auto",
                    help="batched call: replay the step's launches from captured hipGraphs (graph.CapturedViews: 2 graph "
                         "launches per step) or issue its ~45 launches from Python (GaussianRasterizerViews). auto = time "
                         "both for a few steps after the warm-up and keep the faster one (big workloads are GPU-bound either "
                         "way and lose ~1 %% to graph boundaries; small ones are host-bound without graphs)")
    ap.add_argument("--no-capture", action="store_true", help="same as --capture off")
    ap.add_argument("--unbatched", action="store_true",
                    help="render the views of a step one call at a time (GaussianRasterizer) instead of through "
                         "GaussianRasterizerViews (same kernels; the depth sorts of all views share their launches)")
    ap.add_argument("--exchange-candidates", default=None,
                    help="--exchange measure: comma-separated wire formats to time (default: dense,direct,rows,sparse_rs under "
                         "RCCL -- the sparse formats in their host-read-free message form, 84-90 us of device side at C3, "
                         "DESIGN.md section 6 --, dense,rows under gloo)")
    ap.add_argument("--exchange-probe-steps", type=int, default=12,
                    help="--exchange measure: timed steps per wire format after the warm-up")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU under torch.distributed.run);
        # under the driver's own `python -m torch.distributed.run ... bench.py --gpus N` the environment is already there
        self_launch(args.gpus, sys.argv[1:])
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: measuring {world}",
              file=sys.stderr, flush=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GSR_BENCH_BACKEND=gloo + GSR_BENCH_SHARE_GPU=1: every rank on cuda:0 with the gloo backend -- lets the whole
    # multi-rank control flow (view sharding, arena, exchange, barriers, max over ranks) run on a ONE-GPU box
    # (tests/test_multirank_gpu.py); RCCL refuses two ranks on one device. Not a measurement configuration.
    backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")
    share_gpu = os.environ.get("GSR_BENCH_SHARE_GPU", "0") == "1"
    dev = torch.device("cuda", 0 if share_gpu else local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get("GSR_BENCH_PG_TIMEOUT_S", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
    torch.cuda.set_device(dev)

    def allreduce_scalars(vals, op):
        """Small host-side reductions of the harness (timings, decisions) -- through the host under gloo."""
        tt = torch.tensor(vals, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=op)
        return [float(x) for x in tt.tolist()]

    from dreamscene_amd import _lib, multiview, rasterizer as R, synth
    from dreamscene_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, RasterContext

    _lib.load()   # fail loudly if the HIP library is missing: there is no fallback
    H = W = args.res
    V = max(1, args.views_per_step)
    batched = V > 1 and not args.unbatched
    cap_mode = "off" if (args.no_capture or not batched) else args.capture
    use_capture = [cap_mode in ("on", "auto")]       # (a cell: "auto" decides after the warm-up)
    how = (f"{V} views/step through ONE batched call" if batched else
           f"{V} views/step, one GaussianRasterizer call per view (drop-in interface)")
    if args.scene == "object":
        K, D = 16, args.sh_degree
        g = synth.g_object(args.gaussians, seed=0, K=K, init_opacity=args.init_opacity)
        cams = synth.object_cameras(8, H, W)
        workload = (f"C3{'-init (all opacities 0.1)' if args.init_opacity else ''}: G-object {args.gaussians} Gaussians "
                    f"(K=16, SH degree {D}), orbit cameras @{W}x{H}, fwd+bwd, {how}")
    else:
        K, D = 4, 1
        g = synth.g_indoor(seed=0, per_wall=max(1, args.gaussians // 5), K=K)
        cams = synth.indoor_cameras(8, H, W)
        workload = f"G-indoor {g['means3D'].shape[0]} Gaussians (K=4, SH degree 1), in-room cameras @{W}x{H}, fwd+bwd, {how}"
    P = g["means3D"].shape[0]
    params = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in g.items()}
    gi_np, gda_np = synth.upstream_grads(H, W, seed=rank)
    gi, gda = torch.tensor(gi_np, device=dev), torch.tensor(gda_np, device=dev)
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
    # view j of rank r: camera (r * V + j) of the orbit; view 0 of rank 0 is the C3 camera
    my_cams = [cams[(rank * V + j) % len(cams)] for j in range(V)]
    cam = my_cams[0]
    arena = multiview.GradArena(P, K, dev)
    exchange = multiview.GradExchange(arena, sh_degree=D, mode="dense" if args.exchange == "measure" else args.exchange)
    prof_holder = [None]       # the contexts below share one profile slot (set for the stage pass / the timed region)
    host_stats = R.HostStats()   # seconds blocked on the pair counts, summed over the calls of the contexts below

    def ctx(accumulate):
        # view 0 of a step overwrites the arena, views 1.. are added on the device; the parameter gradients live in the
        # arena (what the exchange works on), autograd only delivers means2D.grad
        return RasterContext(grad_arena=arena, accumulate=accumulate, fwd_variant=args.fwd_mode, profile=prof_holder[0],
                             host_stats=host_stats)

    settings_list = [GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=t([1.0, 1.0, 1.0]),
        scale_modifier=1.0, viewmatrix=t(c.world_view_transform), projmatrix=t(c.full_proj_transform),
        sh_degree=D, campos=t(c.camera_center), prefiltered=False, score_flag=False) for c in my_cams]
    settings = settings_list[0]
    contexts = [ctx(False)] + [ctx(True) for _ in range(V - 1)]
    rasts = [GaussianRasterizer(raster_settings=s_, context=c_) for s_, c_ in zip(settings_list, contexts)]
    plain_rast0 = GaussianRasterizer(raster_settings=settings)        # no arena: autograd returns view 0's own gradients
    leaves = [params[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]

    from dreamscene_amd.graph import CapturedViews
    from dreamscene_amd.views import GaussianRasterizerViews
    views_ctx = ctx(False)
    rast_views = GaussianRasterizerViews(settings_list, context=views_ctx)
    rast_captured = CapturedViews(context=views_ctx)

    def set_profile(p):
        prof_holder[0] = p
        for c_ in contexts + [views_ctx]:
            c_.profile = p

    skip_reduce = [False]

    def reduce_grads():
        if skip_reduce[0]:
            return
        if exchange is not None:
            exchange.reduce()
        else:
            multiview.allreduce_grads(arena)

    def step_batched():
        means2D = torch.zeros((V,) + tuple(params["means3D"].shape), device=dev, requires_grad=True)
        if use_capture[0] and prof_holder[0] is None:      # (the stage timers record events: the profiled passes run eagerly)
            outs = rast_captured(settings_list, means3D=params["means3D"], means2D=means2D, opacities=params["opacities"],
                                 shs=params["shs"], scales=params["scales"], rotations=params["rotations"])
        else:
            outs = rast_views(means3D=params["means3D"], means2D=means2D, shs=params["shs"], colors_precomp=None,
                              opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"],
                              cov3D_precomp=None)
        (g2d,) = torch.autograd.grad([t_ for (img, _, da) in outs for t_ in (img, da)], [means2D], [gi, gda] * V)
        reduce_grads()
        last_batched[0] = (outs, g2d)
        return outs[0], g2d[0]

    last_batched = [None]          # (outputs of all views, means2D gradients [V,P,3]) of the latest batched step
    plain_rasts = [GaussianRasterizer(raster_settings=s_) for s_ in settings_list]
    # the same modules with the opt-in internal streams (RasterContext.side_streams / GSR_SIDE_STREAMS=2: the forward of a call
    # whose inputs are provably unchanged since an earlier call runs on an internal stream beside the previous views)
    stream_rasts = [GaussianRasterizer(raster_settings=s_, context=RasterContext(side_streams=2)) for s_ in settings_list]
    dropin_rasts = [plain_rasts]

    def step_dropin():
        """The reference's interface, used the way its trainers use it (training/object_trainer.py:302-382): the V views of a
        step are rendered one GaussianRasterizer call after the other, all outputs are kept, then the backward of every
        view runs (here one torch.autograd.grad per view, last view first -- the order autograd runs them in -- so that
        no torch-side gradient accumulation is timed with the rasterizer). No context, no arena: the module as imported.
        With several ranks the per-view gradients have to meet in the arena for the exchange: the arena form below."""
        if world > 1:
            return step_dropin_arena()
        held = []
        for rast in dropin_rasts[0]:
            means2D = torch.zeros_like(params["means3D"], requires_grad=True)
            img, radii, da = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"], colors_precomp=None,
                                  opacities=params["opacities"], scales=params["scales"],
                                  rotations=params["rotations"], cov3D_precomp=None)
            held.append((img, radii, da, means2D))
        g2d0 = None
        for img, radii, da, means2D in reversed(held):
            gr = torch.autograd.grad([img, da], leaves + [means2D], [gi, gda])
            g2d0 = gr[-1]
        return held[0][:3], g2d0

    def step_dropin_arena():
        out0 = g2d0 = None
        for j, rast in enumerate(rasts):
            means2D = torch.zeros_like(params["means3D"], requires_grad=True)
            img, radii, da = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"], colors_precomp=None,
                                  opacities=params["opacities"], scales=params["scales"],
                                  rotations=params["rotations"], cov3D_precomp=None)
            (g2d,) = torch.autograd.grad([img, da], [means2D], [gi, gda])
            if j == 0:
                out0, g2d0 = (img, radii, da), g2d
        reduce_grads()
        return out0, g2d0

    step = step_batched if batched else step_dropin

    def view0_with_own_gradients():
        """One untimed drop-in call of view 0 WITHOUT the arena: autograd returns that view's own gradients (parity check)."""
        means2D = torch.zeros_like(params["means3D"], requires_grad=True)
        img, radii, da = plain_rast0(means3D=params["means3D"], means2D=means2D, shs=params["shs"], colors_precomp=None,
                                     opacities=params["opacities"], scales=params["scales"],
                                     rotations=params["rotations"], cov3D_precomp=None)
        grads = torch.autograd.grad([img, da], leaves + [means2D], [gi, gda])
        return img, da, radii, list(grads)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    prof = None
    if not args.no_roofline:
        prof = _lib.Profile()
    for _ in range(args.warmup):
        step()
    sync()
    capture_probe = None
    if cap_mode == "auto":
        rates = {}
        for flag in (False, True):
            use_capture[0] = flag
            for _ in range(4):
                step()
            sync()
            tp = time.perf_counter()
            for _ in range(15):
                step()
            sync()
            rates[flag] = 15 * V / (time.perf_counter() - tp)
        if world > 1:      # every rank must take the same decision
            lo = allreduce_scalars([rates[False], rates[True]], dist.ReduceOp.MIN)
            rates = {False: lo[0], True: lo[1]}
        use_capture[0] = rates[True] > rates[False]
        capture_probe = {"eager_views_per_s": round(rates[False], 1), "captured_views_per_s": round(rates[True], 1)}
    captured = bool(use_capture[0])

    # ---- the exchange, measured then chosen (world > 1): `--exchange measure` times the step with each wire format and
    # keeps the fastest; `rccl` = the plain in-place all-reduce of the REAL arena (236 P bytes at K = 16) on its own
    exchange_probe, rccl = None, None
    if world > 1:
        rccl = measure_allreduce(arena, dev, backend, allreduce_scalars)
        if args.exchange == "measure":
            # gloo (functional tests on one device) stages device tensors through the host for dense / rows only
            # (rows: the device form of round 6 -- self-describing messages, no host read: multiview._RowMessages)
            # (sparse_rs: the device form as well -- equal-split all-to-all + all-gather of fixed-size messages)
            cands = ["dense", "direct", "rows", "sparse_rs"] if backend == "nccl" else ["dense", "rows"]
            if args.exchange_candidates:
                cands = [f for f in (x.strip() for x in args.exchange_candidates.split(",")) if f]
                unknown = [f for f in cands if f not in ("dense", "direct", "rows", "sparse_rs")]
                if unknown or not cands:
                    raise SystemExit(f"--exchange-candidates: unknown format(s) {unknown}")
            exchange_probe = {}
            for fmt in cands:
                ok = 1.0
                try:
                    exchange.mode = fmt
                    exchange.strict = True        # (warm-up: a row message that does not fit is repeated with more room)
                    for _ in range(3):
                        step()
                    sync()
                    exchange.strict = False       # timed: no host read at all; an overflow would show in overflowed_steps
                    over0 = exchange.overflowed_steps
                    tp = time.perf_counter()
                    for _ in range(args.exchange_probe_steps):
                        step()
                    sync()
                    dt_f = (time.perf_counter() - tp) / args.exchange_probe_steps
                    exchange.finish()
                    if exchange.overflowed_steps != over0:
                        raise RuntimeError(f"{exchange.overflowed_steps - over0} timed steps overflowed their row messages")
                except Exception as e:          # a format that fails on this stack is dropped on EVERY rank
                    ok, dt_f = 0.0, float("inf")
                    print(f"bench.py rank {rank}: exchange format {fmt} failed: {e!r}", file=sys.stderr, flush=True)
                ok = allreduce_scalars([ok], dist.ReduceOp.MIN)[0]
                dt_f = allreduce_scalars([dt_f if ok else 0.0], dist.ReduceOp.MAX)[0]
                exchange_probe[fmt] = {"ms_per_step": round(dt_f * 1e3, 4), "last": dict(exchange.last)} if ok else \
                    {"failed": True}
            good = {f: v["ms_per_step"] for f, v in exchange_probe.items() if "ms_per_step" in v}
            exchange.mode = min(good, key=good.get) if good else "dense"
            exchange.strict = True
            for _ in range(2):
                step()
            sync()
            exchange.strict = False               # (the timed region: checked behind it, `exchange.overflowed_steps` on the line)
    stage_ms = {}
    dominant = None
    if prof is not None:
        # stage pass (untimed, after the warmup so that first-launch costs stay out of it): every stage timer on
        set_profile(prof)
        n_stage = 8
        for _ in range(n_stage):
            step()
        sync()
        res = prof.collect()
        n_views_prof = n_stage * V                    # per view (a stage may be one launch per view or per batch)
        stage_ms = {s: ms / n_views_prof for s, (ms, c) in res.items()}
        # the roofline entry is for the dominant single KERNEL: "sort" and "scan" are groups of small launches
        # (18 and 2 per view) and are reported in stage_us only
        single = {k: v for k, v in stage_ms.items() if k not in ("sort", "scan")}
        dominant = max(single, key=single.get)
        prof.reset()
        set_profile(None)
        for _ in range(3):               # (back to the timed configuration: no event records, captured graphs replay)
            step()

    sync()
    host_stats.wait_s = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    host_wait_s = host_stats.wait_s                  # of which: blocked on the pair counts of the projection
    host_enqueue_s = time.perf_counter() - t0        # the host is done enqueueing; the GPU may still be working
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = allreduce_scalars([elapsed], dist.ReduceOp.MAX)[0]

    # ---- a longer window of the same step (the --steps region above is ~20 ms at the driver's K = 20)
    sustained = None
    if args.sustain_seconds > 0:
        n_s = 0
        ts = time.perf_counter()
        while True:
            for _ in range(25):
                step()
            n_s += 25
            if world > 1:
                go = allreduce_scalars([time.perf_counter() - ts], dist.ReduceOp.MIN)[0] < args.sustain_seconds
            else:
                go = time.perf_counter() - ts < args.sustain_seconds
            if not go:
                break
        sync()
        dt_s = time.perf_counter() - ts
        if world > 1:
            dt_s = allreduce_scalars([dt_s], dist.ReduceOp.MAX)[0]
        sustained = {"views_per_s": round(world * n_s * V / dt_s, 3), "steps": n_s, "seconds": round(dt_s, 3)}

    # ---- changing cameras + an optimizer step between the steps: 64 cameras sampled like the reference's random cameras
    # (radius 5.2-5.5, polar 60-90 deg, FoV 0.32-0.60: config.py:88-99), 4 new ones per step, fused Adam on the summed
    # gradients in between. Exercises what frozen parameters and 4 fixed cameras cannot: the pair-capacity speculation, the
    # forward-variant choice, camera re-packing and graph re-capture.
    rotating = None
    if args.rotate_seconds > 0 and batched and args.scene == "object" and world == 1:
        from dreamscene_amd.optim import FusedAdam
        rng = np.random.default_rng(7)
        cams_r = [synth.orbit_camera(float(rng.uniform(5.2, 5.5)), float(rng.uniform(60.0, 90.0)), 360.0 * i / 64.0,
                                     float(rng.uniform(0.32, 0.60)), H, W) for i in range(64)]
        sl_r = [GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=t([1.0, 1.0, 1.0]), scale_modifier=1.0,
            viewmatrix=t(c.world_view_transform), projmatrix=t(c.full_proj_transform), sh_degree=D,
            campos=t(c.camera_center), prefiltered=False, score_flag=False) for c in cams_r]
        names = ("means3D", "scales", "rotations", "opacities", "shs")
        saved = {n_: params[n_].detach().clone() for n_ in names}
        opt = FusedAdam([params[n_] for n_ in names], lr=2e-5, eps=1e-15)
        arena_grads = [arena.views[n_].view(params[n_].shape) for n_ in names]
        # (both paths are timed here, whatever the headline step chose; at C3 both turn out GPU-bound -- 1 040 us of kernels per
        #  step with the wider random cameras and 128 us of Adam, tools/rotating_probe.py -- smaller workloads are paced by the
        #  host's ~45 launches per step on the eager path)
        rot_paths = [None] + ([CapturedViews(context=views_ctx)] if (batched and cap_mode != "off") else [])
        rot_captured = None

        def step_rot(i):
            sl = [sl_r[(V * i + j) % 64] for j in range(V)]
            means2D = torch.zeros((V,) + tuple(params["means3D"].shape), device=dev, requires_grad=True)
            if rot_captured is not None:
                outs = rot_captured(sl, means3D=params["means3D"], means2D=means2D, opacities=params["opacities"],
                                    shs=params["shs"], scales=params["scales"], rotations=params["rotations"])
            else:
                outs = GaussianRasterizerViews(sl, context=views_ctx)(
                    means3D=params["means3D"], means2D=means2D, shs=params["shs"], colors_precomp=None,
                    opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"], cov3D_precomp=None)
            torch.autograd.grad([t_ for (img, _, da) in outs for t_ in (img, da)], [means2D], [gi, gda] * V)
            opt.step(grads=arena_grads)

        rot_rates = {}
        for path_ in rot_paths:
            rot_captured = path_
            # (every path starts from the benchmark's parameters with a fresh optimizer: the synthetic upstream gradients make the
            #  scene drift -- more pairs per view, step after step: 1.05 -> 1.14 ms over 900 steps, tools/rotating_probe.py -- and
            #  the path that ran second was being timed on a heavier scene)
            with torch.no_grad():
                for n_ in names:
                    params[n_].copy_(saved[n_])
            opt = FusedAdam([params[n_] for n_ in names], lr=2e-5, eps=1e-15)
            for i in range(6):
                step_rot(i)
            sync()
            n_r = 0
            tr = time.perf_counter()
            while time.perf_counter() - tr < args.rotate_seconds / len(rot_paths):
                for _ in range(16):
                    step_rot(6 + n_r)
                    n_r += 1
            sync()
            dt_r = time.perf_counter() - tr
            name_ = "graph.CapturedViews" if path_ is not None else "views.GaussianRasterizerViews"
            rot_rates[name_] = (n_r * V / dt_r, n_r, dt_r, dict(path_.stats) if path_ is not None else None)
        best_ = max(rot_rates, key=lambda k_: rot_rates[k_][0])
        rotating = {"views_per_s": round(rot_rates[best_][0], 3), "steps": rot_rates[best_][1], "seconds": round(rot_rates[best_][2], 3),
                    "cameras": 64, "views_per_step": V,
                    "optimizer": "dreamscene_amd.optim.FusedAdam (one launch over the five parameter "
                    "groups, gradients read from the arena), lr 2e-5",
                    "through": best_, "by_path_views_per_s": {k_: round(v_[0], 1) for k_, v_ in rot_rates.items()},
                    "capture_stats": rot_rates.get("graph.CapturedViews", (None,) * 4)[3]}
        with torch.no_grad():                      # back to the benchmark's parameters for what follows
            for n_ in names:
                params[n_].copy_(saved[n_])
        del saved, opt

    # ---- what training really feeds the rasterizer: object_render(test=False) adds fresh noise to
    # the activated scales of EVERY view (scene_gaussian.py:1004-1008, scale_aug_ratio = 1.0: config.py:23) => scales
    # [V,P,3], cov3D per view in K1, one scale gradient per view in K8; act_SH = 0 with probability sh_deg_aug_ratio = 0.1
    # (:938-941); background random or black with probability bg_aug_ratio = 0.5 (:943-947). Cameras: the 64 random ones.
    # The noise is drawn inside the timed loop with torch, as the trainers do. Then the same plain step in the reference's
    # INITIAL state (every opacity 0.1: gs_renderer.py:598 -- ~87 layers blend before a pixel stops), and forward-only
    # rendering (video_inference, object_trainer.py:81-118: test=True views under no_grad).
    training_like = init_state = forward_only = trainer_step = None
    if args.train_seconds > 0 and args.scene == "object" and world == 1:
        import random as _random
        rng_t = np.random.default_rng(11)
        pyrng = _random.Random(11)
        cams_t = [synth.orbit_camera(float(rng_t.uniform(5.2, 5.5)), float(rng_t.uniform(60.0, 90.0)), 360.0 * i / 64.0,
                                     float(rng_t.uniform(0.32, 0.60)), H, W) for i in range(64)]
        white, black = t([1.0, 1.0, 1.0]), t([0.0, 0.0, 0.0])
        cam_t = [(c, t(c.world_view_transform), t(c.full_proj_transform), t(c.camera_center)) for c in cams_t]

        def train_settings(i):
            out_ = []
            for j in range(V):
                c, vm_, pm_, cp_ = cam_t[(V * i + j) % 64]
                bg_ = white
                if pyrng.random() < 0.66:           # bg_aug_ratio of the shipped config (configs/objects/sample.yaml:74)
                    bg_ = torch.rand(3, device=dev) if pyrng.random() < 0.5 else black
                out_.append(GaussianRasterizationSettings(
                    image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=bg_, scale_modifier=1.0,
                    viewmatrix=vm_, projmatrix=pm_, sh_degree=(0 if pyrng.random() < 0.1 else D), campos=cp_,
                    prefiltered=False, score_flag=False))
            return out_

        def noisy_scales(shape):
            sc = params["scales"].detach()
            return torch.clamp(sc + torch.randn(shape, device=dev) * ((0.2 ** 0.5) * sc / 4), 0.0).requires_grad_(True)

        tl_ctx = ctx(False)
        tl_paths = [None] + ([CapturedViews(context=tl_ctx)] if (batched and cap_mode != "off") else [])     # (both timed, as above)
        tl_captured = None

        def step_train_batched(i):
            sl = train_settings(i)
            means2D = torch.zeros((V,) + tuple(params["means3D"].shape), device=dev, requires_grad=True)
            sc = noisy_scales((V,) + tuple(params["scales"].shape))
            if tl_captured is not None:
                outs = tl_captured(sl, means3D=params["means3D"], means2D=means2D, opacities=params["opacities"],
                                   shs=params["shs"], scales=sc, rotations=params["rotations"])
            else:
                outs = GaussianRasterizerViews(sl, context=tl_ctx)(
                    means3D=params["means3D"], means2D=means2D, shs=params["shs"], colors_precomp=None,
                    opacities=params["opacities"], scales=sc, rotations=params["rotations"], cov3D_precomp=None)
            torch.autograd.grad([t_ for (img, _, da) in outs for t_ in (img, da)], [means2D, sc], [gi, gda] * V)

        def step_train_dropin(i):
            sl = train_settings(i)
            held = []
            for j in range(V):
                means2D = torch.zeros_like(params["means3D"], requires_grad=True)
                sc = noisy_scales(tuple(params["scales"].shape))
                # (no arena here: exactly the unmodified trainers' call -- autograd receives every parameter gradient)
                img, radii, da = GaussianRasterizer(raster_settings=sl[j])(
                    means3D=params["means3D"], means2D=means2D, shs=params["shs"], colors_precomp=None,
                    opacities=params["opacities"], scales=sc, rotations=params["rotations"], cov3D_precomp=None)
                held.append((img, da, means2D, sc))
            for img, da, means2D, sc in reversed(held):      # all views forward, then their backwards: the trainers' order
                torch.autograd.grad([img, da], [params["means3D"], params["shs"], params["opacities"], params["rotations"],
                                                means2D, sc], [gi, gda])

        def timed(fn, seconds, chunk=8, indexed=True):
            for i in range(6):
                fn(i) if indexed else fn()
            sync()
            n_, t_ = 0, time.perf_counter()
            while time.perf_counter() - t_ < seconds:
                for _ in range(chunk):
                    fn(6 + n_) if indexed else fn()
                    n_ += 1
            sync()
            return n_, time.perf_counter() - t_

        set_profile(None)
        tl = {}
        if batched:
            tl_rates = {}
            for path_ in tl_paths:
                tl_captured = path_
                n_, dt_ = timed(step_train_batched, args.train_seconds / len(tl_paths))
                tl_rates["graph.CapturedViews" if path_ is not None else "views.GaussianRasterizerViews"] = (n_ * V / dt_, n_, dt_, path_)
            best_ = max(tl_rates, key=lambda k_: tl_rates[k_][0])
            tl_captured = tl_rates[best_][3]                  # (the initial-state leg below uses the faster one)
            tl.update(views_per_s=round(tl_rates[best_][0], 3), steps=tl_rates[best_][1], seconds=round(tl_rates[best_][2], 3),
                      through=best_, by_path_views_per_s={k_: round(v_[0], 1) for k_, v_ in tl_rates.items()})
        n_, dt_ = timed(step_train_dropin, args.train_seconds)
        tl.update(dropin_views_per_s=round(n_ * V / dt_, 3), dropin_steps=n_,
                  what="per-view scale noise (scales [V,P,3], drawn with torch inside the loop), SH degree 0 w.p. 0.1, "
                       "random / black background w.p. 0.66, 64 random cameras, 4 new ones per step; batched: gradients to the "
                       "noisy scales and means2D through autograd, the other parameter gradients summed in the arena; "
                       "drop-in: the unmodified trainers' call, every gradient through autograd")
        training_like = tl
        # the reference's initial state: every opacity 0.1
        held_op = params["opacities"].detach().clone()
        with torch.no_grad():
            params["opacities"].fill_(0.1)
        n_, dt_ = timed(step, args.train_seconds, chunk=4, indexed=False)
        init_state = {"views_per_s": round(n_ * V / dt_, 3), "steps": n_, "seconds": round(dt_, 3),
                      "what": "the headline step with every opacity at 0.1 (gs_renderer.py:598)"}
        if batched:
            n_, dt_ = timed(step_train_batched, args.train_seconds, chunk=4)
            init_state["training_like_views_per_s"] = round(n_ * V / dt_, 3)
        with torch.no_grad():
            params["opacities"].copy_(held_op)
        del held_op
        for _ in range(3):
            step()
        # A whole reference-shaped training step (training/object_trainer.py:293-400: activations on fresh tensors, the four
        # object_render calls INCLUDING the disp glue and its boolean-mask host read, stand-in guidance loss + TV + scale loss,
        # backward, densification statistics, Adam): tools/train_step.py, three legs, at C3 and in the opacity-0.1 state.
        # `rasterizer_ms` = what the rasterizer alone takes for the step's views on the same path (the entries above).
        try:
            from tools import train_step as TS
            trainer_step = {"c3": TS.measure(P, H, W, V, K, D, dev, init_opacity=False, seconds=args.train_seconds),
                            "init_state": TS.measure(P, H, W, V, K, D, dev, init_opacity=True, seconds=args.train_seconds),
                            "what": "one object-training step shaped like training/object_trainer.py:293-400 "
                                    "(tools/train_step.py): as_imported = the package as a DreamScene checkout imports it "
                                    "(per-view calls, disp glue with its host read, torch Adam); views_fused = "
                                    "GaussianRasterizerViews + statistics in K8 + FusedAdam; raw_leaves = "
                                    "scene.rasterize_models_views (activations and noise fused into K1 / K8)"}
            ras = {"as_imported": tl.get("dropin_views_per_s"), "views_fused": tl.get("views_per_s"),
                   "raw_leaves": tl.get("views_per_s")}
            for leg_, vps_ in ras.items():
                e_ = trainer_step["c3"].get(leg_, {})
                if vps_ and "ms_per_step" in e_:
                    e_["rasterizer_ms"] = round(1e3 * V / vps_, 3)
                    e_["rasterizer_share"] = round(e_["rasterizer_ms"] / e_["ms_per_step"], 3)
        except Exception as e_:           # (the harness must never take the bench line with it)
            trainer_step = {"error": repr(e_)[:300]}
        torch.cuda.empty_cache()
        # forward only (video_inference): test=True views, no_grad
        fo_rast = [GaussianRasterizer(raster_settings=s_) for s_ in settings_list]
        fo_views = GaussianRasterizerViews(settings_list)

        def fwd_dropin():
            with torch.no_grad():
                for r_ in fo_rast:
                    r_(means3D=params["means3D"], means2D=None, shs=params["shs"], colors_precomp=None,
                       opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"],
                       cov3D_precomp=None)

        def fwd_batched():
            with torch.no_grad():
                fo_views(means3D=params["means3D"], means2D=torch.empty((V, 0)), shs=params["shs"], colors_precomp=None,
                         opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"],
                         cov3D_precomp=None)

        n_, dt_ = timed(fwd_dropin, args.train_seconds / 2, indexed=False)
        forward_only = {"dropin_views_per_s": round(n_ * V / dt_, 3)}
        if batched:
            n_, dt_ = timed(fwd_batched, args.train_seconds / 2, indexed=False)
            forward_only["batched_views_per_s"] = round(n_ * V / dt_, 3)
        forward_only["what"] = "forward only under no_grad (video_inference, object_trainer.py:81-118): one GaussianRasterizer " \
                               "call per view / the same views through one GaussianRasterizerViews call"
        # importance scoring (prune_list, scene_gaussian.py:1063-1079: 48 sphere cameras, one score_render each, scores summed):
        # the reference's per-camera loop through the drop-in module against views.importance_scores (cameras 16 at a time
        # through the batched forward, every view's per-splat pixel counts added to ONE buffer by the kernel's integer atomics)
        from dreamscene_amd import views as VW_
        sph = synth.sphere_cameras(48, H, W)
        sl_sc = [GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=white, scale_modifier=1.0,
            viewmatrix=t(c.world_view_transform), projmatrix=t(c.full_proj_transform), sh_degree=D, campos=t(c.camera_center),
            prefiltered=False, score_flag=True) for c in sph]
        sc_rasts = [GaussianRasterizer(raster_settings=s_) for s_ in sl_sc]
        m2d_sc = torch.zeros_like(params["means3D"])

        def score_loop():
            imp = None
            with torch.no_grad():
                for r_ in sc_rasts:
                    sc_ = r_(means3D=params["means3D"], means2D=m2d_sc, shs=params["shs"], colors_precomp=None,
                             opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"],
                             cov3D_precomp=None)[0]
                    imp = sc_ if imp is None else imp.add_(sc_)
            return imp

        def score_batched():
            return VW_.importance_scores(sl_sc, params["means3D"].detach(), params["opacities"].detach(),
                                         shs=params["shs"].detach(), scales=params["scales"].detach(),
                                         rotations=params["rotations"].detach())

        def reps(fn, n_rep=3):
            fn(); fn()
            sync()
            t_ = time.perf_counter()
            for _ in range(n_rep):
                r_ = fn()
            sync()
            return (time.perf_counter() - t_) / n_rep, r_
        dt_loop, imp_a = reps(score_loop)
        dt_b, imp_b = reps(score_batched)
        den = float(imp_a.abs().max().clamp_min(1.0))
        forward_only["score_views"] = {
            "cameras": 48, "per_camera_loop_views_per_s": round(48 / dt_loop, 1), "batched_views_per_s": round(48 / dt_b, 1),
            "speedup": round(dt_loop / dt_b, 2), "max_abs_diff_over_max": float((imp_a - imp_b).abs().max()) / den,
            "what": "the 48-camera importance-score sum of prune_list (scene_gaussian.py:1063-1079), score weight = opacity per "
                    "contributing (pixel, splat): one score_flag GaussianRasterizer call per camera + imp_list += score / "
                    "views.importance_scores (16 cameras per batched forward, integer pixel counts in one buffer)"}
        del sc_rasts, sl_sc, imp_a, imp_b

    # the drop-in figure: the same views through one GaussianRasterizer call per view (untimed w.r.t. `value`)
    dropin = None
    if batched and not args.no_dropin:
        set_profile(None)
        for _ in range(3):
            step_dropin()
        sync()
        n_drop = 0
        td = time.perf_counter()
        while True:                   # >= max(10 steps, --sustain-seconds), like `sustained`
            for _ in range(10):
                step_dropin()
            n_drop += 10
            go = time.perf_counter() - td < args.sustain_seconds
            if world > 1:
                go = allreduce_scalars([1.0 if go else 0.0], dist.ReduceOp.MIN)[0] > 0.5
            if not go:
                break
        sync()
        dt_d = time.perf_counter() - td
        if world > 1:
            dt_d = allreduce_scalars([dt_d], dist.ReduceOp.MAX)[0]
        dropin = {"views_per_s": world * n_drop * V / dt_d, "steps": n_drop, "seconds": round(dt_d, 3)}
        if world == 1:
            dropin_rasts[0] = stream_rasts
            for _ in range(3):
                step_dropin()
            sync()
            n_ds, tds = 0, time.perf_counter()
            while time.perf_counter() - tds < max(0.5, args.sustain_seconds / 2):
                for _ in range(10):
                    step_dropin()
                n_ds += 10
            sync()
            dropin["internal_streams"] = {
                "views_per_s": round(n_ds * V / (time.perf_counter() - tds), 3), "streams": 2, "steps": n_ds,
                "what": "the same calls with RasterContext(side_streams=2) / GSR_SIDE_STREAMS=2 (opt-in): a call whose inputs are "
                        "the same live tensors at the same autograd versions as at an earlier call forks from that call's event "
                        "and its FORWARD runs on an internal stream beside the previous views' (backward on the caller's stream)",
                "stats": R.side_stream_stats()}
            dropin_rasts[0] = plain_rasts

    # what the exchange would have to move for this rank's step (the arena holds the sum over the step's V views)
    exch = None
    if exchange is not None:
        skip_reduce[0] = True          # this rank's own step, before any exchange: the rows ITS views reached
        step() if batched else step_dropin_arena()      # (the plain per-view module does not write the arena)
        skip_reduce[0] = False
        torch.cuda.synchronize(dev)
        nz = int(exchange.nonzero_rows().numel())
        exchange.finish()
        exch = {"row_floats": exchange.row_floats, "dense_bytes": int(4 * exchange.row_floats * P),
                "overflowed_steps": exchange.overflowed_steps,
                "nonzero_row_frac": round(nz / max(P, 1), 4), "format": exchange.mode, "last": exchange.last or None,
                "probe_ms_per_step": exchange_probe}

    N_pairs = None
    roofline = None
    if prof is not None:
        # launch duration of the dominant kernel: HIP events around its launches on the launch stream. The timed region
        # replays captured graphs, which cannot carry event records, so the same launches are issued eagerly here (same
        # kernels, same arguments) right after it, only that stage's timer on.
        prof.set_stages([dominant])
        set_profile(prof)
        n_roof = max(10, min(50, args.steps // 4))
        for _ in range(n_roof):
            step()
        sync()
        res = prof.collect()
        ms, cnt = res[dominant]
        set_profile(None)
        # pair count of this rank's view (for the algorithmic byte count)
        with torch.no_grad():
            o, _ = R.rasterize_forward_raw(settings, params["means3D"], params["opacities"], params["shs"], None,
                                           params["scales"], params["rotations"], None)
        N_pairs = int(o["N"])
        if cnt:
            avg_s = ms / cnt * 1e-3                   # the stage timer brackets exactly one launch of the kernel
            per_launch = V if batched else 1     # batched: one launch covers the step's V views
            ab = algorithmic_bytes(dominant, P, N_pairs, H * W, K, D, views=per_launch)
            achieved = ab / avg_s / 1e9
            # HBM traffic and VALU instructions per launch of that kernel: rocprofv3 --pmc passes of this very command,
            # digested by tools/profile_digest.py into profiles/traffic.json (per launch, each pass normalised by its own
            # dispatch count); only used when they were collected for this configuration and call pattern
            traffic, valu, traffic_kernel = None, None, None
            tf = os.path.join(ROOT, "profiles", "traffic.json")
            key = f"{args.scene}{'-init' if args.init_opacity else ''}_{P}_{W}" + ("" if batched else "_dropin")
            if os.path.exists(tf):
                try:
                    ent = json.load(open(tf)).get(key, {})
                    if ent.get("views_per_step") == V and bool(ent.get("batched_call")) == batched:
                        # the EXACT template instance the timed launch runs (k_render_bwd<256> and <128> both appear in a
                        # profile of this command: the eager warm-up calls use other item lengths)
                        seg_t = R.pick_seg_len(int(N_pairs * 1.5) if N_pairs else None, per_launch)
                        want = timed_kernel_name(dominant, K, seg_t, batched)
                        names_b = list(ent.get("per_launch_bytes", {}))
                        kn_hit = pick_kernel(names_b, want, STAGE_KERNEL.get(dominant), dominant)
                        if kn_hit is not None:
                            traffic = ent["per_launch_bytes"][kn_hit]
                            traffic_kernel = kn_hit
                        kn_sq = pick_kernel(list(ent.get("sq_per_launch", {})), want, STAGE_KERNEL.get(dominant), dominant)
                        for kn, sq in ent.get("sq_per_launch", {}).items():
                            if kn == kn_sq and "SQ_INSTS_VALU" in sq:
                                n_valu = sq["SQ_INSTS_VALU"]
                                n_salu = sq.get("SQ_INSTS_SALU", 0.0)
                                mix = VALU_MIX.get(dominant, {"plain": 0.5, "other": 0.5, "trans": 0.0})
                                ns_valu = sum(mix[c] * ISSUE_NS[c] for c in mix)
                                # (scalar instructions issue beside the vector instructions of the other waves: removing 22 of
                                #  them per step from K6's loop changed nothing, tools/ab_lib.sh A/B -- they are not priced)
                                busy_ns = n_valu * ns_valu
                                valu = {"insts_per_launch": int(n_valu), "salu_per_launch": int(n_salu),
                                        "issue_ns": ISSUE_NS, "valu_mix": mix, "simds": SIMDS,
                                        # modelled share of the SIMDs' time the launch's VECTOR instructions need to issue
                                        "issue_frac": round(busy_ns * 1e-9 / (SIMDS * avg_s), 4),
                                        # VALU instructions per second against the best the part issues (plain class)
                                        "inst_rate_frac_of_peak": round(n_valu * ISSUE_NS["plain"] * 1e-9 / (SIMDS * avg_s), 4),
                                        "source": f"profiles/{ent.get('tag', '?')}_pmc.txt (SQ_INSTS_VALU / SQ_INSTS_SALU per "
                                                  "launch) x the measured issue costs (profiles/r03_valu_rate.txt) over the launch "
                                                  "duration measured live here; packed fp32 is no faster than scalar fp32 on "
                                                  "gfx950, so the scalar issue rate IS the fp32 peak"}
                except Exception:
                    traffic, valu = None, None
            # what bounds the kernel: the compositing kernels issue VALU instructions on > 80 % of the cycles and move far
            # fewer bytes than their algorithmic count (early termination, L2-resident splat table); the streaming kernels
            # are HBM-bound. `achieved` / `peak` / `frac` are the HBM-roofline numbers the contract asks for in both cases.
            bound = "valu" if dominant in ("render_fwd", "render_bwd") else "hbm"
            # The whole path, three ways (per step of V views per GPU; `elapsed` is the --steps region):
            #  * work_equivalent: SURVEY.md section 8(d)(i)'s per-VIEW contract bytes (848 P + 124 N + 56 HW at K=16, D=3) x views/s
            #    -- what V separate per-view calls would have to move. NOT a bandwidth: the batched K1 / K8 read the parameter
            #    rows once per launch, so the step moves fewer bytes than this (a figure above what a copy achieves, 6.3 TB/s,
            #    is possible and means nothing about the memory system);
            #  * launch_accurate: the sum over the step's stages of the algorithmic bytes of the launches that really run
            #    (algorithmic_bytes(stage, views=V)) / the step time = the HBM-roofline fraction of the step;
            #  * counters: measured HBM bytes per step (rocprofv3 --pmc, profiles/traffic.json) / the step time.
            S_ = 12 * (D + 1) ** 2
            e2e_bytes = P * (44 + S_) + P * 48 + N_pairs * 12 + N_pairs * 24 + N_pairs * 44 + H * W * 28 + \
                N_pairs * 44 + H * W * 28 + P * 40 + P * (44 + S_ + 40) + P * (44 + 12 * K + 12)
            step_s = elapsed / args.steps
            stages_all = ("preprocess", "scan", "duplicate", "sort", "ranges", "render_fwd", "render_bwd", "preprocess_bwd")
            per_stage = {st_: algorithmic_bytes(st_, P, N_pairs, H * W, K, D, views=per_launch) *
                         (1 if batched else V) for st_ in stages_all}
            la_bytes = sum(per_stage.values())
            counter_bytes = None
            try:
                ent_ = json.load(open(tf)).get(key, {}) if os.path.exists(tf) else {}
                if ent_.get("views_per_step") == V and bool(ent_.get("batched_call")) == batched and ent_.get("per_step_bytes"):
                    counter_bytes = int(sum(ent_["per_step_bytes"].values()))
            except Exception:
                counter_bytes = None
            whole = {"launch_accurate_bytes_per_step": int(la_bytes),
                     "launch_accurate_GBps": round(la_bytes / step_s / 1e9, 1),
                     "launch_accurate_frac_of_hbm_peak": round(la_bytes / step_s / 1e9 / HBM_PEAK_GBS, 4),
                     "launch_accurate_bytes_by_stage": {k_: int(v_) for k_, v_ in per_stage.items()},
                     "counter_bytes_per_step": counter_bytes,
                     "counter_frac_of_hbm_peak": (round(counter_bytes / step_s / 1e9 / HBM_PEAK_GBS, 4)
                                                  if counter_bytes else None),
                     "work_equivalent": {"contract_bytes_per_view": int(e2e_bytes),
                                         "GBps": round(e2e_bytes * V / step_s / 1e9, 1),
                                         "frac_of_hbm_peak": round(e2e_bytes * V / step_s / 1e9 / HBM_PEAK_GBS, 4),
                                         "note": "per-view contract bytes x views/s: work done per second in units of the "
                                                 "per-view byte contract, not bytes moved (the batched launches read the "
                                                 "parameter rows once for all views)"}}
            roofline = {"bound": bound, "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "traffic_kernel": traffic_kernel, "valu": valu,
                        "avg_launch_us": round(avg_s * 1e6, 2), "views_per_launch": per_launch,
                        "launches_timed": int(cnt),
                        "timed_how": "HIP events around the kernel's launches on the launch stream, eager pass right after "
                                     "the timed region" + (" (the timed region replays captured graphs)" if captured else ""),
                        "whole_path": whole,
                        "algorithmic_bytes": int(ab),
                        "stage_us_per_view": {s: round(v * 1e3, 2) for s, v in stage_ms.items()}}

    cpu_baseline = None
    grad_err = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # parity ON THE TIMED PATH: one more step exactly as timed (the batched call: captured graphs or the views module,
        # K1 / K8 of all views in one launch, K6 / K7 <seg> of the batch, gradients summed in the arena) -- every view's image
        # and means2D gradient and the arena's SUM over the step's V views are compared with the scalar C oracle's per-view
        # results and their float64 sum (`max_grad_err_vs_oracle.batched_sum`)
        timed_path = None
        if batched:
            set_profile(None)
            step()
            torch.cuda.synchronize(dev)
            outs_b, g2d_b = last_batched[0]
            timed_path = {"through": "graph.CapturedViews" if captured else "views.GaussianRasterizerViews",
                          "cams": my_cams,
                          "images": [o_[0].detach().cpu().numpy() for o_ in outs_b],
                          "depth_alphas": [o_[2].detach().cpu().numpy() for o_ in outs_b],
                          "radii": [o_[1].detach().cpu().numpy() for o_ in outs_b],
                          "means2D": g2d_b.detach().cpu().numpy(),
                          "arena": {n_: arena.views[n_].detach().cpu().numpy().copy() for n_ in
                                    ("means3D", "shs", "opacities", "scales", "rotations")}}
        out = view0_with_own_gradients()  # (and the plain per-view module, view 0's own gradients: the drop-in path)
        torch.cuda.synchronize(dev)
        cpu_baseline, grad_err = cpu_baseline_leg(g, cam, D, K, H, W, gi_np, gda_np, out, extra_cams=cams[1:],
                                                  init_opacity=args.init_opacity, timed_path=timed_path)

    if rank == 0:
        views = world * args.steps * V
        line = {
            "metric": f"fwd+bwd views/s @{W}x{H}, {P} Gaussians",
            "value": round(views / elapsed, 3),
            "unit": "views/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "dropin_views_per_s": round(dropin["views_per_s"], 3) if dropin else
            (None if batched else round(views / elapsed, 3)),
            "dropin_internal_streams": dropin.get("internal_streams") if dropin else None,
            "sustained_views_per_s": sustained["views_per_s"] if sustained else None,
            "sustained": sustained,
            "rotating_cameras": rotating,
            "dropin_graphs": dropin_ring_stats(),
            "training_like": training_like,
            "trainer_step": trainer_step,
            "init_views_per_s": init_state["views_per_s"] if init_state else None,
            "init_state": init_state,
            "forward_only": forward_only,
            "host_enqueue_ms_per_step": round(host_enqueue_s / args.steps * 1e3, 4),
            "host_wait_ms_per_step": round(host_wait_s / args.steps * 1e3, 4),
            "host_busy_ms_per_step": round((host_enqueue_s - host_wait_s) / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "gaussians": P, "resolution": [H, W], "tile_pairs_N": N_pairs,
                       "views_per_step_per_gpu": V, "batched_call": batched, "captured_graphs": captured,
                       "batched_through": ("graph.CapturedViews (launches replayed from captured hipGraphs)" if captured else
                                           "views.GaussianRasterizerViews (launches issued from Python)") if batched else None,
                       "seg_len": {"batched": R.pick_seg_len(int(N_pairs * 1.5) if N_pairs else None, V if batched else 1),
                                   "one_view_per_call": R.pick_seg_len(int(N_pairs * 1.5) if N_pairs else None, 1),
                                   "policy": "entries per forward checkpoint / backward work item: 128 when pair capacity x "
                                             "views of the launch < 8 M, else 256 (rasterizer.pick_seg_len)"},
                       "capture_mode": cap_mode, "capture_probe": capture_probe,
                       "capture_stats": dict(rast_captured.stats) if captured else None,
                       "dropin": ("`dropin_views_per_s`: the same views through one GaussianRasterizer call per view (the "
                                  f"reference's interface), {dropin['steps']} steps / {dropin['seconds']} s after the timed region") if dropin else
                                 ("not measured (--no-dropin)" if batched else
                                  "`value` IS the drop-in figure (one GaussianRasterizer call per view)"),
                       "parallelism": f"{V} view(s)/GPU/step x {world} GPUs, gradients summed on the device, then 1 RCCL "
                                      f"gradient exchange per step (multiview.GradExchange, format {exchange.mode}: "
                                      f"{exchange.last})" if world > 1 else
                                      f"single GPU, {V} view(s) per step, gradients summed on the device"},
            "roofline": roofline,
            "exchange": exch,
            "rccl": rccl,
            "gpus_requested": args.gpus,
            "cpu_baseline": cpu_baseline,
            "max_grad_err_vs_oracle": grad_err,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def dropin_ring_stats():
    from dreamscene_amd import dropin
    return {"enabled": dropin.ENABLED, "rings": dropin.stats()}


def measure_allreduce(arena, dev, backend, allreduce_scalars, reps: int = 20):
    """The in-place sum of the real gradient arena over all ranks on its own (no rendering): ms per all-reduce (max over
    ranks) and the bus bandwidth 2 (W-1)/W bytes / t the ring formula defines. RCCL over xGMI under the nccl backend; under
    gloo (functional tests, ranks sharing one device) the tensor goes through the host and the figure means nothing."""
    from dreamscene_amd import multiview
    W = dist.get_world_size()
    for _ in range(3):
        multiview.allreduce_grads(arena)
    torch.cuda.synchronize(dev)
    dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        multiview.allreduce_grads(arena)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / reps
    dt = allreduce_scalars([dt], dist.ReduceOp.MAX)[0]
    nbytes = arena.nbytes()
    return {"ranks": W, "backend": ("rccl (torch.distributed nccl)" if backend == "nccl" else backend),
            "arena_bytes": int(nbytes), "allreduce_ms": round(dt * 1e3, 4),
            "bus_GBps": round(2 * (W - 1) / W * nbytes / dt / 1e9, 2),
            "alg_GBps": round(nbytes / dt / 1e9, 2), "reps": reps}


def _torch_cpu_leg(P, res, n_views, budget_s):
    """The PyTorch-CPU oracle (vectorised forward, autograd backward -- BASELINE.md section 3's stand-in for the
    reference's non-existent CPU path) on all host cores: fwd+bwd views/s at (P Gaussians, res x res)."""
    from oracle import torch_oracle as TO
    from dreamscene_amd import synth
    g = synth.g_object(P, seed=0, K=16)
    cams = synth.object_cameras(max(n_views, 1), res, res)
    gi, gda = (torch.tensor(x) for x in synth.upstream_grads(res, res, 0))
    done, t_tot = 0, 0.0
    for cam in cams[:n_views]:
        t = {k: torch.tensor(v, requires_grad=True) for k, v in g.items()}
        m2d = torch.zeros(P, 3, requires_grad=True)
        s = TO.Settings(res, res, cam.tanfovx, cam.tanfovy, torch.ones(3), 1.0, torch.tensor(cam.world_view_transform),
                        torch.tensor(cam.full_proj_transform), 3, torch.tensor(cam.camera_center), False, False)
        t0 = time.perf_counter()
        img, radii, da = TO.rasterize(t["means3D"], m2d, t["opacities"], shs=t["shs"], scales=t["scales"],
                                      rotations=t["rotations"], settings=s)
        ((img * gi).sum() + (da * gda).sum()).backward()
        t_tot += time.perf_counter() - t0
        done += 1
        if t_tot > budget_s:
            break
    return {"value": round(done / t_tot, 5), "unit": "views/s", "views_timed": done, "seconds": round(t_tot, 1),
            "config": f"{P} Gaussians @{res}x{res}, K=16, SH degree 3"}


def _cpu_legs_child(leg, P, K, D, H, W, init_opacity, budget_s, threads):
    """Runs in a SUBPROCESS (python bench.py --cpu-legs ...) under a hard timeout: one all-core CPU figure per process.
    A CPU leg must never be able to stall the bench line (256-thread OpenMP / torch thread pools on an unknown host)."""
    from dreamscene_amd import synth
    if leg == "omp":
        from oracle import c_oracle as CO
        g = synth.g_object(P, seed=0, K=K, init_opacity=init_opacity)
        cams = synth.object_cameras(8, H, W)
        gi_np, gda_np = synth.upstream_grads(H, W, seed=0)

        def one(c):
            v = CO.make_view(P, K, D, H, W, c.tanfovx, c.tanfovy, [1.0, 1.0, 1.0], c.world_view_transform,
                             c.full_proj_transform, c.camera_center)
            t0 = time.perf_counter()
            f = CO.forward(v, g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"], omp=True)
            CO.backward(v, f, gi_np, gda_np, g["means3D"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"], omp=True)
            return time.perf_counter() - t0
        one(cams[0])                                        # warm-up (thread pool, page faults)
        n_omp, dt_omp = 0, 0.0
        for c2 in cams:
            dt_omp += one(c2)
            n_omp += 1
            if dt_omp > budget_s:
                break
        print("CPU_LEG " + json.dumps({"value": round(n_omp / dt_omp, 5), "threads": CO.threads(True), "views": n_omp,
                                       "seconds": round(dt_omp, 1)}), flush=True)
        return
    # PyTorch-CPU oracle: C1 at 32 / 64 / physical-core threads (a pool of one thread per LOGICAL core makes its many small
    # ops crawl: the round-2 run with 256 threads produced no number in 50 s), then C2 at the best of them. Printed as it
    # grows: whatever finished before the parent's timeout counts.
    cand = []
    for th in (32, 64, int(threads)):
        th = max(1, min(int(th), int(threads)))
        if th not in cand:
            cand.append(th)
    out = {"tried": {}}
    best = None
    for th in cand:
        torch.set_num_threads(th)
        r = _torch_cpu_leg(10_000, 256, 3, 6.0)
        out["tried"][str(th)] = r["value"]
        if best is None or r["value"] > best[1]["value"]:
            best = (th, r)
        out.update(threads=best[0], C1=best[1])
        print("CPU_LEG " + json.dumps(out), flush=True)
    torch.set_num_threads(best[0])
    out["C2"] = _torch_cpu_leg(100_000, 512, 1, 20.0)
    print("CPU_LEG " + json.dumps(out), flush=True)


def _run_cpu_leg(args, timeout_s):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-legs", json.dumps(args)]
    note = None
    try:
        txt = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT).stdout
    except subprocess.TimeoutExpired as e:
        txt = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        note = f"cut off after {timeout_s} s"
    res = None
    for line in txt.splitlines():
        if line.startswith("CPU_LEG "):
            res = json.loads(line[len("CPU_LEG "):])
    return res, note


def cpu_baseline_leg(g, cam, D, K, H, W, gi_np, gda_np, hip_out, extra_cams=(), init_opacity=False, timed_path=None):
    """The CPU path timed beside the GPU numbers, on a bounded sample, host core count stated.
    The reference has NO CPU path for the rasterizer, so the baselines are this repo's CPU restatements:
      * `value`: the C port of the same algorithm (oracle/gsr_oracle.c, OpenMP build) on ALL host cores, on views of the
        SAME workload as `value` of the bench line -- kind "port";
      * `single_thread`: the scalar build of the same file (the parity checker), one thread; its first view also yields the
        HIP path's max gradient error at the full benchmark size;
      * `torch_all_cores`: the PyTorch-CPU oracle, torch.set_num_threads(all cores), at C1 (10 k @256^2) and C2 (100 k @512^2).
    The all-core legs run in a subprocess under a hard timeout (a baseline must not be able to stall the bench line)."""
    from oracle import c_oracle as CO
    CO.build()
    P = g["means3D"].shape[0]
    v = CO.make_view(P, K, D, H, W, cam.tanfovx, cam.tanfovy, [1.0, 1.0, 1.0], cam.world_view_transform,
                     cam.full_proj_transform, cam.camera_center)
    t0 = time.perf_counter()
    f = CO.forward(v, g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
    b = CO.backward(v, f, gi_np, gda_np, g["means3D"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
    dt1 = time.perf_counter() - t0                  # scalar, one thread: the checker (and the single-thread figure)
    cores = os.cpu_count()
    try:
        import psutil
        phys = int(psutil.cpu_count(logical=False) or cores)
    except Exception:
        phys = max(1, cores // 2)
    omp, omp_note = _run_cpu_leg(["omp", P, K, D, H, W, bool(init_opacity), 6.0, 0], 45)
    # the PyTorch-CPU oracle: best of {32, 64, physical cores} threads, one subprocess, hard timeout
    t_best, t_note = _run_cpu_leg(["torch", 0, 0, 0, 0, 0, False, 0.0, phys], 45)
    note = f"; all-core C leg {omp_note}" if omp_note else ""
    legs = {"omp": omp} if omp else {}
    torch_legs = dict(t_best or {}, physical_cores=phys, **({"note": t_note} if t_note else {}))
    img, da, radii, grads = hip_out
    names = ["dL_dmeans3D", "dL_dshs", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dmeans2D"]
    worst, worst_rel, worst_frac, per = 0.0, 0.0, 0.0, {}
    for n, gt in zip(names, grads):
        ref = np.asarray(b[n], dtype=np.float64).reshape(-1)
        e = np.abs(gt.detach().cpu().numpy().astype(np.float64).reshape(-1) - ref)
        mref = max(1e-6, float(np.abs(ref).max()))          # the bar: 1e-5 of the tensor's OWN largest entry (no floor of 1)
        scale = max(1.0, mref)                              # (rounds 1-5 reported against max(1, max|ref|): kept beside it)
        per[n] = {"max_err_over_max_ref": float(e.max() / mref), "max_ref": mref, "max_err_over_scale": float(e.max() / scale),
                  "frac_over_1e-5": float((e > 1e-5 * mref).mean())}
        worst = max(worst, per[n]["max_err_over_scale"])
        worst_rel = max(worst_rel, per[n]["max_err_over_max_ref"])
        worst_frac = max(worst_frac, per[n]["frac_over_1e-5"])
    d_img = np.abs(img.detach().cpu().numpy() - f["image"]).max(axis=0)
    what = f"{P} Gaussians @{W}x{H}, orbit cameras{', all opacities 0.1' if init_opacity else ''}"
    single = {"value": round(1.0 / dt1, 5), "unit": "views/s", "cores": 1,
              "sample": f"1 fwd+bwd view of the same workload, scalar build of oracle/gsr_oracle.c, {dt1:.1f} s"}
    if "omp" in legs:
        o = legs["omp"]
        base = {"value": o["value"], "unit": "views/s", "cores": o["threads"], "kind": "port",
                "sample": f"{o['views']} fwd+bwd views of the same workload ({what}) through oracle/gsr_oracle.c built with "
                          f"OpenMP ({o['threads']} threads), {o['seconds']} s; host has {cores} cores{note}"}
    else:
        base = dict(single, kind="port", sample=single["sample"] + f"; host has {cores} cores{note}")
    base["single_thread"] = single
    base["torch_all_cores"] = dict(torch_legs, cores=cores,
                                   note="PyTorch-CPU oracle (oracle/torch_oracle.py), fp32, fwd+bwd views/s at C1 = 10 k @256^2 "
                                        "(tried with 32 / 64 / physical-core threads: `tried`, views/s each) and at C2 = 100 k "
                                        "@512^2 with the best of them (`threads`)")
    err = {"bit_exact_radii": bool(np.array_equal(radii.cpu().numpy(), f["radii"])),
           "image_max_abs": float(d_img.max()), "image_frac_pixels_over_1e-5": float((d_img > 1e-5).mean()),
           "grads_max_err_over_max_ref": worst_rel, "grads_max_err_over_max1": worst,
           "grads_max_frac_entries_over_1e-5": worst_frac,
           "per_tensor": per, "tol": "1e-5 * max|ref| per tensor (max_err_over_max_ref; frac_over_1e-5 counts against it); "
                                     "max_err_over_scale = the max(1, max|ref|) figure of rounds 1-5",
           "what": "view 0 through the plain per-view module (the drop-in path) against the scalar C oracle"}
    if timed_path is not None:
        err["batched_sum"] = timed_path_check(timed_path, (f, b), g, D, K, H, W, gi_np, gda_np)
    # one word for the reader of the line: is every checked tensor inside the bar? (round 6: a stale-rows bug in the arena path showed
    # in `batched_sum` as 4e-2 on ONE tensor while `value` looked fine -- and went unnoticed until the numbers were read)
    bs = err.get("batched_sum") or {}
    worst_all = max([worst_rel] + [v["max_err_over_max_ref"] for v in (bs.get("per_tensor") or {}).values()] +
                    [bs.get(k, 0.0) for k in ("depth_alpha_max_err_over_max_ref", "means2D_grad_max_err_over_max_ref")])
    err["within_1e-5_of_own_scale"] = bool(worst_all <= 1e-5 and err["bit_exact_radii"] and err["image_max_abs"] <= 1e-5 and
                                           bs.get("bit_exact_radii_all_views", True) and bs.get("image_max_abs_all_views", 0.0) <= 1e-5)
    if not err["within_1e-5_of_own_scale"]:
        print(f"bench.py: PARITY CHECK FAILED on the line (worst {worst_all:.3e} of a tensor's own scale): max_grad_err_vs_oracle",
              file=sys.stderr, flush=True)
    return base, err


def timed_path_check(tp, view0, g, D, K, H, W, gi_np, gda_np):
    """The path that was TIMED against the oracle: the V views of one batched step (tp, captured by main() right after the
    timed region). Oracle: the scalar C build per view (view 0's results are handed in, the other views run on a thread each
    -- ctypes releases the GIL), per-view gradients summed in float64. Checked: radii bit-exact per view, every view's image /
    depth_alpha / means2D gradient, and the arena (the SUM over the views, what the optimizer / the exchange consumes)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle as CO
    P = g["means3D"].shape[0]
    cams = tp["cams"]
    V = len(cams)

    def one(c):
        v = CO.make_view(P, K, D, H, W, c.tanfovx, c.tanfovy, [1.0, 1.0, 1.0], c.world_view_transform,
                         c.full_proj_transform, c.camera_center)
        f = CO.forward(v, g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
        b = CO.backward(v, f, gi_np, gda_np, g["means3D"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
        return f, b
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=max(1, V - 1)) as ex:
        rest = list(ex.map(one, cams[1:]))
    res = [view0] + rest
    dt = time.perf_counter() - t0
    names = {"means3D": "dL_dmeans3D", "shs": "dL_dshs", "opacities": "dL_dopacity", "scales": "dL_dscales",
             "rotations": "dL_drotations"}
    per, worst, worst_rel, worst_frac = {}, 0.0, 0.0, 0.0
    for an, on in names.items():
        ref = sum(np.asarray(b[on], dtype=np.float64).reshape(-1) for _, b in res)
        e = np.abs(tp["arena"][an].astype(np.float64).reshape(-1) - ref)
        mref = max(1e-6, float(np.abs(ref).max()))
        scale = max(1.0, mref)
        per[on] = {"max_err_over_max_ref": float(e.max() / mref), "max_ref": mref, "max_err_over_scale": float(e.max() / scale),
                   "frac_over_1e-5": float((e > 1e-5 * mref).mean())}
        worst, worst_frac = max(worst, per[on]["max_err_over_scale"]), max(worst_frac, per[on]["frac_over_1e-5"])
        worst_rel = max(worst_rel, per[on]["max_err_over_max_ref"])
    img_err, da_err, m2d_err, radii_ok = 0.0, 0.0, 0.0, True
    for k, (f, b) in enumerate(res):
        img_err = max(img_err, float(np.abs(tp["images"][k] - f["image"]).max()))
        da_ref = f["depth_alpha"]
        da_err = max(da_err, float(np.abs(tp["depth_alphas"][k] - da_ref).max() / max(1e-6, float(np.abs(da_ref).max()))))
        ref = np.asarray(b["dL_dmeans2D"], dtype=np.float64)
        m2d_err = max(m2d_err, float(np.abs(tp["means2D"][k].astype(np.float64) - ref).max() / max(1e-6, float(np.abs(ref).max()))))
        radii_ok = radii_ok and bool(np.array_equal(tp["radii"][k], f["radii"]))
    return {"through": tp["through"], "views": V, "bit_exact_radii_all_views": radii_ok,
            "image_max_abs_all_views": img_err, "depth_alpha_max_err_over_max_ref": da_err,
            "means2D_grad_max_err_over_max_ref": m2d_err,
            "arena_sum_max_err_over_max_ref": worst_rel, "arena_sum_max_err_over_max1": worst,
            "arena_sum_max_frac_entries_over_1e-5": worst_frac, "per_tensor": per,
            "oracle_seconds": round(dt, 1), "tol": "1e-5 * max|ref| per tensor",
            "what": f"one step exactly as timed ({V} views through ONE batched call, gradients summed in the arena) against "
                    "the scalar C oracle's per-view results and the float64 sum of its per-view gradients"}


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--cpu-legs":
        _cpu_legs_child(*json.loads(sys.argv[2]))
    else:
        main()
