#!/usr/bin/env python
"""bench.py -- fwd+bwd views/s of the MI355X-native Gaussian rasterizer on BASELINE.json's metric config.

A "step" = one pass of the hot path over one batch of views: every GPU renders `--views-per-step` views
(default 4 = the reference's C_batch_size, configs/objects/sample.yaml:60, training/object_trainer.py:302-382:
4 views are rendered and their gradients accumulated before each optimizer step), each view being one
GaussianRasterizer forward (K1-K6) + backward (K7-K8) from fixed upstream gradients on image and depth_alpha,
through the same nn.Module / autograd boundary the reference calls (scene_gaussian.py:966-1021). The per-view
parameter gradients are summed on the device (K8 accumulate mode) and, with N GPUs, the sums are combined by ONE
in-place RCCL all-reduce per step (weak scaling: per-GPU work is fixed). `value` = views/s over all ranks.
Inputs are synthetic (dreamscene_amd/synth.py), resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` for the dominant kernel
(HIP events on the launch stream, via the library's GsrProfile) and `cpu_baseline` (scalar C oracle, N=1 only).
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--fwd-mode", type=int, default=None, help="force the forward compositing variant (0 / 1)")
    ap.add_argument("--unbatched", action="store_true",
                    help="render the views of a step one call at a time (GaussianRasterizer) instead of through "
                         "GaussianRasterizerViews (same kernels; the depth sorts of all views share their launches)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from dreamscene_amd import _lib, multiview, rasterizer as R, synth
    from dreamscene_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

    _lib.load()   # fail loudly if the HIP library is missing: there is no fallback
    R.FWD_MODE = args.fwd_mode
    H = W = args.res
    if args.scene == "object":
        K, D = 16, args.sh_degree
        g = synth.g_object(args.gaussians, seed=0, K=K, init_opacity=args.init_opacity)
        cams = synth.object_cameras(8, H, W)
        workload = f"C3: G-object {args.gaussians} Gaussians (K=16, SH degree {D}), 1 orbit view/GPU @{W}x{H}, fwd+bwd"
    else:
        K, D = 4, 1
        g = synth.g_indoor(seed=0, per_wall=max(1, args.gaussians // 5), K=K)
        cams = synth.indoor_cameras(8, H, W)
        workload = f"G-indoor {g['means3D'].shape[0]} Gaussians (K=4, SH degree 1), 1 in-room view/GPU @{W}x{H}, fwd+bwd"
    P = g["means3D"].shape[0]
    V = max(1, args.views_per_step)
    params = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in g.items()}
    gi_np, gda_np = synth.upstream_grads(H, W, seed=rank)
    gi, gda = torch.tensor(gi_np, device=dev), torch.tensor(gda_np, device=dev)
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
    # view j of rank r: camera (r * V + j) of the orbit; view 0 of rank 0 is the C3 camera
    my_cams = [cams[(rank * V + j) % len(cams)] for j in range(V)]
    cam = my_cams[0]
    rasts = []
    for c in my_cams:
        st_ = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=t([1.0, 1.0, 1.0]),
            scale_modifier=1.0, viewmatrix=t(c.world_view_transform), projmatrix=t(c.full_proj_transform),
            sh_degree=D, campos=t(c.camera_center), prefiltered=False, score_flag=False)
        rasts.append(GaussianRasterizer(raster_settings=st_))
    settings = rasts[0].raster_settings
    arena = multiview.GradArena(P, K, dev)
    R.GRAD_ARENA = arena
    leaves = [params[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]

    from dreamscene_amd.views import GaussianRasterizerViews
    rast_views = GaussianRasterizerViews([r.raster_settings for r in rasts])

    def step_batched():
        means2D = torch.zeros((V,) + tuple(params["means3D"].shape), device=dev, requires_grad=True)
        outs = rast_views(means3D=params["means3D"], means2D=means2D, shs=params["shs"], colors_precomp=None,
                          opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"],
                          cov3D_precomp=None)
        R.ACCUMULATE = False             # view 0 overwrites the arena, views 1..V-1 are added on the device
        grads = torch.autograd.grad([t for (img, _, da) in outs for t in (img, da)], leaves + [means2D], [gi, gda] * V)
        multiview.allreduce_grads(arena)
        img, radii, da = outs[0]
        g0 = list(grads[:-1]) + [grads[-1][0]]
        return (img, da, radii, g0)

    def step():
        if V > 1 and not args.unbatched and not capture[0]:   # (the parity capture wants view 0's own gradients)
            return step_batched()
        out0 = None
        for j, rast in enumerate(rasts):
            means2D = torch.zeros_like(params["means3D"], requires_grad=True)
            img, radii, da = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"], colors_precomp=None,
                                  opacities=params["opacities"], scales=params["scales"],
                                  rotations=params["rotations"], cov3D_precomp=None)
            R.ACCUMULATE = j > 0          # views 2..V are added to the arena on the device
            grads = torch.autograd.grad([img, da], leaves + [means2D], [gi, gda])
            if j == 0:
                out0 = (img, da, radii, [x.clone() for x in grads] if capture[0] else grads)
        R.ACCUMULATE = False
        multiview.allreduce_grads(arena)
        return out0

    capture = [False]

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
    stage_ms = {}
    dominant = None
    if prof is not None:
        # stage pass (untimed, after the warmup so that first-launch costs stay out of it): every stage timer on
        R.PROFILE = prof
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
        prof.set_stages([dominant])      # timed region records only the dominant kernel's events ...
        prof.set_sampling(3)             # ... of every 3rd launch (two event records per launch are not free)

    sync()
    R.HOST_WAIT_S[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    host_wait_s = R.HOST_WAIT_S[0]                   # of which: blocked on the pair counts of the projection
    host_enqueue_s = time.perf_counter() - t0        # the host is done enqueueing; the GPU may still be working
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    N_pairs = None
    roofline = None
    if prof is not None:
        res = prof.collect()
        ms, cnt = res[dominant]
        R.PROFILE = None
        # pair count of this rank's view (for the algorithmic byte count)
        with torch.no_grad():
            o, _ = R.rasterize_forward_raw(settings, params["means3D"], params["opacities"], params["shs"], None,
                                           params["scales"], params["rotations"], None)
        N_pairs = int(o["N"])
        if cnt:
            avg_s = ms / cnt * 1e-3                   # the stage timer brackets exactly one launch of the kernel
            per_launch = V if (V > 1 and not args.unbatched) else 1     # batched: one launch covers the step's V views
            ab = algorithmic_bytes(dominant, P, N_pairs, H * W, K, D, views=per_launch)
            achieved = ab / avg_s / 1e9
            traffic = None
            tf = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tf):
                try:
                    traffic = json.load(open(tf)).get(f"{args.scene}_{P}_{W}", {}).get(dominant)
                    traffic = traffic * per_launch if traffic is not None else None   # (stored per view)
                except Exception:
                    traffic = None
            # the whole path's algorithmic bytes per view (848 P + 124 N + 56 HW at K=16, D=3)
            S_ = 12 * (D + 1) ** 2
            e2e_bytes = P * (44 + S_) + P * 48 + N_pairs * 12 + N_pairs * 24 + N_pairs * 44 + H * W * 28 + \
                N_pairs * 44 + H * W * 28 + P * 40 + P * (44 + S_ + 40) + P * (44 + 12 * K + 12)
            roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "avg_launch_us": round(avg_s * 1e6, 2), "views_per_launch": per_launch,
                        "whole_path": {"algorithmic_bytes_per_view": int(e2e_bytes),
                                       "achieved_GBps": round(e2e_bytes * world * args.steps * V / elapsed / 1e9 / world, 1),
                                       "frac_of_hbm_peak": round(e2e_bytes * args.steps * V / elapsed / 1e9 / HBM_PEAK_GBS, 4)},
                        "algorithmic_bytes": int(ab),
                        "stage_us_per_view": {s: round(v * 1e3, 2) for s, v in stage_ms.items()}}

    cpu_baseline = None
    grad_err = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        capture[0] = True                 # one more (untimed) step keeping view 0's own gradients for the check
        out = step()
        torch.cuda.synchronize(dev)
        cpu_baseline, grad_err = cpu_baseline_leg(g, cam, D, K, H, W, gi_np, gda_np, out, extra_cams=cams[1:])

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
            "host_enqueue_ms_per_step": round(host_enqueue_s / args.steps * 1e3, 4),
            "host_wait_ms_per_step": round(host_wait_s / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "gaussians": P, "resolution": [H, W], "tile_pairs_N": N_pairs,
                       "views_per_step_per_gpu": V,
                       "parallelism": f"{V} view(s)/GPU/step x {world} GPUs, gradients summed on the device, then 1 RCCL "
                                      f"all-reduce of {arena.nbytes()} B per step" if world > 1 else
                                      f"single GPU, {V} view(s) per step, gradients summed on the device"},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "max_grad_err_vs_oracle": grad_err,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline_leg(g, cam, D, K, H, W, gi_np, gda_np, hip_out, hip_n_contrib=None, extra_cams=()):
    """Times the scalar C oracle (a CPU port of the same algorithm; the reference has no CPU path)
    on a bounded sample of the same workload (fwd+bwd views of the orbit, single thread, ~10 s), and reuses the first
    view to report the HIP path's max gradient error at the full benchmark size."""
    from oracle import c_oracle as CO
    CO.build()
    P = g["means3D"].shape[0]
    v = CO.make_view(P, K, D, H, W, cam.tanfovx, cam.tanfovy, [1.0, 1.0, 1.0], cam.world_view_transform,
                     cam.full_proj_transform, cam.camera_center)
    t0 = time.perf_counter()
    f = CO.forward(v, g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
    b = CO.backward(v, f, gi_np, gda_np, g["means3D"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
    dt = time.perf_counter() - t0
    n_timed, dt_total = 1, dt
    for c2 in extra_cams:                 # more views of the same orbit: a steadier figure (bounded: ~1 s each)
        v2 = CO.make_view(P, K, D, H, W, c2.tanfovx, c2.tanfovy, [1.0, 1.0, 1.0], c2.world_view_transform,
                          c2.full_proj_transform, c2.camera_center)
        t1 = time.perf_counter()
        f2 = CO.forward(v2, g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
        CO.backward(v2, f2, gi_np, gda_np, g["means3D"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
        dt_total += time.perf_counter() - t1
        n_timed += 1
        if dt_total > 12.0:
            break
    img, da, radii, grads = hip_out
    names = ["dL_dmeans3D", "dL_dshs", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dmeans2D"]
    worst, worst_frac, per = 0.0, 0.0, {}
    for n, gt in zip(names, grads):
        ref = np.asarray(b[n], dtype=np.float64).reshape(-1)
        e = np.abs(gt.detach().cpu().numpy().astype(np.float64).reshape(-1) - ref)
        scale = max(1.0, float(np.abs(ref).max()))
        per[n] = {"max_err_over_scale": float(e.max() / scale), "frac_over_1e-5": float((e > 1e-5 * scale).mean())}
        worst = max(worst, per[n]["max_err_over_scale"])
        worst_frac = max(worst_frac, per[n]["frac_over_1e-5"])
    d_img = np.abs(img.detach().cpu().numpy() - f["image"]).max(axis=0)
    nc_diff = int((hip_n_contrib != f["n_contrib"]).sum()) if hip_n_contrib is not None else None
    base = {"value": round(n_timed / dt_total, 5), "unit": "views/s", "cores": 1, "kind": "port",
            "sample": f"{n_timed} fwd+bwd views of the same workload ({P} Gaussians @{W}x{H}, orbit cameras) through "
                      f"oracle/gsr_oracle.c, single thread, {dt_total:.1f} s; host has {os.cpu_count()} cores"}
    # hard gates (alpha < 1/255, T < 1e-4) put a few (pixel, splat) pairs on the other side of a rounding
    # difference at this size (SEMANTICS.md section 6): report how many pixels / entries, not only the max
    return base, {"bit_exact_radii": bool(np.array_equal(radii.cpu().numpy(), f["radii"])),
                  "image_max_abs": float(d_img.max()), "image_frac_pixels_over_1e-5": float((d_img > 1e-5).mean()),
                  "grads_max_err_over_max1": worst, "grads_max_frac_entries_over_1e-5": worst_frac,
                  "per_tensor": per, "tol": "1e-5 * max(1, max|ref|)"}


if __name__ == "__main__":
    main()
