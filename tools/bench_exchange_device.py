"""The DEVICE-side halves of multiview.GradExchange's wire formats, timed on one GPU: everything a format
does besides the collective itself -- finding the non-zero rows (K8's reached bitmap -> indices), gathering them into
(index, row) messages, the strided pack / unpack of the active SH columns at D < 3, adding / storing received rows, the local
sum of `direct`, clearing the arena, and the host reads each format needs. The arena is a REAL one: the sum of a rank's 4 views
of the configuration (K8 accumulate), so the row sets are what an exchange would see. W = 8 ranks are emulated by applying this
rank's own message W times (same sizes, same kernels). Prints one JSON object; times in microseconds (CUDA events, median of
`reps`). usage: python tools/bench_exchange_device.py [--gaussians P] [--res R] [--init-opacity]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamscene_amd import _lib, multiview, synth  # noqa: E402
from dreamscene_amd.rasterizer import GaussianRasterizationSettings, RasterContext  # noqa: E402
from dreamscene_amd.views import GaussianRasterizerViews  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=500_000)
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--views", type=int, default=4)
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--reps", type=int, default=9)
ap.add_argument("--init-opacity", action="store_true")
args = ap.parse_args()
P, res, V, W, K, D = args.gaussians, args.res, args.views, args.world, 16, 3
dev = torch.device("cuda", 0)
_lib.load()
g = synth.g_object(P, seed=0, K=K, init_opacity=args.init_opacity)
cams = synth.object_cameras(8, res, res)[:V]
params = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in g.items()}
gi, gda = (torch.tensor(x, device=dev) for x in synth.upstream_grads(res, res, seed=0))
t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
sl = [GaussianRasterizationSettings(image_height=res, image_width=res, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=t([1, 1, 1]),
                                    scale_modifier=1.0, viewmatrix=t(c.world_view_transform),
                                    projmatrix=t(c.full_proj_transform), sh_degree=D, campos=t(c.camera_center),
                                    prefiltered=False, score_flag=False) for c in cams]
arena = multiview.GradArena(P, K, dev)
rast = GaussianRasterizerViews(sl, context=RasterContext(grad_arena=arena))
for _ in range(3):
    m2d = torch.zeros((V, P, 3), device=dev, requires_grad=True)
    outs = rast(means3D=params["means3D"], means2D=m2d, opacities=params["opacities"], shs=params["shs"], scales=params["scales"],
                rotations=params["rotations"])
    torch.autograd.grad([x for (img, _, da) in outs for x in (img, da)], [m2d], [gi, gda] * V)
torch.cuda.synchronize()
snapshot = arena.flat.clone()


def timed(fn, reps=args.reps, restore=False):
    """median GPU time (events) and median host wall time (incl. any host read inside fn) in us"""
    gpu, wall = [], []
    for _ in range(reps + 2):
        if restore:
            arena.flat.copy_(snapshot)
            arena.reached_valid = True
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        r = fn()
        e1.record()
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) * 1e6)
        gpu.append(e0.elapsed_time(e1) * 1e3)
    gpu, wall = sorted(gpu[2:]), sorted(wall[2:])
    return {"gpu_us": round(gpu[len(gpu) // 2], 1), "wall_us": round(wall[len(wall) // 2], 1)}, r


res_ = {"P": P, "res": res, "views_per_rank": V, "world_emulated": W, "init_opacity": bool(args.init_opacity)}
out = {}
# ---- the HIP library's pack / unpack (csrc/exchange.hip), what GradExchange uses on a cuda arena
hip = {}
for deg in (3, 0):
    ex = multiview.GradExchange(arena, sh_degree=deg, mode="rows")
    F = ex.row_floats
    arena.reached_valid = True
    h = {"row_floats": F}
    h["_message (bitmap -> indices + rows, one host read)"], (idx_h, rows_h) = timed(ex._message, restore=True)
    h["rows"] = int(idx_h.numel())
    idx_h, rows_h = idx_h.clone(), rows_h.clone()
    h["_add_rows x1"], _ = timed(lambda: ex._add_rows(idx_h, rows_h), restore=True)
    h["_set_rows x1"], _ = timed(lambda: ex._set_rows(idx_h, rows_h), restore=True)
    bper = (P + W - 1) // W
    lo_n = int((idx_h < bper).sum())

    def owner():
        mine = torch.zeros((bper, F), device=dev)
        own_set = ex._dev_rows.rowset([(mine, F, F)], bper)
        bits = torch.zeros((bper + 63) // 64, dtype=torch.int64, device=dev)
        for _ in range(W):
            ex._dev_rows.unpack(own_set, idx_h[:lo_n], rows_h[:lo_n], mode=0, row_base=0, touched=bits)
        return multiview._DeviceRows(dev).pack(own_set, bits, F, lo_n)
    h["sparse_rs owner side (W messages added into the owned slice, touched rows packed; one host read)"], _ = timed(owner)
    g_ = lambda k: h[k]["gpu_us"]
    w_ = lambda k: h[k]["wall_us"]
    km, ka, ks, ko = (next(k for k in h if k.startswith(p_)) for p_ in ("_message", "_add_rows", "_set_rows", "sparse_rs owner"))
    zero_us = 18.4
    h["device_side_total_us"] = {"rows": round(w_(km) + zero_us + W * g_(ka), 1),
                                 "sparse_rs": round(w_(km) + w_(ko) + zero_us + 2.0 * g_(ks), 1)}
    hip[f"D{deg}"] = h
res_["hip_kernels"] = hip
# ---- round 6: self-describing row messages (gsr_rowmsg_pack / gsr_rowmsg_apply): the `rows` format without a host read. The W
# ranks are REAL here: rank r's arena is the sum of its own V views (cameras r V .. r V + V - 1 of a 360-degree orbit), so the union
# of the ranks' rows -- what the apply kernel writes -- is what an 8-GPU step would see.
rm = {}
reached0 = arena.reached.clone()
orbit = synth.object_cameras(W * V, res, res)
for deg in (3, 0):
    ex = multiview.GradExchange(arena, sh_degree=deg, mode="rows", strict=False)
    F = ex.row_floats
    msgs = multiview._RowMessages(dev)
    cap = (max(P // 4, 1024) + 1023) // 1024 * 1024
    msg, allm, nbytes = msgs.buffers(P, F, W, cap)
    counts = []
    # (and the slice messages of the sparse reduce-scatter, capacities as the policy would set them for these counts)
    per = ex._slice_rows(W)
    cap1 = (int(0.2 * per * 1.25) + 1535) // 512 * 512
    cap2 = (int(0.3 * per * 1.25) + 1535) // 512 * 512
    sl_msgs = multiview._RowMessages(dev)
    sl_msgs.slice_buffers(P, F, W, per, cap1, cap2)
    sends = []
    for r in range(W):
        sl_r = [GaussianRasterizationSettings(image_height=res, image_width=res, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=t([1, 1, 1]),
                                              scale_modifier=1.0, viewmatrix=t(c.world_view_transform),
                                              projmatrix=t(c.full_proj_transform), sh_degree=D, campos=t(c.camera_center),
                                              prefiltered=False, score_flag=False) for c in orbit[r * V:(r + 1) * V]]
        rast_r = GaussianRasterizerViews(sl_r, context=RasterContext(grad_arena=arena))
        for _ in range(2):
            m2d = torch.zeros((V, P, 3), device=dev, requires_grad=True)
            outs = rast_r(means3D=params["means3D"], means2D=m2d, opacities=params["opacities"], shs=params["shs"],
                          scales=params["scales"], rotations=params["rotations"])
            torch.autograd.grad([x for (img, _, da) in outs for x in (img, da)], [m2d], [gi, gda] * V)
        torch.cuda.synchronize()
        msgs.pack(ex._arena_rowset(), arena.reached, cap)
        allm[r * nbytes:(r + 1) * nbytes].copy_(msg)
        counts.append(int(msg[:4].view(torch.int32).item()))
        sl_msgs.pack_slices(ex._arena_rowset(), arena.reached, W, per, cap1)
        sends.append(sl_msgs.send1.clone())
    own = arena.flat.clone()                       # (the last rank's arena: the one the timed calls run on)

    def restore_own():
        arena.flat.copy_(own)
    h = {"row_floats": F, "cap_rows": cap, "rows_per_rank": counts, "message_bytes": nbytes}
    rs_ = ex._arena_rowset()

    def timed_own(fn):
        gpu = []
        for _ in range(args.reps + 2):
            torch.cuda.synchronize()
            # (no synchronisation between the restore and the timed call: the 118 MB copy keeps the GPU busy while the host enqueues
            #  the call, so the events bracket the kernel and not the host's launch path -- the first numbers of this section did
            #  include it: ~30-60 us of Python + ctypes per call on an idle GPU)
            restore_own(); restore_own()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            gpu.append(e0.elapsed_time(e1) * 1e3)
        gpu = sorted(gpu[2:])
        return round(gpu[len(gpu) // 2], 1)
    h["pack_us (one launch)"] = timed_own(lambda: msgs.pack(rs_, arena.reached, cap))
    h["apply_us (one launch, W messages, rank-ordered sums stored)"] = timed_own(lambda: msgs.apply(rs_, W, cap))
    ok, worst = msgs.result()
    h["applied"], h["rows_max"] = bool(ok), int(worst)
    union = torch.zeros_like(arena.reached)
    lay_bitmap = 256
    for r in range(W):
        union |= allm[r * nbytes + lay_bitmap:r * nbytes + lay_bitmap + arena.reached.numel() * 8].view(torch.int64)
    sh = torch.arange(64, device=dev, dtype=torch.int64)
    h["union_rows"] = int((((union.unsqueeze(1) >> sh) & 1).reshape(-1)[:P]).sum())
    h["device_side_total_us"] = round(h["pack_us (one launch)"] + h["apply_us (one launch, W messages, rank-ordered sums stored)"], 1)
    h["host_reads"] = 0
    # sparse_rs on the device: the all-to-all and the all-gather emulated by copies; timed on the LAST rank (owner W - 1)
    n1, n2 = sl_msgs.n1, sl_msgs.n2
    own_counts = []
    for o in range(W):
        for r in range(W):
            sl_msgs.recv1[r * n1:(r + 1) * n1].copy_(sends[r][o * n1:(o + 1) * n1])
        sl_msgs.reduce_owned(max(0, min(per, P - o * per)), per, F, W, cap1, cap2)
        sl_msgs.all2[o * n2:(o + 1) * n2].copy_(sl_msgs.own2)
        own_counts.append(int(sl_msgs.own2[:4].view(torch.int32).item()))
    s_ = {"slice_rows": per, "cap_rows": [cap1, cap2], "message_bytes": [n1, n2], "owner_union_rows": own_counts,
          "wire_bytes_per_rank": int((W - 1) * (n1 + n2))}
    s_["pack_slices_us"] = timed_own(lambda: sl_msgs.pack_slices(rs_, arena.reached, W, per, cap1))
    s_["reduce_owned_us"] = timed_own(lambda: sl_msgs.reduce_owned(max(0, min(per, P - (W - 1) * per)), per, F, W, cap1, cap2))
    s_["apply_slices_us"] = timed_own(lambda: sl_msgs.apply_slices(rs_, W, per, cap2))
    ok2, worst2 = sl_msgs.result()
    s_["applied"], s_["rows_max"] = bool(ok2), [int(sl_msgs.worst_in), int(worst2)]
    s_["device_side_total_us"] = round(s_["pack_slices_us"] + s_["reduce_owned_us"] + s_["apply_slices_us"], 1)
    h["sparse_rs_device"] = s_
    rm[f"D{deg}"] = h
res_["row_messages"] = rm
arena.flat.copy_(snapshot)
arena.reached.copy_(reached0)
arena.reached_valid = True
# ---- the torch index arithmetic they replace (the reference of tests/test_exchange_rows.py)
for deg in (3, 2, 1, 0):
    ex = multiview.GradExchange(arena, sh_degree=deg, mode="dense")
    ex._dev_rows = None
    F = ex.row_floats
    d = {"row_floats": F}
    arena.reached_valid = True
    d["nonzero_rows (bitmap -> indices, one host read inside torch.nonzero)"], idx = timed(ex.nonzero_rows)
    n = int(idx.numel())
    d["rows"] = n
    d["row_frac"] = round(n / P, 4)
    d["_rows_of (gather index+row message)"], rows = timed(lambda: ex._rows_of(idx))
    d["message_bytes"] = int(n * (4 + 4 * F))
    d["_add_rows x1 (scatter-add of one rank's message)"], _ = timed(lambda: ex._add_rows(idx, rows), restore=True)
    d["_set_rows x1 (sparse_rs: store an owner's reduced rows)"], _ = timed(lambda: ex._set_rows(idx, rows), restore=True)
    d["arena.flat.zero_()"], _ = timed(lambda: arena.flat.zero_(), restore=True)
    d["wire_buffer() (dense/direct pack; the arena itself at D = 3)"], wire = timed(ex.wire_buffer, restore=True)
    d["wire_bytes"] = int(wire.numel() * 4)
    d["_unpack"], _ = timed(lambda: ex._unpack(wire), restore=True)
    # direct: local sum of W slices of 1/W of the wire buffer, rank order
    per = (wire.numel() + W - 1) // W
    recv = torch.zeros(per * W, device=dev)

    def local_sum():
        mine = recv.view(W, per)[0].clone()
        for r in range(1, W):
            mine.add_(recv.view(W, per)[r])
        return mine
    if hip:
        d["direct: local sum of W slices"], _ = timed(lambda: multiview._sum_slices(recv, W, per))
        d["direct: local sum, W - 1 torch adds (before)"], _ = timed(local_sum)
    else:
        d["direct: local sum of W slices"], _ = timed(local_sum)
    # sparse_rs owner side: bounds by searchsorted + W index_add_ of 1/W of the rows each
    bper = (P + W - 1) // W

    def rs_owner():
        bounds = torch.searchsorted(idx, torch.arange(0, W + 1, device=dev, dtype=idx.dtype) * bper)
        sc = (bounds[1:] - bounds[:-1]).tolist()        # host read
        mine = torch.zeros((max(bper, 1), F), device=dev)
        touched = torch.zeros(max(bper, 1), dtype=torch.bool, device=dev)
        lo, hi = int(bounds[0]), int(bounds[1])
        li = idx[lo:hi]
        for _ in range(W):
            mine.index_add_(0, li, rows[lo:hi])
            touched[li] = True
        own = torch.nonzero(touched).reshape(-1)         # host read
        return sc, own
    d["sparse_rs: owner side (bounds + W index_add_ + union, 2 host reads)"], _ = timed(rs_owner)
    # totals per format at W ranks (device side only; what has to be added to the wire time)
    g_ = lambda k: d[k]["gpu_us"]
    w_ = lambda k: d[k]["wall_us"]
    nz, ro, ad, st, ze = (next(k for k in d if k.startswith(p_)) for p_ in ("nonzero_rows", "_rows_of", "_add_rows", "_set_rows", "arena.flat"))
    pk, up, ls, rs = (next(k for k in d if k.startswith(p_)) for p_ in ("wire_buffer", "_unpack", "direct:", "sparse_rs:"))
    d["device_side_total_us"] = {
        "dense": round((g_(pk) + g_(up)) if deg < 3 else 0.0, 1),
        "direct": round(((g_(pk) + g_(up)) if deg < 3 else 0.0) + g_(ls), 1),
        "rows": round(w_(nz) + g_(ro) + g_(ze) + W * g_(ad), 1),
        # (the owners' reduced rows are the UNION over the ranks' views: ~2x one rank's rows at C3 -> two messages' worth of stores)
        "sparse_rs": round(w_(nz) + g_(ro) + w_(rs) + g_(ze) + 2.0 * g_(st), 1)}
    out[f"D{deg}"] = d
res_["torch_ops_by_degree"] = out
print(json.dumps(res_))
