"""View-level data parallelism for the rasterizer: one view per GPU, Gaussian gradients summed over ranks.

The reference renders C_batch_size views one after another against the same parameters and sums their
gradients in a single backward (training/object_trainer.py:302-382, training/scene_trainer.py:801-881); it has
no distributed code. Views are independent, so the path shards with no data-path collective in
forward/backward; the only exchange is the sum of per-view parameter gradients, one in-place all-reduce per step
over RCCL/xGMI (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).

GradArena: every parameter gradient of one view is written by the HIP backward straight into ONE flat fp32
buffer (planar layout [means3D P*3 | scales P*3 | rotations P*4 | opacities P | shs P*K*3]); the all-reduce runs
in place on that buffer and the tensors handed to autograd are views of it -- no pack / unpack copies.
Per-view densification statistics (norm of means2D.grad, radii>0, max radii; gs_renderer.py:1034-1065) are not
linear in the view, so they are reduced separately (reduce_view_stats).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist

FIELDS = (("means3D", 3), ("scales", 3), ("rotations", 4), ("opacities", 1), ("shs", None))


class GradArena:
    def __init__(self, P: int, K: int, device, dtype=torch.float32):
        self.P, self.K = int(P), int(K)
        sizes = [(n, P * (3 * K if w is None else w)) for n, w in FIELDS]
        # keep every region 16-byte aligned (the C ABI requires it for shs / rotations)
        offs, total = {}, 0
        for n, sz in sizes:
            total = (total + 3) & ~3
            offs[n] = (total, sz)
            total += sz
        self.flat = torch.zeros(total, dtype=dtype, device=device)
        shapes = dict(means3D=(P, 3), scales=(P, 3), rotations=(P, 4), opacities=(P, 1), shs=(P, K, 3))
        self.views: Dict[str, torch.Tensor] = {n: self.flat[o:o + sz].view(shapes[n]) for n, (o, sz) in offs.items()}

    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()


def allreduce_grads(arena: GradArena, group=None, async_op: bool = False):
    """Sum the packed per-view gradients over all ranks, in place. No-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None
    return dist.all_reduce(arena.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def reduce_view_stats(means2D_grad: torch.Tensor, radii: torch.Tensor, group=None):
    """Per-view densification statistics reduced so that every replica takes identical densify decisions:
    sum over views of ||means2D.grad[:, :2]||, count of views in which the Gaussian was visible, max radius."""
    norm = torch.norm(means2D_grad[:, :2], dim=-1)
    vis = (radii > 0).to(norm.dtype)
    maxr = radii.to(torch.int32).clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        packed = torch.stack([norm, vis], dim=0)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(maxr, op=dist.ReduceOp.MAX, group=group)
        norm, vis = packed[0], packed[1]
    return norm, vis, maxr


def shard_views(n_views: int, rank: int, world: int):
    """Static round-robin view -> rank assignment (view i runs on rank i % world)."""
    return [i for i in range(n_views) if i % world == rank]


def render_views_data_parallel(rasterize_view, params: Dict[str, torch.Tensor], cameras, upstream, arena: GradArena,
                               group=None):
    """Render this rank's share of `cameras` (fwd+bwd) and leave the SUM over all views of every parameter
    gradient in `arena` on every rank.

    rasterize_view(params, camera, grad_out) must run one view forward+backward and write that view's parameter
    gradients into the tensors of grad_out (a dict of arena-shaped tensors), overwriting them.
    Equivalent, to fp32 summation order, to the sequential accumulation the reference performs."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    mine = shard_views(len(cameras), rank, world)
    acc: Optional[torch.Tensor] = None
    outs = []
    for j, vi in enumerate(mine):
        outs.append(rasterize_view(params, cameras[vi], arena.views, upstream[vi]))
        if len(mine) > 1:
            acc = arena.flat.clone() if acc is None else acc.add_(arena.flat)
    if acc is not None:
        arena.flat.copy_(acc)
    if not mine:
        arena.flat.zero_()
    allreduce_grads(arena, group)
    return outs
