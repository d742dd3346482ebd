"""Host-side pieces of the post-raster epilogue that are not per-step kernels.

* `DensifyStats`: the three per-Gaussian statistics tensors of GaussianModel (`max_radii2D`, `xyz_gradient_accum`,
  `denom`; gs_renderer.py:612-613, 1061-1065) updated INSIDE the rasterizer backward (K8) for the views whose FORWARD
  runs under `with stats.collect(context):` -- instead of five boolean-mask indexing kernels after `loss.backward()`
  (object_trainer.py:386-390). With several views per call the LAST view counts (that is what the reference's trainers
  do with the loop's last viewspace_points / visibility_filter / radii); `views="all"` or a list of indices opts in to
  more (then denom grows once per counted view and max_radii2D is the maximum over them).
* `importance_prune_mask`: the 3D-Gaussian-filtering threshold of `calculate_v_imp_score` + `prune_gaussians`
  (scene_gaussian.py:1046-1061, gs_renderer.py:1082-1087) with two k-th order statistics instead of two full sorts.
"""
from __future__ import annotations

import contextlib

import torch

from . import rasterizer as R


class DensifyStats:
    def __init__(self, P: int, device):
        self.max_radii2D = torch.zeros(P, dtype=torch.float32, device=device)
        self.xyz_gradient_accum = torch.zeros(P, dtype=torch.float32, device=device)
        self.denom = torch.zeros(P, dtype=torch.float32, device=device)

    def tensors(self) -> tuple:
        return (self.max_radii2D, self.xyz_gradient_accum, self.denom)

    @contextlib.contextmanager
    def collect(self, context: "R.RasterContext", views=None):
        """Rasterizer calls using `context` whose FORWARD runs inside this block update the statistics in their backward
        (visible Gaussians only; the forward takes a snapshot of the context, so the backward may run later and on
        autograd's thread). views: for multi-view calls, which views count -- None = the last one, "all", or indices."""
        prev = (context.densify_stats, context.stats_views)
        context.densify_stats, context.stats_views = self.tensors(), views
        try:
            yield self
        finally:
            context.densify_stats, context.stats_views = prev

    def mean_grad(self) -> torch.Tensor:
        """grads = xyz_gradient_accum / denom with NaN -> 0 (gs_renderer.py:1035-1036)."""
        g = self.xyz_gradient_accum / self.denom
        g[g.isnan()] = 0.0
        return g


def v_importance(scaling_activated: torch.Tensor, imp_list: torch.Tensor, v_pow: float) -> torch.Tensor:
    """scene_gaussian.py:1046-1061: (volume / (volume at 90 % of the descending order)) ** v_pow * importance."""
    volume = torch.prod(scaling_activated, dim=1)
    n = volume.shape[0]
    index = int(n * 0.9)
    # sorted_descending[index] == the (n - index)-th smallest value
    kth_percent_largest = torch.kthvalue(volume, n - index).values
    return torch.pow(volume / kth_percent_largest, v_pow) * imp_list


def importance_prune_mask(v_list: torch.Tensor, percent: float) -> torch.Tensor:
    """gs_renderer.py:1082-1087: prune everything at or below the value at int(percent * (n - 1)) of the ascending order."""
    n = v_list.shape[0]
    index_nth_percentile = int(percent * (n - 1))
    value_nth_percentile = torch.kthvalue(v_list.reshape(-1), index_nth_percentile + 1).values
    return (v_list <= value_nth_percentile).squeeze()
