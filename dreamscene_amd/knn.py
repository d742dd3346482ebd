"""`distCUDA2`: mean squared distance to the 3 nearest neighbours, on the MI355X (HIP kernels in csrc/knn.hip).

Drop-in for `from simple_knn._C import distCUDA2` (gs_renderer.py:9), which the reference uses once per model
initialisation to seed the Gaussian scales (gs_renderer.py:590-594). SURVEY.md section 8(f), rank 1."""
from __future__ import annotations

import torch

from . import _lib as L


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """points [N,3] float tensor on a cuda (ROCm) device -> [N] float32: mean of the squared distances to the 3
    nearest other points (exact)."""
    lib = L.load()
    if points.device.type != "cuda":
        raise L.GsrError("distCUDA2 needs a tensor on a cuda (ROCm) device; there is no CPU fallback")
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError(f"points must be [N,3], got {tuple(points.shape)}")
    pts = points.detach().to(torch.float32).contiguous()
    n = int(pts.shape[0])
    out = torch.empty(n, dtype=torch.float32, device=pts.device)
    if n == 0:
        return out
    nbytes = int(lib.gsr_knn_scratch_bytes(n))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        stream = torch.cuda.current_stream(pts.device).cuda_stream
        L.check(lib.gsr_knn_mean_dist2(pts.data_ptr(), n, out.data_ptr(), scratch.data_ptr(), nbytes, stream),
                "gsr_knn_mean_dist2")
    return out
