"""ORACLE (test infrastructure, NOT product code): mean squared distance to the 3 nearest neighbours, the quantity
`simple_knn._C.distCUDA2` returns (gs_renderer.py:590-593; the CUDA source is un-vendored, README.md:48: parity
unpinned, the definition is the one the call site relies on: `scales = log(sqrt(dist2))`).
numpy / scipy on the CPU; brute force for small inputs, cKDTree otherwise."""
import numpy as np


def mean_dist2_brute(points: np.ndarray) -> np.ndarray:
    p = np.asarray(points, dtype=np.float32)
    n = p.shape[0]
    out = np.empty(n, np.float32)
    for i in range(n):
        d = p - p[i]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        d2[i] = np.float32(np.finfo(np.float32).max)
        best = np.sort(d2)[:3].astype(np.float32)
        if best.shape[0] < 3:
            best = np.concatenate([best, np.full(3 - best.shape[0], np.finfo(np.float32).max, np.float32)])
        out[i] = (best[0] + best[1] + best[2]) / np.float32(3.0)
    return out


def mean_dist2(points: np.ndarray) -> np.ndarray:
    p = np.asarray(points, dtype=np.float64)
    if p.shape[0] <= 2048:
        return mean_dist2_brute(points)
    from scipy.spatial import cKDTree
    d, _ = cKDTree(p).query(p, k=4, workers=-1)
    return (d[:, 1:] ** 2).mean(axis=1).astype(np.float32)
