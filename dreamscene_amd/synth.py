"""Seeded synthetic workloads (host numpy; independent of the reference's Python) -- SURVEY.md section 8(d).

G-object(P, seed) mirrors the statistics of the reference's object initialisation:
  4096 seed points ~ U(ball r=0.5) (config.py:220-221, gs_renderer.py:353-369), each replicated with
  U(ball r<=0.05) jitter (gs_renderer.py:380-398); isotropic log-scale = log sqrt(mean sq. dist to 3 NN)
  (gs_renderer.py:590-594) + N(0,0.3) per axis; random unit quaternions; opacity logits N(0,1.5)
  (or the all-0.1 init value, gs_renderer.py:598); SH DC = RGB2SH(U(0,1)) (utils/sh_utils.py:122-123),
  higher bands N(0,0.05).
G-indoor(seed): 5 wall sheets on the faces of the room box [-3.5,-2.5,0 -> 3.5,2.5,5]
  (configs/scenes/sample_indoor.yaml:216, gs_renderer.py:221-235) with +-1/50 jitter, K=4.
All tensors are the *activated* values the rasterizer boundary receives (post exp / sigmoid / normalize).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np

from .camera import Camera, look_at_camera, orbit_camera

SH_C0 = 0.28209479177387814


def _ball(rng, n, r):
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v * (r * rng.uniform(size=(n, 1)) ** (1.0 / 3.0))


def _knn_scale(xyz: np.ndarray) -> np.ndarray:
    from scipy.spatial import cKDTree
    d, _ = cKDTree(xyz).query(xyz, k=4, workers=-1)
    d2 = np.clip((d[:, 1:] ** 2).mean(axis=1), 1e-7, None)
    return np.sqrt(d2)


def g_object(P: int, seed: int = 0, K: int = 16, init_opacity: bool = False) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    n_seed = min(4096, P)
    seeds = _ball(rng, n_seed, 0.5)
    rep = -(-P // n_seed)
    xyz = (np.repeat(seeds, rep, axis=0) + _ball(rng, n_seed * rep, 0.05))[:P]
    if rep == 1:
        xyz = seeds[:P]
    base = _knn_scale(xyz)
    scales = np.exp(np.log(base)[:, None] + rng.normal(scale=0.3, size=(P, 3)))
    q = rng.normal(size=(P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    if init_opacity:
        opac = np.full((P, 1), 0.1)
    else:
        opac = 1.0 / (1.0 + np.exp(-rng.normal(scale=1.5, size=(P, 1))))
    shs = rng.normal(scale=0.05, size=(P, K, 3))
    shs[:, 0, :] = (rng.uniform(size=(P, 3)) - 0.5) / SH_C0
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return dict(means3D=f(xyz), scales=f(scales), rotations=f(q), opacities=f(opac), shs=f(shs))


def g_indoor(seed: int = 0, per_wall: int = 400_000, K: int = 4) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    lo, hi = np.array([-3.5, -2.5, 0.0]), np.array([3.5, 2.5, 5.0])
    sheets = []
    faces = [(0, lo[0]), (0, hi[0]), (1, lo[1]), (1, hi[1]), (2, hi[2])]   # 4 walls + ceiling
    for axis, val in faces:
        p = rng.uniform(lo, hi, size=(per_wall, 3))
        p[:, axis] = val + rng.uniform(-0.02, 0.02, size=per_wall)
        sheets.append(p)
    xyz = np.concatenate(sheets, axis=0)
    P = xyz.shape[0]
    base = _knn_scale(xyz)
    scales = np.exp(np.log(base)[:, None] + rng.normal(scale=0.3, size=(P, 3)))
    q = rng.normal(size=(P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = 1.0 / (1.0 + np.exp(-rng.normal(scale=1.5, size=(P, 1))))
    shs = rng.normal(scale=0.05, size=(P, K, 3))
    shs[:, 0, :] = (rng.uniform(size=(P, 3)) - 0.5) / SH_C0
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return dict(means3D=f(xyz), scales=f(scales), rotations=f(q), opacities=f(opac), shs=f(shs))


def object_cameras(n: int, H: int, W: int, radius: float = 5.35, theta: float = 75.0, fovx: float = 0.46) -> List[Camera]:
    """Orbit cameras, azimuth 45deg*i (radius / FoV mid-range of config.py:88-99)."""
    return [orbit_camera(radius, theta, 45.0 * i, fovx, H, W) for i in range(n)]


def sphere_cameras(n: int, H: int, W: int, radius: float = 5.35, fovx: float = 0.46) -> List[Camera]:
    """n cameras spread over the sphere around the object, looking at the origin -- the kind of set the reference's
    importance scoring renders (loadSphereCam, utils/cam_utils.py:1847: 48 cameras). Fibonacci lattice, poles avoided."""
    cams = []
    for i in range(n):
        z = 1.0 - 2.0 * (i + 0.5) / n
        theta = float(np.degrees(np.arccos(np.clip(z, -0.985, 0.985))))
        phi = float((i * 137.50776405) % 360.0)
        cams.append(orbit_camera(radius, theta, phi, fovx, H, W))
    return cams


def indoor_cameras(n: int, H: int, W: int, fovx: float = 0.96) -> List[Camera]:
    """Cameras inside the room looking outward (utils/cam_utils.py:952, 2278-2327)."""
    cams = []
    for i in range(n):
        a = 2.0 * np.pi * i / max(n, 1)
        eye = np.array([0.9 * np.cos(a), 0.6 * np.sin(a), 2.5])
        tgt = eye + np.array([np.cos(a), np.sin(a), 0.0])
        cams.append(look_at_camera(eye, tgt, fovx, H, W))
    return cams


def upstream_grads(H: int, W: int, seed: int = 0):
    """dL/dimage [3,H,W], dL/d(depth_alpha) [2,H,W] ~ N(0,1)*1e-3."""
    rng = np.random.default_rng(1000 + seed)
    return (rng.normal(size=(3, H, W)) * 1e-3).astype(np.float32), (rng.normal(size=(2, H, W)) * 1e-3).astype(np.float32)
