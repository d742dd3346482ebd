"""Host-side camera matrices in the exact layout the rasterizer boundary consumes.

Restates (does not import) the reference's camera construction so that callers / benches / tests on the GPU box
can build the same `viewmatrix`, `projmatrix`, `campos`, `tanfov` the reference feeds to
`GaussianRasterizationSettings` (scene_gaussian.py:951-964):
    world_view_transform = getWorld2View2(R, T)^T               utils/cam_utils.py:196-197, graphics_utils.py:47-58
    projection_matrix    = getProjectionMatrix(...)^T           utils/graphics_utils.py:61-81, cam_utils.py:198-204
    full_proj_transform  = world_view_transform @ projection    utils/cam_utils.py:205-209
    camera_center        = inverse(world_view_transform)[3,:3]  utils/cam_utils.py:210
    orbit pose           = circle_poses + GenSingleCam          utils/cam_utils.py:277-309, 1894-1911
Pinned against RCamera by tests/golden/cameras.npz.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

ZNEAR, ZFAR = 0.01, 100.0   # utils/cam_utils.py:182-183


def fov2focal(fov: float, pixels: float) -> float:
    return pixels / (2.0 * math.tan(fov / 2.0))


def focal2fov(focal: float, pixels: float) -> float:
    return 2.0 * math.atan(pixels / (2.0 * focal))


def world_to_view(R: np.ndarray, T: np.ndarray) -> np.ndarray:
    """W2C 4x4 (column-vector convention) from a C2W rotation R and W2C translation T."""
    Rt = np.zeros((4, 4), dtype=np.float64)
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).T
    Rt[:3, 3] = np.asarray(T, dtype=np.float64)
    Rt[3, 3] = 1.0
    # reference round-trips through inv(inv(Rt)) with translate=0, scale=1 (graphics_utils.py:53-58)
    return np.linalg.inv(np.linalg.inv(Rt)).astype(np.float32)


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> np.ndarray:
    ty, tx = math.tan(fovy / 2.0), math.tan(fovx / 2.0)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    P = np.zeros((4, 4), dtype=np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


@dataclass
class Camera:
    """What RCamera exposes to the render glue (utils/cam_utils.py:148-217), as numpy float32."""
    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    world_view_transform: np.ndarray   # [4,4], row-vector convention (transposed W2C)
    full_proj_transform: np.ndarray    # [4,4]
    camera_center: np.ndarray          # [3]

    @property
    def tanfovx(self) -> float:
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self) -> float:
        return math.tan(self.FoVy * 0.5)

    @staticmethod
    def from_RT(R, T, fovx: float, fovy: float, H: int, W: int) -> "Camera":
        wvt = world_to_view(R, T).T.copy()
        proj = projection_matrix(ZNEAR, ZFAR, fovx, fovy).T.copy()
        full = (wvt.astype(np.float32) @ proj.astype(np.float32)).astype(np.float32)
        center = np.linalg.inv(wvt.astype(np.float32))[3, :3].astype(np.float32)
        return Camera(int(H), int(W), float(fovx), float(fovy), wvt.astype(np.float32), full, center)


def orbit_pose(radius: float, theta_deg: float, phi_deg: float):
    """(R, T) of a z-up orbit camera looking at the origin; theta = polar angle from +z, phi = azimuth."""
    th, ph = math.radians(theta_deg), math.radians(phi_deg)
    c = np.array([radius * math.sin(th) * math.sin(ph), radius * math.sin(th) * math.cos(ph), radius * math.cos(th)],
                 dtype=np.float32)

    def nrm(v):
        return v / max(float(np.sqrt((v * v).sum())), 1e-20)
    fwd = nrm(c)
    up = np.array([0, 0, 1], dtype=np.float32)
    right = nrm(np.cross(fwd, up))
    up = nrm(np.cross(right, fwd))
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = np.stack((-right, up, fwd), axis=-1)
    pose[:3, 3] = c
    m = np.linalg.inv(pose)
    R = -np.transpose(m[:3, :3])
    R[:, 0] = -R[:, 0]
    T = -m[:3, 3]
    return R, T


def orbit_camera(radius: float, theta_deg: float, phi_deg: float, fovx: float, H: int, W: int) -> Camera:
    R, T = orbit_pose(radius, theta_deg, phi_deg)
    fovy = focal2fov(fov2focal(fovx, H), W)     # the reference's (quirky) FoVy rule, cam_utils.py:1910
    return Camera.from_RT(R, T, fovx, fovy, H, W)


def look_at_camera(eye, target, fovx: float, H: int, W: int, world_up=(0.0, 0.0, 1.0)) -> Camera:
    """General look-at (used for the indoor bench cameras that look outward from inside the room)."""
    eye = np.asarray(eye, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(world_up, dtype=np.float64))
    r /= np.linalg.norm(r)
    d = np.cross(f, r)                          # camera +y points down in image space
    R = np.stack((r, d, f), axis=-1)            # C2W rotation, columns = camera axes
    T = -R.T @ eye
    fovy = focal2fov(fov2focal(fovx, H), W)
    return Camera.from_RT(R.astype(np.float32), T.astype(np.float32), fovx, fovy, H, W)
