"""The render glue on either side of the rasterizer boundary, restated for callers / tests on the GPU box.

Mirrors what the reference does around `GaussianRasterizer` (nothing here is imported from it):
  * activations of the Gaussian parameters: exp scales, sigmoid opacity, L2-normalised quaternions,
    cat(f_dc, f_rest) SH                                   gs_renderer.py:168-182, 464-488
  * settings construction from an RCamera-like camera       scene_gaussian.py:949-964
  * the zero `means2D` leaf that receives the screen-space gradient   scene_gaussian.py:919-933
  * post-processing depth_alpha -> (disp, alpha)            scene_gaussian.py:1023-1032
  * the returned dict                                       scene_gaussian.py:1036-1044
  * the training-time augmentations of object_render (test=False): active SH degree dropped to 0, background replaced
    by noise or black, multiplicative noise on the SH coefficients and on the scales -- same random decisions in the
    same order from the same generators                      scene_gaussian.py:938-947, 1001-1008
Pinned by tests/golden/object_render.npz and object_render_train.npz (the reference's own object_render executed over the
CPU oracle, test=True and, with seeded generators, test=False).
"""
from __future__ import annotations

import math
import random
from typing import Optional

import torch


class GaussianParams:
    """Raw (pre-activation) parameters as GaussianModel stores them (gs_renderer.py:184-203)."""

    def __init__(self, xyz, log_scales, raw_rotation, logit_opacity, f_dc, f_rest, active_sh_degree: int):
        self._xyz, self._scaling, self._rotation = xyz, log_scales, raw_rotation
        self._opacity, self._features_dc, self._features_rest = logit_opacity, f_dc, f_rest
        self.active_sh_degree = int(active_sh_degree)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    def parameters(self):
        return [self._xyz, self._scaling, self._rotation, self._opacity, self._features_dc, self._features_rest]


def _cam_tensors(cam, device, dtype=torch.float32):
    t = lambda a: torch.as_tensor(a, dtype=dtype, device=device)
    return t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center)


def object_render(params: GaussianParams, camera, bg_color: torch.Tensor, scaling_modifier: float = 1.0,
                  score_flag: bool = False, rasterizer_cls=None, settings_cls=None, test: bool = True,
                  black_video: bool = False, sh_deg_aug_ratio: float = 0.1, bg_aug_ratio: float = 0.3,
                  shs_aug_ratio: float = 1.0, scale_aug_ratio: float = 1.0, rng=random, host_noise: bool = False):
    """SceneGaussian.object_render / score_render. test=True (the default here): no random augmentation. test=False:
    the reference's training-time augmentations, drawing from `rng` (Python's `random`) and torch's generator in the
    reference's order. host_noise: draw the torch noise on the CPU generator and move it to the parameters' device (so
    that a seeded run reproduces the CPU-captured fixture on a GPU)."""
    if rasterizer_cls is None or settings_cls is None:
        from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
        rasterizer_cls = rasterizer_cls or GaussianRasterizer
        settings_cls = settings_cls or GaussianRasterizationSettings
    xyz = params.get_xyz
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass

    def rand_like(t, fn):
        return fn(t.shape, dtype=t.dtype).to(t.device) if host_noise else (torch.rand_like(t) if fn is torch.rand
                                                                           else torch.randn_like(t))
    if black_video:
        bg_color = torch.zeros_like(bg_color)
    act_SH = 0 if (rng.random() < sh_deg_aug_ratio and not test) else params.active_sh_degree     # :938-941
    if rng.random() < bg_aug_ratio and not test:                                                  # :943-947
        bg_color = rand_like(bg_color, torch.rand) if rng.random() < 0.5 else torch.zeros_like(bg_color)
    tanfovx = math.tan(camera.FoVx * 0.5)
    tanfovy = math.tan(camera.FoVy * 0.5)
    vm, pm, cp = _cam_tensors(camera, xyz.device)
    settings = settings_cls(image_height=int(camera.image_height), image_width=int(camera.image_width),
                            tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
                            viewmatrix=vm, projmatrix=pm, sh_degree=act_SH, campos=cp,
                            prefiltered=False, score_flag=score_flag)
    rasterizer = rasterizer_cls(raster_settings=settings)
    scales = params.get_scaling
    shs = params.get_features
    if rng.random() < shs_aug_ratio and not test:                                                 # :1001-1003
        shs = shs + (rand_like(shs, torch.randn) * ((0.2 ** 0.5) * shs))
    if rng.random() < scale_aug_ratio and not test:                                               # :1005-1008
        scales = torch.clamp(scales + (rand_like(scales, torch.randn) * ((0.2 ** 0.5) * scales / 4)), 0.0)
    res = rasterizer(means3D=xyz, means2D=screenspace_points, shs=shs, colors_precomp=None,
                     opacities=params.get_opacity, scales=scales, rotations=params.get_rotation, cov3D_precomp=None)
    score: Optional[torch.Tensor] = None
    if score_flag:
        score, rendered_image, radii, depth_alpha = res
    else:
        rendered_image, radii, depth_alpha = res
    depth, alpha = torch.chunk(depth_alpha, 2)
    focal = 1 / (2 * math.tan(camera.FoVx / 2))
    disp = focal / (depth + (alpha * 10) + 1e-5)
    try:
        min_d = disp[alpha <= 0.1].min()
    except Exception:
        min_d = disp.min()
    disp = torch.clamp((disp - min_d) / (disp.max() - min_d), 0.0, 1.0)
    out = {"image": rendered_image, "depth": disp, "alpha": alpha, "viewspace_points": screenspace_points,
           "visibility_filter": radii > 0, "radii": radii, "scales": scales}
    if score_flag:
        out["important_score"] = score
    return out
