"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the glue the fused multi-model path replaces.

What `SceneGaussian.scene_render` does before it calls the rasterizer (scene_gaussian.py:753-852), in plain torch on the
CPU and differentiable, so that autograd through it defines the gradients of the RAW leaf tensors:
  * activations per model: exp(_scaling), normalize(_rotation), sigmoid(_opacity), cat(_features_dc, _features_rest)
    (gs_renderer.py:464-488)
  * torch.cat over the models in list order (:753-843)
  * SH noise  shs <- shs + n * (0.2**0.5 * shs)                      (:844-847)
  * scale noise scales <- clamp(scales + n * (0.2**0.5 * scales / 4), 0) (:849-852)
The noise samples n are inputs here (the reference draws them with randn_like at that point).
Pinned by tests/golden/scene_render.npz (the reference's own scene_render, test=True, run over the CPU oracle).
"""
from __future__ import annotations

import math

import torch

LEAVES = ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")


def activate_and_cat(models, scale_noise=None, sh_noise=None):
    """models: list of 6-tuples of torch tensors (raw leaves). Returns dict of concatenated activated tensors."""
    xyz = torch.cat([m[0] for m in models])
    scales = torch.cat([torch.exp(m[1]) for m in models])
    rots = torch.cat([torch.nn.functional.normalize(m[2]) for m in models])
    opac = torch.cat([torch.sigmoid(m[3]) for m in models])
    shs = torch.cat([torch.cat((m[4], m[5]), dim=1) for m in models])
    if sh_noise is not None:
        variance = (0.2 ** 0.5) * shs
        shs = shs + (sh_noise * variance)
    if scale_noise is not None:
        variance = (0.2 ** 0.5) * scales / 4
        scales = torch.clamp(scales + (scale_noise * variance), 0.0)
    return dict(means3D=xyz, scales=scales, rotations=rots, opacities=opac, shs=shs)


def scene_render(models, camera, bg_color, active_sh_degree, rasterizer_cls, settings_cls, scale_noise=None,
                 sh_noise=None, scaling_modifier=1.0):
    """test=True path of scene_render over a given (oracle) rasterizer class; same output dict."""
    a = activate_and_cat(models, scale_noise, sh_noise)
    xyz = a["means3D"]
    screenspace_points = torch.zeros_like(xyz, requires_grad=True) + 0
    screenspace_points.retain_grad()
    t = lambda v: torch.as_tensor(v, dtype=torch.float32)
    settings = settings_cls(image_height=int(camera.image_height), image_width=int(camera.image_width),
                            tanfovx=math.tan(camera.FoVx * 0.5), tanfovy=math.tan(camera.FoVy * 0.5), bg=bg_color,
                            scale_modifier=scaling_modifier, viewmatrix=t(camera.world_view_transform),
                            projmatrix=t(camera.full_proj_transform), sh_degree=active_sh_degree,
                            campos=t(camera.camera_center), prefiltered=False, score_flag=False)
    rasterizer = rasterizer_cls(raster_settings=settings)
    rendered_image, radii, depth_alpha = rasterizer(means3D=xyz, means2D=screenspace_points, shs=a["shs"],
                                                    colors_precomp=None, opacities=a["opacities"], scales=a["scales"],
                                                    rotations=a["rotations"], cov3D_precomp=None)
    depth, alpha = torch.chunk(depth_alpha, 2)
    focal = 1 / (2 * math.tan(camera.FoVx / 2))
    disp = focal / (depth + (alpha * 10) + 1e-5)
    try:
        min_d = disp[alpha <= 0.1].min()
    except Exception:
        min_d = disp.min()
    disp = torch.clamp((disp - min_d) / (disp.max() - min_d), 0.0, 1.0)
    return {"image": rendered_image, "depth": disp, "alpha": alpha, "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0, "radii": radii, "scales": a["scales"]}
