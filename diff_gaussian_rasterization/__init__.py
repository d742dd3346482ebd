"""Drop-in for the package DreamScene imports (scene_gaussian.py:11-12):

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

Thin re-export of the MI355X-native implementation in dreamscene_amd (HIP kernels behind libgsrast.so)."""
from dreamscene_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
