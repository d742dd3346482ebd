"""ORACLE (test infrastructure, NOT product code): ctypes front-end of oracle/gsr_oracle.c.

numpy in / numpy out. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
See gsr_oracle.c for the reference citations; parity of the rasterizer arithmetic is UNPINNED.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgsr_oracle.so")


class OrcView(C.Structure):
    _fields_ = [("P", C.c_int32), ("M", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
                ("bg", C.c_float * 3), ("view", C.c_float * 16), ("proj", C.c_float * 16),
                ("campos", C.c_float * 3), ("prefiltered", C.c_int32), ("score_mode", C.c_int32)]


_SO_OMP = os.path.join(_HERE, "_build", "libgsr_oracle_omp.so")


def build(force: bool = False) -> str:
    """Builds the scalar checker and the all-core timing build (same source, -DORC_OMP -fopenmp)."""
    src = os.path.join(_HERE, "gsr_oracle.c")
    mk = os.path.join(_HERE, "Makefile")
    for so, target in ((_SO, "_build/libgsr_oracle.so"), (_SO_OMP, "_build/libgsr_oracle_omp.so")):
        if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(mk)):
            subprocess.check_call(["make", "-s", "-C", _HERE, "-B", target])
    return _SO


_libs = {}


def lib(omp: bool = False):
    """omp=False: the scalar, deterministic checker. omp=True: the all-core build, for timing the CPU path only."""
    if omp not in _libs:
        build()
        L = C.CDLL(_SO_OMP if omp else _SO)
        L.orc_bin_sort.restype = C.c_uint64
        L.orc_threads.restype = C.c_int
        _libs[omp] = L
    return _libs[omp]


def threads(omp: bool = False) -> int:
    return int(lib(omp).orc_threads())


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def make_view(P, M, D, H, W, tanfovx, tanfovy, bg, view, proj, campos, scale_modifier=1.0, prefiltered=False,
              score_mode=0) -> OrcView:
    v = OrcView()
    v.P, v.M, v.D, v.H, v.W = int(P), int(M), int(D), int(H), int(W)
    v.tanfovx, v.tanfovy, v.scale_modifier = float(tanfovx), float(tanfovy), float(scale_modifier)
    v.bg[:] = [float(x) for x in np.asarray(bg).reshape(3)]
    v.view[:] = [float(x) for x in np.asarray(view, dtype=np.float32).reshape(16)]
    v.proj[:] = [float(x) for x in np.asarray(proj, dtype=np.float32).reshape(16)]
    v.campos[:] = [float(x) for x in np.asarray(campos, dtype=np.float32).reshape(3)]
    v.prefiltered, v.score_mode = int(bool(prefiltered)), int(score_mode)
    return v


def forward(view: OrcView, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
            cov3D_precomp=None, score: bool = False, omp: bool = False) -> dict:
    """Full forward K1-K6. Returns every intermediate (the bit-exact artefacts included)."""
    L = lib(omp)
    P, H, W = view.P, view.H, view.W
    means3D, opacities = _f32(means3D), _f32(opacities)
    shs, colors_precomp, scales, rotations, cov3D_precomp = map(_f32, (shs, colors_precomp, scales, rotations, cov3D_precomp))
    o = dict(depth=np.zeros(P, np.float32), xy=np.zeros((P, 2), np.float32),
             conic_opacity=np.zeros((P, 4), np.float32), rgb=np.zeros((P, 3), np.float32),
             radii=np.zeros(P, np.int32), rect=np.zeros((P, 4), np.int32), tiles_touched=np.zeros(P, np.uint32),
             clamped=np.zeros((P, 3), np.uint8), cov3D=np.zeros((P, 6), np.float32))
    L.orc_preprocess(C.byref(view), _p(means3D), _p(scales), _p(rotations), _p(cov3D_precomp), _p(opacities),
                     _p(shs), _p(colors_precomp), _p(o["depth"]), _p(o["xy"]), _p(o["conic_opacity"]), _p(o["rgb"]),
                     _p(o["radii"]), _p(o["rect"]), _p(o["tiles_touched"]), _p(o["clamped"]), _p(o["cov3D"]))
    N = int(L.orc_bin_sort(P, H, W, _p(o["rect"]), _p(o["depth"]), _p(o["tiles_touched"]), None, None, None))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    o["keys"] = np.zeros(max(N, 1), np.uint64)
    o["point_list"] = np.zeros(max(N, 1), np.uint32)
    o["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    L.orc_bin_sort(P, H, W, _p(o["rect"]), _p(o["depth"]), _p(o["tiles_touched"]), _p(o["keys"]),
                   _p(o["point_list"]), _p(o["ranges"]))
    o["keys"], o["point_list"], o["N"] = o["keys"][:N], o["point_list"][:N], N
    o["image"] = np.zeros((3, H, W), np.float32)
    o["depth_alpha"] = np.zeros((2, H, W), np.float32)
    o["final_T"] = np.zeros((H, W), np.float32)
    o["n_contrib"] = np.zeros((H, W), np.uint32)
    o["important_score"] = np.zeros(P, np.float32) if score else None
    pl = o["point_list"] if N else np.zeros(1, np.uint32)
    L.orc_render_fwd(C.byref(view), _p(o["ranges"]), _p(pl), _p(o["xy"]), _p(o["conic_opacity"]), _p(o["rgb"]),
                     _p(o["depth"]), _p(o["image"]), _p(o["depth_alpha"]), _p(o["final_T"]), _p(o["n_contrib"]),
                     _p(o["important_score"]))
    return o


def backward(view: OrcView, fwd: dict, dL_dimage, dL_ddepth_alpha, means3D, shs=None, scales=None, rotations=None,
             cov3D_precomp=None, cam_grads: bool = False, omp: bool = False) -> dict:
    L = lib(omp)
    P, M = view.P, view.M
    means3D = _f32(means3D)
    shs, scales, rotations, cov3D_precomp = map(_f32, (shs, scales, rotations, cov3D_precomp))
    dL_dimage, dL_ddepth_alpha = _f32(dL_dimage), _f32(dL_ddepth_alpha)
    g = dict(dL_dxy_ndc=np.zeros((P, 2), np.float32), dL_dconic=np.zeros((P, 3), np.float32),
             dL_dopacity=np.zeros((P, 1), np.float32), dL_drgb=np.zeros((P, 3), np.float32),
             dL_ddepth=np.zeros(P, np.float32))
    pl = fwd["point_list"] if fwd["N"] else np.zeros(1, np.uint32)
    L.orc_render_bwd(C.byref(view), _p(fwd["ranges"]), _p(pl), _p(fwd["xy"]), _p(fwd["conic_opacity"]),
                     _p(fwd["rgb"]), _p(fwd["depth"]), _p(fwd["final_T"]), _p(fwd["n_contrib"]), _p(dL_dimage),
                     _p(dL_ddepth_alpha), _p(g["dL_dxy_ndc"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]),
                     _p(g["dL_drgb"]), _p(g["dL_ddepth"]))
    g["dL_dmeans3D"] = np.zeros((P, 3), np.float32)
    g["dL_dmeans2D"] = np.zeros((P, 3), np.float32)
    g["dL_dscales"] = None if cov3D_precomp is not None else np.zeros((P, 3), np.float32)
    g["dL_drotations"] = None if cov3D_precomp is not None else np.zeros((P, 4), np.float32)
    g["dL_dcov3D"] = np.zeros((P, 6), np.float32) if cov3D_precomp is not None else None
    g["dL_dshs"] = np.zeros((P, M, 3), np.float32) if shs is not None else None
    g["dL_dview"] = np.zeros((4, 4), np.float32) if cam_grads else None
    g["dL_dproj"] = np.zeros((4, 4), np.float32) if cam_grads else None
    g["dL_dcampos"] = np.zeros(3, np.float32) if cam_grads else None
    L.orc_preprocess_bwd(C.byref(view), _p(means3D), _p(scales), _p(rotations), _p(cov3D_precomp), _p(shs),
                         _p(fwd["radii"]), _p(fwd["clamped"]), _p(g["dL_dxy_ndc"]), _p(g["dL_dconic"]),
                         _p(g["dL_drgb"]), _p(g["dL_ddepth"]), _p(g["dL_dmeans3D"]), _p(g["dL_dmeans2D"]),
                         _p(g["dL_dscales"]), _p(g["dL_drotations"]), _p(g["dL_dcov3D"]), _p(g["dL_dshs"]),
                         _p(g["dL_dview"]), _p(g["dL_dproj"]), _p(g["dL_dcampos"]))
    g["dL_dcolors"] = g["dL_drgb"] if shs is None else None
    return g


# ---------------------------------------------------------------------------------------------------------------------
# The scalar fp32 oracle behind the reference's module interface (`GaussianRasterizer(raster_settings)(means3D=..., ...)`),
# with autograd: forward = orc_* forward, backward = the oracle's hand-derived backward. Used by tests/golden/make_golden.py
# to run the reference's UNCHANGED object_render / scene_render in fp32 and record what crosses the rasterizer boundary
# (tests/golden/raster_boundary.npz), and by the tests that replay those records. Test infrastructure only.
def _torch():
    import torch
    return torch


class _Recorder(dict):
    pass


def make_rasterizer_module(recorder: Optional[dict] = None, score_mode: int = 0):
    """Returns a class with the reference's constructor signature; every call appends what it saw to `recorder`."""
    torch = _torch()

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, opacities, shs, scales, rotations, s):
            P = int(means3D.shape[0])
            M = int(shs.shape[1])
            v = make_view(P, M, int(s.sh_degree), int(s.image_height), int(s.image_width), float(s.tanfovx), float(s.tanfovy),
                          s.bg.detach().cpu().numpy(), s.viewmatrix.detach().cpu().numpy(), s.projmatrix.detach().cpu().numpy(),
                          s.campos.detach().cpu().numpy(), scale_modifier=float(s.scale_modifier), score_mode=score_mode)
            n = lambda t: t.detach().cpu().numpy().astype(np.float32)
            a = dict(means3D=n(means3D), opacities=n(opacities), shs=n(shs), scales=n(scales), rotations=n(rotations))
            f = forward(v, a["means3D"], a["opacities"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"],
                        score=bool(s.score_flag))
            ctx.v, ctx.f, ctx.a = v, f, a
            ctx.rec = None
            if recorder is not None:
                ctx.rec = dict(inputs=a, image=f["image"].copy(), radii=f["radii"].copy(), depth_alpha=f["depth_alpha"].copy(),
                               settings=dict(image_height=int(s.image_height), image_width=int(s.image_width),
                                             tanfovx=float(s.tanfovx), tanfovy=float(s.tanfovy),
                                             bg=s.bg.detach().cpu().numpy().astype(np.float32),
                                             scale_modifier=float(s.scale_modifier),
                                             viewmatrix=s.viewmatrix.detach().cpu().numpy().astype(np.float32),
                                             projmatrix=s.projmatrix.detach().cpu().numpy().astype(np.float32),
                                             sh_degree=int(s.sh_degree),
                                             campos=s.campos.detach().cpu().numpy().astype(np.float32)))
                recorder.setdefault("calls", []).append(ctx.rec)
            img, da = torch.tensor(f["image"]), torch.tensor(f["depth_alpha"])
            radii = torch.tensor(f["radii"])
            ctx.mark_non_differentiable(radii)
            ctx.opac_shape = tuple(opacities.shape)
            return img, radii, da

        @staticmethod
        def backward(ctx, g_img, _g_radii, g_da):
            torch = _torch()
            H, W = ctx.v.H, ctx.v.W
            gi = np.zeros((3, H, W), np.float32) if g_img is None else g_img.detach().cpu().numpy().astype(np.float32)
            gd = np.zeros((2, H, W), np.float32) if g_da is None else g_da.detach().cpu().numpy().astype(np.float32)
            a = ctx.a
            b = backward(ctx.v, ctx.f, gi, gd, a["means3D"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
            if ctx.rec is not None:
                ctx.rec["upstream"] = dict(dL_dimage=gi, dL_ddepth_alpha=gd)
                ctx.rec["grads"] = {k: np.asarray(b[k]).copy() for k in
                                    ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dshs", "dL_dscales", "dL_drotations")}
            t = lambda k: torch.tensor(np.asarray(b[k]))
            return (t("dL_dmeans3D"), t("dL_dmeans2D"), t("dL_dopacity").reshape(ctx.opac_shape), t("dL_dshs"),
                    t("dL_dscales"), t("dL_drotations"), None)

    class GaussianRasterizer(torch.nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None):
            if shs is None or colors_precomp is not None or scales is None or rotations is None or cov3D_precomp is not None:
                raise Exception("this oracle module covers the trainers' call: shs + scales + rotations")
            if self.raster_settings.score_flag:
                raise Exception("score_flag: use oracle.c_oracle.forward(score=True)")
            return _Fn.apply(means3D, means2D, opacities, shs, scales, rotations, self.raster_settings)

    return GaussianRasterizer
