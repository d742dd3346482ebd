"""ctypes binding of libgsrast.so (the C ABI in include/gsrast.h). Fails loudly when the HIP library is
missing or does not export the declared symbols: there is NO CPU / PyTorch fallback in the product path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSR_LIB: an experimental build of the same library (dreamscene_amd/build.py, A/B measurements); default: the product
LIB_PATH = os.environ.get("GSR_LIB") or os.path.join(_HERE, "libgsrast.so")

STAGES = ["preprocess", "scan", "duplicate", "sort", "ranges", "render_fwd", "render_bwd", "preprocess_bwd"]

_f = C.c_void_p  # device pointers travel as integers


class GsrView(C.Structure):
    _fields_ = [("P", C.c_int32), ("sh_stride", C.c_int32), ("sh_degree", C.c_int32),
                ("image_height", C.c_int32), ("image_width", C.c_int32),
                ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
                ("prefiltered", C.c_int32), ("score_mode", C.c_int32),
                ("bg", _f), ("viewmatrix", _f), ("projmatrix", _f), ("campos", _f), ("dynamic", _f)]


GSR_MAX_MODELS = 16
GSR_PACKED_VIEW_FLOATS = 44
GSR_PARTIAL_WORDS = 32     # 32-bit words per Gaussian of GsrGrads.partials (16 doubles, 12 used)


class GsrModel(C.Structure):
    _fields_ = [("count", C.c_int32), ("reserved_", C.c_int32), ("xyz", _f), ("scaling", _f), ("rotation", _f),
                ("opacity", _f), ("features_dc", _f), ("features_rest", _f)]


class GsrScene(C.Structure):
    _fields_ = [("n_models", C.c_int32), ("reserved_", C.c_int32), ("models", GsrModel * GSR_MAX_MODELS),
                ("scale_noise", _f), ("sh_noise", _f), ("scales_out", _f), ("rotations_out", _f),
                ("opacities_out", _f)]


class GsrModelGrads(C.Structure):
    _fields_ = [("xyz", _f), ("scaling", _f), ("rotation", _f), ("opacity", _f), ("features_dc", _f),
                ("features_rest", _f)]


class GsrSceneGrads(C.Structure):
    _fields_ = [("models", GsrModelGrads * GSR_MAX_MODELS), ("dL_dscales_out", _f)]


class GsrGaussians(C.Structure):
    _fields_ = [("means3D", _f), ("opacities", _f), ("shs", _f), ("colors_precomp", _f), ("scales", _f),
                ("rotations", _f), ("cov3D_precomp", _f), ("scene", C.POINTER(GsrScene))]


class GsrGeom(C.Structure):
    _fields_ = [("splat", _f), ("radii", _f), ("tiles_touched", _f), ("block_offsets", _f), ("scratch", _f),
                ("scratch_bytes", C.c_size_t), ("sorted_idx", _f)]


class GsrBinning(C.Structure):
    _fields_ = [("point_list", _f), ("ranges", _f), ("tile_work", _f), ("bwd_items_cap", C.c_uint32),
                ("seg_len", C.c_uint32), ("keys_sorted", _f), ("scratch", _f), ("scratch_bytes", C.c_size_t),
                ("count_on_device", C.c_int32), ("fwd_mode", C.c_int32), ("stats_host", _f)]


class GsrImages(C.Structure):
    _fields_ = [("color", _f), ("depth_alpha", _f), ("final_T", _f), ("n_contrib", _f), ("tile_depth", _f),
                ("ckpt", _f), ("important_score", _f)]


class GsrImageGrads(C.Structure):
    _fields_ = [("dL_dcolor", _f), ("dL_ddepth_alpha", _f)]


class GsrGrads(C.Structure):
    _fields_ = [("dL_dmeans3D", _f), ("dL_dmeans2D", _f), ("dL_dopacities", _f), ("dL_dshs", _f), ("dL_dcolors", _f),
                ("dL_dscales", _f), ("dL_drotations", _f), ("dL_dcov3D", _f), ("dL_dview", _f), ("dL_dproj", _f),
                ("dL_dcampos", _f), ("partials", _f), ("accumulate", C.c_int32), ("reserved_", C.c_int32),
                ("stat_max_radii2D", _f), ("stat_xyz_gradient_accum", _f), ("stat_denom", _f),
                ("scene", C.POINTER(GsrSceneGrads)), ("reached_mask", C.c_void_p),
                ("reach", C.c_void_p), ("scratch_clean", C.c_int32), ("zero_outside", C.c_int32)]


class GsrAdamGroup(C.Structure):
    _fields_ = [("param", _f), ("grad", _f), ("exp_avg", _f), ("exp_avg_sq", _f), ("numel", C.c_int64),
                ("lr", C.c_float), ("reserved_", C.c_float)]


GSR_MAX_ADAM_GROUPS = 32
GSR_ROWSET_MAX_REGIONS = 8


class GsrRowRegion(C.Structure):
    _fields_ = [("ptr", _f), ("width", C.c_int32), ("stride", C.c_int32)]


class GsrRowSet(C.Structure):
    _fields_ = [("rows", C.c_int32), ("n_regions", C.c_int32), ("regions", GsrRowRegion * GSR_ROWSET_MAX_REGIONS)]


# every symbol include/gsrast.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("gsr_adam_step", C.c_int, [C.POINTER(GsrAdamGroup), C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double,
                                C.c_int32, C.c_void_p]),
    ("gsr_version", C.c_int, []),
    ("gsr_strerror", C.c_char_p, [C.c_int]),
    ("gsr_last_hip_error", C.c_int, []),
    ("gsr_project_scratch_bytes", C.c_size_t, [C.c_int32]),
    ("gsr_sort_scratch_bytes", C.c_size_t, [C.c_uint64, C.c_uint32]),
    ("gsr_num_tiles", C.c_uint32, [C.c_int32, C.c_int32]),
    ("gsr_num_blocks", C.c_uint32, [C.c_int32]),
    ("gsr_profile_create", C.c_void_p, []),
    ("gsr_profile_destroy", None, [C.c_void_p]),
    ("gsr_profile_set_stage_mask", None, [C.c_void_p, C.c_uint32]),
    ("gsr_profile_set_sampling", None, [C.c_void_p, C.c_uint32]),
    ("gsr_profile_collect", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    ("gsr_forward_project", C.c_int, [C.POINTER(GsrView), C.POINTER(GsrGaussians), C.POINTER(GsrGeom),
                                      C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p]),
    ("gsr_forward_project_async", C.c_int, [C.POINTER(GsrView), C.POINTER(GsrGaussians), C.POINTER(GsrGeom),
                                            C.c_void_p, C.c_void_p, C.c_void_p]),
    ("gsr_forward_project_batch", C.c_int, [C.c_int32, C.POINTER(GsrView), C.POINTER(GsrGaussians), C.POINTER(GsrGeom),
                                            C.c_void_p, C.c_void_p, C.c_void_p]),
    ("gsr_pack_views", C.c_int, [C.c_int32, C.POINTER(GsrView), C.c_void_p, C.c_void_p]),
    ("gsr_forward_render", C.c_int, [C.POINTER(GsrView), C.POINTER(GsrGeom), C.c_uint64, C.POINTER(GsrBinning),
                                     C.POINTER(GsrImages), C.c_void_p, C.c_void_p]),
    ("gsr_backward_views", C.c_int, [C.c_int32, C.POINTER(GsrView), C.POINTER(GsrGaussians), C.POINTER(GsrGeom),
                                     C.POINTER(GsrBinning), C.POINTER(GsrImages), C.POINTER(GsrImageGrads),
                                     C.POINTER(GsrGrads), C.c_void_p, C.c_void_p]),
    ("gsr_forward_render_batch", C.c_int, [C.c_int32, C.POINTER(GsrView), C.POINTER(GsrGeom), C.c_uint64,
                                           C.POINTER(GsrBinning), C.POINTER(GsrImages), C.c_void_p, C.c_void_p]),
    ("gsr_rows_scratch_bytes", C.c_size_t, [C.c_int32]),
    ("gsr_rows_pack", C.c_int, [C.POINTER(GsrRowSet), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                C.c_size_t, C.c_void_p]),
    ("gsr_rows_unpack", C.c_int, [C.POINTER(GsrRowSet), C.c_void_p, C.c_void_p, C.c_uint32, C.c_int64, C.c_int32, C.c_void_p,
                                  C.c_void_p]),
    ("gsr_sum_slices", C.c_int, [C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    ("gsr_rowmsg_bytes", C.c_size_t, [C.c_int32, C.c_int32, C.c_uint32]),
    ("gsr_rowmsg_pack", C.c_int, [C.POINTER(GsrRowSet), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    ("gsr_rowmsg_apply", C.c_int, [C.POINTER(GsrRowSet), C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    ("gsr_rowmsg_pack_slices", C.c_int, [C.POINTER(GsrRowSet), C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_uint32,
                                         C.c_void_p]),
    ("gsr_rowmsg_reduce", C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32, C.c_void_p,
                                    C.c_uint32, C.c_void_p]),
    ("gsr_rowmsg_apply_slices", C.c_int, [C.POINTER(GsrRowSet), C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    ("gsr_knn_scratch_bytes", C.c_size_t, [C.c_int32]),
    ("gsr_knn_mean_dist2", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("gsr_backward", C.c_int, [C.POINTER(GsrView), C.POINTER(GsrGaussians), C.POINTER(GsrGeom), C.POINTER(GsrBinning),
                               C.POINTER(GsrImages), C.POINTER(GsrImageGrads), C.POINTER(GsrGrads), C.c_void_p,
                               C.c_void_p]),
]

_lib = None


class GsrError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen the in-tree library and bind every declared entry point. Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: libgsrast.so must share the HIP runtime torch has loaded (its device pointers and streams are
    # only meaningful inside that runtime instance); dlopen resolves libamdhip64 to the already-loaded copy.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise GsrError(f"{LIB_PATH} is missing: build it with `python -m dreamscene_amd.build` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GsrError(f"{LIB_PATH} does not export {name} (declared in include/gsrast.h)") from e
        fn.restype = res
        fn.argtypes = args
    if lib.gsr_version() != 1:
        raise GsrError(f"libgsrast ABI version {lib.gsr_version()} != 1")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        lib = load()
        raise GsrError(f"{what} failed: {lib.gsr_strerror(rc).decode()} (code {rc})")


class Profile:
    """Per-stage HIP-event timing (GsrProfile). Use: p = Profile(); ...pass p.handle...; p.collect()."""

    def __init__(self):
        self.lib = load()
        self.handle = self.lib.gsr_profile_create()
        self.ms = (C.c_double * len(STAGES))()
        self.counts = (C.c_int64 * len(STAGES))()

    def set_stages(self, names=None):
        """Record only the named stages (None = all)."""
        mask = 0xFFFFFFFF if names is None else sum(1 << STAGES.index(n) for n in names)
        self.lib.gsr_profile_set_stage_mask(self.handle, mask)

    def set_sampling(self, every: int = 1):
        """Record one of every `every` occurrences of each stage."""
        self.lib.gsr_profile_set_sampling(self.handle, int(every))

    def collect(self) -> dict:
        check(self.lib.gsr_profile_collect(self.handle, self.ms, self.counts), "gsr_profile_collect")
        return {s: (self.ms[i], self.counts[i]) for i, s in enumerate(STAGES)}

    def reset(self):
        self.collect()
        for i in range(len(STAGES)):
            self.ms[i] = 0.0
            self.counts[i] = 0

    def __del__(self):
        try:
            if self.handle:
                self.lib.gsr_profile_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
