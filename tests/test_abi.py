"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/gsrast.h declares (no compute
calls: there is no GPU here). Also the host-side error behaviour of the drop-in Python interface."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "gsrast.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z_0-9]+)\s*\(", src)))


def test_library_builds_and_exports_header_symbols(built_lib):
    from dreamscene_amd import _lib
    names = _declared_functions()
    assert "gsr_forward_project" in names and "gsr_backward" in names and len(names) >= 14
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gsrast.h but not exported by libgsrast.so"
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert set(names) == bound, f"ctypes binding and header disagree: {set(names) ^ bound}"


def test_ctypes_signatures_have_the_header_s_parameter_counts():
    """A binding with one argument too few still calls -- with the stream in the wrong slot (it happened: gsr_rowmsg_apply_slices
    gained `touched` in the header and not in _lib.SYMBOLS, and the first call crashed the process)."""
    from dreamscene_amd import _lib
    src = open(os.path.join(ROOT, "include", "gsrast.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decl = {}
    for name, args in re.findall(r"\b(gsr_[a-z_0-9]+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        args = " ".join(args.split())
        decl[name] = 0 if args in ("", "void") else args.count(",") + 1
    for name, _, argtypes in _lib.SYMBOLS:
        assert name in decl, name
        assert len(argtypes) == decl[name], f"{name}: {decl[name]} parameters in gsrast.h, {len(argtypes)} in _lib.SYMBOLS"


def test_library_contains_gfx950_code_object(built_lib):
    from dreamscene_amd import _lib
    out = subprocess.run(["strings", "-n", "6", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out
    for k in ("k_preprocess", "k_render_fwd", "k_render_bwd", "k_radix_scatter", "k_emit_pairs", "k_preprocess_bwd"):
        assert k in out, f"kernel {k} missing from the fat binary"


def test_pure_helpers_without_gpu(built_lib):
    lib = built_lib
    assert lib.gsr_version() == 1
    assert lib.gsr_num_tiles(1024, 1024) == 4096 and lib.gsr_num_tiles(70, 90) == 5 * 6
    assert lib.gsr_num_blocks(1) == 1 and lib.gsr_num_blocks(257) == 2
    assert lib.gsr_sort_scratch_bytes(0, 16) > 0
    assert lib.gsr_sort_scratch_bytes(3_000_000, 4096) >= 3 * 4 * 3_000_000
    assert lib.gsr_project_scratch_bytes(500_000) >= 4 * 4 * 500_000
    assert b"invalid argument" in lib.gsr_strerror(-1)
    # argument validation happens before any HIP call: NULL view -> EINVAL
    assert lib.gsr_forward_project(None, None, None, None, None, None) == -1
    assert lib.gsr_backward(None, None, None, None, None, None, None, None, None) == -1


def test_struct_layouts_match_header(built_lib):
    """ctypes mirrors of the POD structs have the sizes the C compiler gives them."""
    from dreamscene_amd import _lib
    src = r'''
    #include <stdio.h>
    #include "gsrast.h"
    int main(){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(GsrView), sizeof(GsrGaussians), sizeof(GsrGeom),
                       sizeof(GsrBinning), sizeof(GsrImages), sizeof(GsrImageGrads), sizeof(GsrGrads), sizeof(GsrRowRegion),
                       sizeof(GsrRowSet)); return 0; }
    '''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    mine = [ctypes.sizeof(t) for t in (_lib.GsrView, _lib.GsrGaussians, _lib.GsrGeom, _lib.GsrBinning, _lib.GsrImages,
                                       _lib.GsrImageGrads, _lib.GsrGrads, _lib.GsrRowRegion, _lib.GsrRowSet)]
    assert sizes == mine


def _build_c_caller(tmp, with_hip: bool) -> str:
    """tests/c_caller/caller.c: a caller written in plain C against include/gsrast.h (gcc, no Python, no torch)."""
    from dreamscene_amd import _lib
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = os.path.join(tmp, "caller_gpu" if with_hip else "caller")
    cmd = ["gcc", "-O1", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_caller", "caller.c"), "-o", exe, "-L", libdir, "-lgsrast",
           f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    if with_hip:
        cmd[1:1] = ["-DWITH_HIP", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]
        cmd += ["-lamdhip64"]
    subprocess.check_call(cmd)
    return exe


def test_c_only_caller_without_gpu(built_lib, tmp_path):
    """The boundary is usable from plain C: header compiles as C11, the library links, and the entry points that need
    no device (version, error strings, sizes, argument validation) behave as declared."""
    exe = _build_c_caller(str(tmp_path), with_hip=False)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "C_CALLER_NOGPU_OK" in out.stdout, (out.returncode, out.stdout, out.stderr)


@pytest.mark.gpu
def test_c_only_caller_renders_on_the_gpu(built_lib, tmp_path):
    """A C program (HIP runtime C API for memory, libgsrast.so for everything else) renders 256 Gaussians forward and
    backward on the default stream: no Python / torch anywhere in the process."""
    exe = _build_c_caller(str(tmp_path), with_hip=True)
    out = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "C_CALLER_GPU_OK" in out.stdout, (out.returncode, out.stdout, out.stderr)


def test_interface_errors_like_the_reference():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    fields = GaussianRasterizationSettings._fields
    assert fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                      "projmatrix", "sh_degree", "campos", "prefiltered", "score_flag")
    s = GaussianRasterizationSettings(image_height=8, image_width=8, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3),
                                      scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0,
                                      campos=torch.zeros(3), prefiltered=False, score_flag=False)
    r = GaussianRasterizer(raster_settings=s)
    x = torch.zeros(4, 3)
    with pytest.raises(Exception):       # neither shs nor colors
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), scales=x, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception):       # both cov3D and scales
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), shs=torch.zeros(4, 1, 3), scales=x,
          rotations=torch.zeros(4, 4), cov3D_precomp=torch.zeros(4, 6))
    # CPU tensors: the product path refuses loudly (no CPU fallback)
    from dreamscene_amd._lib import GsrError
    with pytest.raises(GsrError):
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), shs=torch.zeros(4, 1, 3), scales=x,
          rotations=torch.zeros(4, 4))


def test_product_path_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under dreamscene_amd/ or diff_gaussian_rasterization/ may import it."""
    for pkg in ("dreamscene_amd", "diff_gaussian_rasterization"):
        for dp, _, fs in os.walk(os.path.join(ROOT, pkg)):
            for f in fs:
                txt = open(os.path.join(dp, f), errors="ignore").read() if f.endswith((".py", ".hip", ".h")) else ""
                if f.endswith(".py"):
                    assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M), f
                    assert "libgsr_oracle" not in txt and "c_oracle" not in txt and "torch_oracle" not in txt, f
                elif txt:
                    assert not re.search(r"#\s*include.*oracle", txt), f
