"""Builds libgsrast.so (hand-written HIP for gfx950) in-tree with hipcc. No torch extension machinery: the
library is a plain C-ABI shared object (include/gsrast.h) loaded through ctypes (dreamscene_amd/_lib.py)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libgsrast.so")
ARCH = "gfx950"

# per-file extra flags: the files whose fp32 results feed integer artefacts are built without FMA contraction
SOURCES = {
    "api.hip": [],
    "preprocess.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
    "binning.hip": ["-ffp-contract=off"],
    "render.hip": ["-fno-slp-vectorize"],
    "knn.hip": ["-ffp-contract=off"],
    "optim.hip": ["-ffp-contract=off"],
    "exchange.hip": [],
}
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-munsafe-fp-atomics", "-fno-gpu-rdc",
          "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libgsrast.so)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = OUT, objdir: str = None) -> str:
    """out / objdir: an experimental variant (other flags) next to the product library, e.g.
    build(extra_flags=["-DGSR_XCD_MAP=0"], out=".../libgsrast_noxcd.so", objdir=".../_obj_noxcd"); GSR_LIB=<path> makes
    dreamscene_amd._lib load it instead (A/B measurements only)."""
    cc = hipcc()
    objdir = objdir or os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "gsr_common.h"), os.path.join(CSRC, "radix_sort.h"),
               os.path.join(ROOT, "include", "gsrast.h"), os.path.abspath(__file__)]
    jobs = []
    for src, flags in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append([cc, "-x", "hip", "-c", s, "-o", o] + COMMON + flags + list(extra_flags))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr:
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(out, objs):
        run([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out] + objs)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
