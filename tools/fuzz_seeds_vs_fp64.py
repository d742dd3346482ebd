"""Who is right on the fuzz seeds where the HIP path and the scalar fp32 C oracle differ by more than 1e-5 of a tensor's own scale?
For the given seeds of tests/test_fuzz.py: the HIP gradients, the C oracle's and the float64 autograd oracle's
(oracle/torch_oracle.py) -- per tensor, max |HIP - fp64| and max |C - fp64| relative to max |fp64|. (Float64 decides the gates a
little differently on a handful of (pixel, splat) pairs; a seed where it does is reported as such.)
usage: python tools/fuzz_seeds_vs_fp64.py 107 196 243 260 337 381"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamscene_amd import rasterizer as R, synth  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402
from tests.test_fuzz import _random_config  # noqa: E402
from tests.test_gpu_parity import _run_hip  # noqa: E402
from tests.test_oracle_consistency import _torch_run  # noqa: E402
from tests.util import oracle_view  # noqa: E402

CO.build()
DEV = "cuda:0"
PAIRS = (("means3D", "dL_dmeans3D", "dL_dmeans3D"), ("scales", "dL_dscales", "dL_dscales"), ("rotations", "dL_drotations", "dL_drotations"),
         ("opacities", "dL_dopacities", "dL_dopacity"), ("shs", "dL_dshs", "dL_dshs"), ("means2D", "dL_dmeans2D", "dL_dmeans2D"),
         ("view", "dL_dview", "dL_dview"), ("proj", "dL_dproj", "dL_dproj"), ("campos", "dL_dcampos", "dL_dcampos"))
for seed in [int(a) for a in sys.argv[1:]]:
    g, cam, bg, P, K, D = _random_config(seed)
    H, W = cam.image_height, cam.image_width
    gi, gda = synth.upstream_grads(H, W, seed)
    out, st = _run_hip(g, cam, bg, D, want_keys=False)
    o = R.rasterize_backward_raw(st, torch.tensor(gi, device=DEV), torch.tensor(gda, device=DEV), cam_grads=True)
    torch.cuda.synchronize()
    v = oracle_view(CO, cam, P, K, D, bg)
    f = CO.forward(v, g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
    b = CO.backward(v, f, gi, gda, g["means3D"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"], cam_grads=True)
    r = _torch_run(g, cam, bg, D, gi=gi, gda=gda)
    gates = int((f["n_contrib"] != r["aux"]["n_contrib"]).sum())
    print(f"seed {seed}: P={P} K={K} D={D} {W}x{H}; pixels whose last contributor differs between fp32 and float64: {gates}")
    for tk, hk, ck in PAIRS:
        ref = np.asarray(r["grads"][tk], dtype=np.float64).reshape(-1)
        if o.get(hk) is None or b.get(ck) is None:
            continue
        h = o[hk].cpu().numpy().astype(np.float64).reshape(-1)
        c = np.asarray(b[ck], dtype=np.float64).reshape(-1)
        if tk in ("view", "proj"):          # (4 x 4 with an unused row / column in one of the layouts: compare what both hold)
            ref = ref.reshape(4, 4).reshape(-1); h = h.reshape(-1)[:16]; c = c.reshape(-1)[:16]
        m = max(1e-300, float(np.abs(ref).max()))
        print(f"   {hk:14s} max|fp64| {m:9.3e}   HIP-fp64 {np.abs(h - ref).max() / m:8.2e}   C-fp64 {np.abs(c - ref).max() / m:8.2e}   HIP-C {np.abs(h - c).max() / m:8.2e}")
