"""GPU diagnostic: for a fuzz seed, the HIP gradients vs the scalar C oracle vs float64 autograd (which of the two fp32
implementations is closer to the ground truth when they disagree on an ill-conditioned configuration)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_fuzz import _random_config
from tests.test_gpu_parity import _run_hip
from tests.test_oracle_consistency import _torch_run
from tests.util import oracle_view, err
from oracle import c_oracle as CO
from dreamscene_amd import rasterizer as R, synth
for seed in [int(a) for a in sys.argv[1:]]:
    g, cam, bg, P, K, D, deg = _random_config(seed)
    H, W = cam.image_height, cam.image_width
    gi, gda = synth.upstream_grads(H, W, seed)
    out, st = _run_hip(g, cam, bg, D, want_keys=False)
    o = R.rasterize_backward_raw(st, torch.tensor(gi, device="cuda:0"), torch.tensor(gda, device="cuda:0"))
    torch.cuda.synchronize()
    r = _torch_run(g, cam, bg, D, gi=gi, gda=gda)
    v = oracle_view(CO, cam, P, K, D, bg)
    f = CO.forward(v, g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
    b = CO.backward(v, f, gi, gda, g["means3D"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
    print(f"seed {seed} degenerate={deg} P={P} {W}x{H}")
    for tk, ck, hk in [("means3D", "dL_dmeans3D", "dL_dmeans3D"), ("scales", "dL_dscales", "dL_dscales"),
                       ("rotations", "dL_drotations", "dL_drotations"), ("opacities", "dL_dopacity", "dL_dopacities"),
                       ("shs", "dL_dshs", "dL_dshs"), ("means2D", "dL_dmeans2D", "dL_dmeans2D")]:
        a64, c32, h32 = r["grads"][tk], np.asarray(b[ck]), o[hk].cpu().numpy().reshape(np.asarray(b[ck]).shape)
        print(f"   {tk:10s} scale {np.abs(a64).max():.3e}  HIP-fp64 {err(h32, a64):.3e}  C-fp64 {err(c32, a64):.3e}  HIP-C {err(h32, c32):.3e}")
