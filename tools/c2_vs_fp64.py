"""Exploration: HIP at C2 (100 k @512^2) against the INDEPENDENT float64 autograd oracle (libm exp, vectorised torch).
Prints how many pixels take a hard gate differently and the gradient errors. usage: python tools/c2_vs_fp64.py [P] [res]"""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamscene_amd import rasterizer as R, synth
from tests.test_oracle_consistency import _torch_run
from tests.util import settings_for
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
res = int(sys.argv[2]) if len(sys.argv) > 2 else 512
K, D = 16, 3
g = synth.g_object(P, seed=0, K=K)
cam = synth.object_cameras(1, res, res)[0]
bg = np.ones(3, np.float32)
gi, gda = synth.upstream_grads(res, res, 0)
t0 = time.time()
torch.set_num_threads(64)
r = _torch_run(g, cam, bg, D, gi=gi, gda=gda, cam_grad=False)
print("torch fp64 oracle: %.1f s" % (time.time() - t0))
dev = "cuda:0"
t = {k: torch.tensor(v, device=dev) for k, v in g.items()}
s = settings_for(cam, bg, D, dev)
out, st = R.rasterize_forward_raw(s, t["means3D"], t["opacities"], t["shs"], None, t["scales"], t["rotations"], None)
o = R.rasterize_backward_raw(st, torch.tensor(gi, device=dev), torch.tensor(gda, device=dev))
torch.cuda.synchronize()
nc = out["n_contrib"].cpu().numpy().view(np.uint32)
rep = {"radii_equal": bool(np.array_equal(out["radii"].cpu().numpy(), r["radii"])),
       "n_contrib_differs_at_pixels": int((nc != r["aux"]["n_contrib"]).sum()), "pixels": int(nc.size)}
d_img = np.abs(out["color"].cpu().numpy() - r["img"]).max(axis=0)
rep["image_max"] = float(d_img.max()); rep["image_pixels_over_1e-5"] = int((d_img > 1e-5).sum())
for tk, hk in (("means3D", "dL_dmeans3D"), ("scales", "dL_dscales"), ("rotations", "dL_drotations"), ("opacities", "dL_dopacities"),
               ("shs", "dL_dshs"), ("means2D", "dL_dmeans2D")):
    ref = np.asarray(r["grads"][tk], dtype=np.float64)
    got = o[hk].cpu().numpy().astype(np.float64).reshape(ref.shape)
    e = np.abs(got - ref); sc = max(1.0, float(np.abs(ref).max()))
    rep[hk] = {"max_over_scale": float(e.max() / sc), "entries_over_1e-5": int((e > 1e-5 * sc).sum()), "scale": sc}
print(json.dumps(rep, indent=1))
