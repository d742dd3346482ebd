"""call T: seed 277 of tools/fuzz_views.py in detail -- which outputs of the batched call differ from the per-view calls, and how"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_views as F
from dreamscene_amd import multiview, rasterizer as R
from dreamscene_amd.rasterizer import GaussianRasterizer
from dreamscene_amd.views import GaussianRasterizerViews
from dreamscene_amd.graph import CapturedViews

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 277
cfg = F.config(seed)
print({k: cfg[k] for k in ("P", "K", "D", "H", "W", "V", "noisy", "arena", "radius")}, [s.sh_degree for s in cfg["sets"]])
fails = F.run(cfg)
print(len(fails), "failures"); [print("  ", f) for f in fails]
P, K, V = cfg["P"], cfg["K"], cfg["V"]
t = {k: torch.tensor(v, device=F.DEV) for k, v in cfg["g"].items()}
with torch.no_grad():
    ref = [GaussianRasterizer(s)(means3D=t["means3D"], means2D=None, shs=t["shs"], opacities=t["opacities"], scales=t["scales"],
                                 rotations=t["rotations"]) for s in cfg["sets"]]
    for rep in range(2):
        outs = GaussianRasterizerViews(cfg["sets"])(means3D=t["means3D"], means2D=torch.zeros((V, P, 3), device=F.DEV),
                                                    shs=t["shs"], opacities=t["opacities"],
                                                    scales=t["scales"], rotations=t["rotations"])
        for k, (o, r) in enumerate(zip(outs, ref)):
            for name, a, b in zip(("image", "radii", "depth_alpha"), o, r):
                if not torch.equal(a, b):
                    d = (a.double() - b.double()).abs()
                    nz = int((d > 0).sum())
                    idx = torch.nonzero(d > 0)[:5].tolist()
                    print(f"rep {rep} view {k} {name}: {nz} entries differ, max {float(d.max()):.3e}, first at {idx}")
# which of the two is the oracle's? (view 0)
from tests.util import oracle_view
from oracle import c_oracle
from dreamscene_amd import synth
cams = synth.object_cameras(V + 1, cfg["H"], cfg["W"], radius=cfg["radius"])[1:]
g = cfg["g"]
for k in range(V):
    s = cfg["sets"][k]
    v = oracle_view(c_oracle, cams[k], P, K, s.sh_degree, s.bg.cpu().numpy())
    f = c_oracle.forward(v, g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
    keys = [kk for kk in f.keys()] if isinstance(f, dict) else dir(f)
    img_o = np.asarray(f["image"])
    for tag, o in (("per-view", ref[k]), ("batched", outs[k])):
        d = np.abs(o[0].cpu().numpy().astype(np.float64) - img_o.reshape(o[0].shape))
        print(f"view {k} {tag} image vs C oracle: max {d.max():.3e}, entries over 1e-5: {(d > 1e-5).sum()}")
    rad_o = np.asarray(f["radii"] if isinstance(f, dict) else f.radii)
    print(f"view {k} radii equal to the oracle: per-view {np.array_equal(ref[k][1].cpu().numpy(), rad_o)}, batched {np.array_equal(outs[k][1].cpu().numpy(), rad_o)}")
print("done")
