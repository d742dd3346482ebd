import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import settings_for, small_scene
from dreamscene_amd import synth
from dreamscene_amd.graph import CapturedViews
from dreamscene_amd.views import GaussianRasterizerViews
DEV = "cuda:0"
V, P, H, W, K, D = 4, 3000, 112, 144, 16, 3
g, _ = small_scene(P=P, H=H, W=W, K=K, seed=17)
t = {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in g.items()}
leaves = [t[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]
cams = synth.object_cameras(8, H, W, radius=3.0)
gis = [torch.tensor(synth.upstream_grads(H, W, seed=k)[0], device=DEV) for k in range(V)]
gdas = [torch.tensor(synth.upstream_grads(H, W, seed=k)[1], device=DEV) for k in range(V)]
rast = CapturedViews()
for step in range(6):
    MODE = os.environ.get("DBG_MODE", "varyD")
    sets = [settings_for(cams[(step + 2 * k) % 8], [0.1 * step, 0.4, 1.0 - 0.2 * k],
                         (D if (step + k) % 3 else 0) if MODE == "varyD" else D, DEV) for k in range(V)]
    er = GaussianRasterizerViews(sets)
    m2 = torch.zeros((V, P, 3), device=DEV, requires_grad=True)
    outs_e = er(means3D=t["means3D"], means2D=m2, shs=t["shs"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    ge = torch.autograd.grad([x for (img, _, da) in outs_e for x in (img, da)], leaves + [m2], [y for k in range(V) for y in (gis[k], gdas[k])])
    ge = [x.clone() for x in ge]
    m2d = torch.zeros((V, P, 3), device=DEV, requires_grad=True)
    outs = rast(sets, means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    img_ok = all(torch.equal(a[0], b[0]) for a, b in zip(outs, outs_e))
    grads = torch.autograd.grad([x for (img, _, da) in outs for x in (img, da)], leaves + [m2d], [y for k in range(V) for y in (gis[k], gdas[k])])
    torch.cuda.synchronize()
    cap = rast._cap
    msg = [f"step {step} img_ok {img_ok} stats {rast.stats}"]
    for n, a, b in zip(("means3D", "shs", "opac", "scales", "rot", "m2d"), grads, ge):
        msg.append(f"{n}: max|cap| {float(a.abs().max()):.3e} max|ref| {float(b.abs().max()):.3e} err {float((a.reshape(b.shape) - b).abs().max()):.3e}")
    if cap is not None and cap.bwd is not None:
        pp = cap.bwd["_partials"]
        msg.append(f"partials max {float(pp.abs().max()):.3e} finite {bool(torch.isfinite(pp).all())}")
    a, b = grads[0], ge[0]
    bad = ((a - b).abs() > 1e-3).any(1)
    if bool(bad.any()):
        idx = torch.nonzero(bad).reshape(-1)
        msg.append(f"bad rows {idx.numel()} first {idx[:8].tolist()} radii per view {[int((o[1] > 0).sum()) for o in outs]}")
        msg.append(f"radii of first bad in each view {[int(o[1][idx[0]]) for o in outs]}; finite {bool(torch.isfinite(a).all())}")
    print("\n   ".join(msg), flush=True)
