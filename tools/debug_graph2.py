import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import settings_for, small_scene, tol_ok
from tests.test_graph import _setup, _eager
from dreamscene_amd import synth, rasterizer as R
from dreamscene_amd.graph import CapturedViews
DEV = "cuda:0"
V, K, D = 4, 16, 3
P, H, W = 3000, 112, 144
g, t = _setup(P, H, W, K)
leaves = [t[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]
cams = synth.object_cameras(8, H, W, radius=3.0)
gis = [torch.tensor(synth.upstream_grads(H, W, seed=k)[0], device=DEV) for k in range(V)]
gdas = [torch.tensor(synth.upstream_grads(H, W, seed=k)[1], device=DEV) for k in range(V)]
rast = CapturedViews()
NE = int(os.environ.get("DBG_EAGER_REPS", "2"))
for step in range(5):
    sets = []
    for k in range(V):
        c = cams[(step + 2 * k) % 8]
        s = settings_for(c, [0.1 * step, 0.4, 1.0 - 0.2 * k], D if (step + k) % 3 else 0, DEV)
        if step == 4:
            s = s._replace(tanfovx=s.tanfovx * 1.25, tanfovy=s.tanfovy * 1.25)
        sets.append(s)
    for _ in range(max(1, NE // 2)):
        ref_outs, ref_grads = _eager(sets, t, gis, gdas)
    m2d = torch.zeros((V, P, 3), device=DEV, requires_grad=True)
    outs = rast(sets, means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], scales=t["scales"],
                rotations=t["rotations"])
    grads = torch.autograd.grad([x for (img, _, da) in outs for x in (img, da)], leaves + [m2d],
                                [y for k in range(V) for y in (gis[k], gdas[k])])
    torch.cuda.synchronize()
    ok_img = all(torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) for a, b in zip(outs, ref_outs))
    errs = [float((a.reshape(b.shape) - b).abs().max()) for a, b in zip(grads, ref_grads)]
    print(f"step {step} img_ok {ok_img} errs {['%.2e' % e for e in errs]} stats {rast.stats}", flush=True)
    cap = rast._cap
    if cap is not None and cap.bwd is not None and max(errs) > 1e-3:
        o = cap.bwd
        print("  ptr grads[0]", grads[0].data_ptr(), "o.means3D", o["dL_dmeans3D"].data_ptr(), "same obj", grads[0] is o["dL_dmeans3D"])
        print("  partials max", float(o["_partials"].abs().max()), "g_color max", float(cap.g_color.abs().max()),
              "gis max", float(gis[0].abs().max()))
        # replay once more and look again
        cap.gC.replay(); torch.cuda.synchronize()
        errs2 = [float((o[n].reshape(b.shape) - b).abs().max()) for n, b in zip(("dL_dmeans3D", "dL_dshs", "dL_dopacities", "dL_dscales", "dL_drotations"), ref_grads)]
        print("  after another replay:", ['%.2e' % e for e in errs2])
        # eager backward on the captured states
        e = R.rasterize_backward_views_raw(cap.states, list(cap.g_color), list(cap.g_da))
        torch.cuda.synchronize()
        errs3 = [float((e[n].reshape(b.shape) - b).abs().max()) for n, b in zip(("dL_dmeans3D", "dL_dshs", "dL_dopacities", "dL_dscales", "dL_drotations"), ref_grads)]
        print("  eager backward on the captured states:", ['%.2e' % x for x in errs3])
        break
