"""Timing experiment: where does a workgroup of K8 spend its time? One 4-view step at C3 with a library built with
-DGSR_K8_STAMPS; the realtime stamps come back in view 0's dL_dmeans2D (the results of that tensor are destroyed).
    python -c "from dreamscene_amd import build as B; B.build(extra_flags=['-DGSR_K8_STAMPS'], \
               out='dreamscene_amd/libgsrast_stamps.so', objdir='dreamscene_amd/_obj_stamps')"
    GSR_LIB=$PWD/dreamscene_amd/libgsrast_stamps.so python tools/k8_stamps.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dreamscene_amd import synth, rasterizer as R
from dreamscene_amd.views import GaussianRasterizerViews
from dreamscene_amd.rasterizer import GaussianRasterizationSettings
dev = torch.device("cuda:0")
P, H, W, K, D, V = 500_000, 1024, 1024, 16, 3, 4
g = synth.g_object(P, seed=0, K=K)
t = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in g.items()}
f = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
cams = synth.object_cameras(V, H, W)
sets = [GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=f([1, 1, 1]),
                                      scale_modifier=1.0, viewmatrix=f(c.world_view_transform), projmatrix=f(c.full_proj_transform),
                                      sh_degree=D, campos=f(c.camera_center), prefiltered=False, score_flag=False) for c in cams]
gi = [f(synth.upstream_grads(H, W, k)[0]) for k in range(V)]
gd = [f(synth.upstream_grads(H, W, k)[1]) for k in range(V)]
rast = GaussianRasterizerViews(sets)
for rep in range(4):
    m2d = torch.zeros((V, P, 3), device=dev, requires_grad=True)
    outs = rast(means3D=t["means3D"], means2D=m2d, shs=t["shs"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    torch.autograd.backward([x for (img, _, da) in outs for x in (img, da)], [y for k in range(V) for y in (gi[k], gd[k])])
    torch.cuda.synchronize()
nwg = (P + 1023) // 1024
st = m2d.grad[0].reshape(-1)[: nwg * 192].reshape(-1, 192)[:, :16].cpu().numpy()   # chunk 0 of workgroup b = Gaussians [64 b, 64 b + 64)
n = int(st[0, 15])
print("workgroups", st.shape[0], "stamps", n, "reached per workgroup: mean %.0f max %.0f" % (st[:, 14].mean(), st[:, 14].max()))
names = ["entry", "classified", "params+SH in, view 0 requested", "zero fill issued + barrier"] + [f"view {k} done" for k in range(n - 5)] + ["rows stored (issued)", "all memory ops complete"]
for k in range(n + 1):
    col = st[:, k] * 0.01
    print(f"{names[k] if k < len(names) else k:40s} mean {col.mean():7.2f} us   p10 {np.percentile(col, 10):7.2f}  p90 {np.percentile(col, 90):7.2f}")
