"""Debug aid: replay the raster_boundary cases through the HIP rasterizer and dump its gradients (gpurun_out/<tag>/)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_boundary_fixture import CASES
from dreamscene_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda:0")
out_dir = sys.argv[1]
os.makedirs(out_dir, exist_ok=True)
t = lambda x: torch.tensor(np.asarray(x, np.float32), device=dev)
res = {}
for name, c in CASES.items():
    s, a = c["settings"], c["inputs"]
    st = GaussianRasterizationSettings(image_height=int(s["image_height"]), image_width=int(s["image_width"]), tanfovx=float(s["tanfovx"]),
        tanfovy=float(s["tanfovy"]), bg=t(s["bg"]), scale_modifier=float(s["scale_modifier"]), viewmatrix=t(s["viewmatrix"]),
        projmatrix=t(s["projmatrix"]), sh_degree=int(s["sh_degree"]), campos=t(s["campos"]), prefiltered=False, score_flag=False)
    p = {k: t(v).requires_grad_(True) for k, v in a.items()}
    m2d = torch.zeros_like(p["means3D"], requires_grad=True)
    img, radii, da = GaussianRasterizer(raster_settings=st)(means3D=p["means3D"], means2D=m2d, shs=p["shs"], colors_precomp=None,
        opacities=p["opacities"], scales=p["scales"], rotations=p["rotations"], cov3D_precomp=None)
    torch.autograd.backward([img, da], [t(c["upstream"]["dL_dimage"]), t(c["upstream"]["dL_ddepth_alpha"])])
    got = dict(dL_dmeans3D=p["means3D"].grad, dL_dmeans2D=m2d.grad, dL_dopacity=p["opacities"].grad, dL_dshs=p["shs"].grad,
               dL_dscales=p["scales"].grad, dL_drotations=p["rotations"].grad)
    for k, v in got.items():
        res[f"{name}/{k}"] = v.cpu().numpy()
        ref = c["grads"][k]
        e = np.abs(v.cpu().numpy().reshape(ref.shape).astype(np.float64) - ref)
        print(name, k, "rel %.2e" % (e.max() / max(1.0, np.abs(ref).max())), "argmax", np.unravel_index(e.argmax(), e.shape))
np.savez_compressed(os.path.join(out_dir, "boundary_hip.npz"), **res)
