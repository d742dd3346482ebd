import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests import test_golden as TG
dev = torch.device("cuda:0")
for seed in (31, 7, 43, 1):
    d, ref, p, out, rec = TG._train_case(seed, dev, host_noise=True)
    errs = {}
    for k, attr in TG.TRAIN_KEYS.items():
        got = out["viewspace_points"].grad if attr is None else getattr(p, attr).grad
        scale = max(1.0, float(np.abs(ref[k]).max()))
        errs[k] = float(np.abs(got.cpu().numpy() - ref[k]).max() / scale)
    print("train", seed, {k: f"{v:.2e}" for k, v in errs.items()})
from dreamscene_amd import render_api
from dreamscene_amd.render_api import GaussianParams
d = TG.load("object_render.npz")
t = lambda k: torch.tensor(d[k], dtype=torch.float32, device=dev, requires_grad=True)
p = GaussianParams(t("xyz"), t("log_scales"), t("raw_rot"), t("logit_opacity"), t("f_dc"), t("f_rest"), int(d["active_sh_degree"]))
cam = TG._cam_from_fixture(d)
out = render_api.object_render(p, cam, torch.tensor(d["bg"], device=dev))
g = lambda k: torch.tensor(d[k], device=dev)
loss = (out["image"] * g("gi")).sum() + (out["depth"] * g("gd")).sum() + (out["alpha"] * g("ga")).sum()
loss.backward()
ref = dict(vsp_grad=out["viewspace_points"].grad, g_xyz=p._xyz.grad, g_scaling=p._scaling.grad, g_rotation=p._rotation.grad,
           g_opacity=p._opacity.grad, g_f_dc=p._features_dc.grad, g_f_rest=p._features_rest.grad)
print("plumbing", {k: f"{float(np.abs(gr.cpu().numpy() - d[k]).max() / max(1.0, float(np.abs(d[k]).max()))):.2e}" for k, gr in ref.items()})
print("depth err", float(np.abs(out["depth"].detach().cpu().numpy() - d["depth"]).max()))
