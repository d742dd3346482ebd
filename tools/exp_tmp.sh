cd $GRAFT_REPO_ROOT
for flags in "" "-DGSR_ABL_LDS2X" "-DGSR_ABL_EXP2X"; do
python - <<PY
from dreamscene_amd import build
build.build(force=True, extra_flags="$flags".split())
PY
python - <<PY
import numpy as np, torch, sys
sys.path.insert(0,'.')
from dreamscene_amd import rasterizer as R, synth, _lib
from dreamscene_amd.rasterizer import GaussianRasterizationSettings
dev=torch.device('cuda:0'); H=W=1024
g=synth.g_object(500000,0,16); cam=synth.object_cameras(1,H,W)[0]; D=3
t=lambda a: torch.tensor(np.asarray(a,dtype=np.float32),device=dev)
s=GaussianRasterizationSettings(H,W,cam.tanfovx,cam.tanfovy,t([1,1,1]),1.0,t(cam.world_view_transform),t(cam.full_proj_transform),D,t(cam.camera_center),False,False)
p={k:t(v) for k,v in g.items()}
prof=_lib.Profile(); R.PROFILE=prof
for it in range(12):
    o,st=R.rasterize_forward_raw(s,p["means3D"],p["opacities"],p["shs"],None,p["scales"],p["rotations"],None,want_aux=False)
torch.cuda.synchronize()
r=prof.collect()
print("flags [$flags] render_fwd us:", round(r["render_fwd"][0]/r["render_fwd"][1]*1e3,1))
PY
done
