cd $GRAFT_REPO_ROOT
python - <<PY
from dreamscene_amd import build
build.build(force=True, extra_flags=["-DGSR_EXP_TIMELINE"])
PY
python - <<PY
import ctypes, numpy as np, torch, sys
sys.path.insert(0,'.')
from dreamscene_amd import rasterizer as R, synth, _lib
from dreamscene_amd.rasterizer import GaussianRasterizationSettings
lib=_lib.load()
dev=torch.device('cuda:0'); H=W=1024
g=synth.g_object(500000,0,16); cam=synth.object_cameras(1,H,W)[0]; D=3
t=lambda a: torch.tensor(np.asarray(a,dtype=np.float32),device=dev)
s=GaussianRasterizationSettings(H,W,cam.tanfovx,cam.tanfovy,t([1,1,1]),1.0,t(cam.world_view_transform),t(cam.full_proj_transform),D,t(cam.camera_center),False,False)
p={k:t(v) for k,v in g.items()}
for it in range(3):
    o,st=R.rasterize_forward_raw(s,p["means3D"],p["opacities"],p["shs"],None,p["scales"],p["rotations"],None,mode="sync")
torch.cuda.synchronize()
buf=(ctypes.c_ulonglong*(16384*8))()
lib.gsr_debug_read_timeline(buf)
a=np.array(buf,dtype=np.uint64).reshape(16384,8).astype(np.int64)
ln=a[:,2]; act=ln>0
t0=a[:,0].min(); start=(a[:,0]-t0)/100.0; end=(a[:,1]-t0)/100.0; dur=end-start
print("active",act.sum(),"span",end.max())
tot=a[act,3].sum(); print("wave0 cycles total",tot," stage %.1f%% barrier %.1f%% rest(compute) %.1f%%"%(100*a[act,4].sum()/tot,100*a[act,5].sum()/tot,100*(tot-a[act,4].sum()-a[act,5].sum())/tot))
print("steps total (wave0)",a[act,6].sum()," batches total",a[act,7].sum()," cycles per step overall %.0f"%((tot-a[act,4].sum()-a[act,5].sum())/max(a[act,6].sum(),1)))
order=np.argsort(-dur)[:10]
for i in order: print("item",i,"start %.1f dur %.1f len %d cycles %d stage %d bar %d steps %d batches %d  -> cyc/step %.0f"%(start[i],dur[i],ln[i],a[i,3],a[i,4],a[i,5],a[i,6],a[i,7],(a[i,3]-a[i,4]-a[i,5])/max(a[i,6],1)))
PY
