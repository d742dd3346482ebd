"""Workload statistics of one view (GPU): per-tile list lengths and traversal depths. Guides kernel tuning."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dreamscene_amd import rasterizer as R, synth
from dreamscene_amd.rasterizer import GaussianRasterizationSettings

def main(scene="object", P=500_000, res=1024):
    dev = torch.device("cuda:0")
    H = W = res
    if scene == "object":
        g = synth.g_object(P, 0, 16); cam = synth.object_cameras(1, H, W)[0]; D = 3
    else:
        g = synth.g_indoor(0, P // 5, 4); cam = synth.indoor_cameras(1, H, W)[0]; D = 1
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
    s = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, t([1, 1, 1]), 1.0, t(cam.world_view_transform),
                                      t(cam.full_proj_transform), D, t(cam.camera_center), False, False)
    p = {k: t(v) for k, v in g.items()}
    o, st = R.rasterize_forward_raw(s, p["means3D"], p["opacities"], p["shs"], None, p["scales"], p["rotations"], None)
    rng = o["ranges"].cpu().numpy().astype(np.int64)
    ln = rng[:, 1] - rng[:, 0]
    nc = o["n_contrib"].cpu().numpy().astype(np.int64)
    gx = W // 16
    tmax = nc.reshape(H // 16, 16, gx, 16).max(axis=(1, 3)).reshape(-1)
    fT = o["final_T"].cpu().numpy()
    done_pix = fT.reshape(H // 16, 16, gx, 16).transpose(0, 2, 1, 3).reshape(-1, 256)
    radii = o["radii"].cpu().numpy()
    print(f"scene={scene} P={g['means3D'].shape[0]} res={res} N={o['N']} visible={(radii>0).sum()} mean radius={radii[radii>0].mean():.1f}")
    print(f"tiles active={(ln>0).sum()} / {ln.size}; list len: mean(active)={ln[ln>0].mean():.0f} max={ln.max()} sum={ln.sum()}")
    print(f"bwd depth (tile max n_contrib): mean(active)={tmax[ln>0].mean():.0f} max={tmax.max()} sum={tmax.sum()}")
    print(f"per-pixel n_contrib: mean over covered px={nc[nc>0].mean():.0f} max={nc.max()}; covered px={(nc>0).sum()}")
    # forward traversal depth: tiles where some pixel never saturates traverse the whole list
    unsat = (done_pix.min(axis=1) >= 0) & ((done_pix > 1e-4 * 1.0).any(axis=1))
    # a pixel is 'done' only when T would drop below 1e-4; approximate: final_T < 1e-3 means it stopped early
    full = ((done_pix > 2e-4).any(axis=1)) & (ln > 0)
    print(f"tiles that traverse their whole list in fwd (some pixel never saturates): {full.sum()}, their list len sum={ln[full].sum()} max={ln[full].max() if full.any() else 0}")
    hist = np.histogram(ln[ln > 0], bins=[1, 64, 256, 1024, 2048, 4096, 8192, 16384, 1 << 20])
    print("list len hist", hist)
    hist = np.histogram(tmax[ln > 0], bins=[0, 1, 64, 256, 1024, 2048, 4096, 8192, 16384, 1 << 20])
    print("bwd depth hist", hist)

if __name__ == "__main__":
    main(*(sys.argv[1:2]), *[int(a) for a in sys.argv[2:]])
