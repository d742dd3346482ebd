"""One-off widening of tests/test_score_views.py on the GPU box: random configurations (P, K, D, image shape, 1..20 sphere cameras,
both score weights, chunk sizes of the batched forward) -- views.importance_scores and the per-camera score_flag loop of the
reference (scene_gaussian.py:1063-1079) against the sum of the scalar C oracle's per-camera scores at 1e-5 x max|ref|;
the image of a score_flag call bit-equal to the plain call's.
usage: python tools/fuzz_score.py [n_configs] [first_seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamscene_amd import build, synth, views  # noqa: E402
from dreamscene_amd.rasterizer import GaussianRasterizer, RasterContext  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402
from tests.util import oracle_view, settings_for  # noqa: E402

DEV = torch.device("cuda:0")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    build.build(); CO.build()
    bad = 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(33_000 + seed)
        P = int(rng.choice([1, 64, 500, 3000, 12_000]))
        K = int(rng.choice([1, 4, 16]))
        D = int(rng.integers(0, int(np.sqrt(K))))
        H, W = int(rng.integers(8, 220)), int(rng.integers(8, 220))
        ncam, mode, chunk = int(rng.integers(1, 21)), int(rng.integers(0, 2)), int(rng.choice([1, 3, 16]))
        g = synth.g_object(max(P, 64), seed=seed, K=K)
        g = {k: np.ascontiguousarray(v[:P]) for k, v in g.items()}
        g["scales"] = (g["scales"] * float(rng.choice([0.5, 2.0, 8.0]))).astype(np.float32)
        cams = synth.sphere_cameras(ncam, H, W, radius=float(rng.choice([2.0, 3.5, 5.35])))
        bg = np.ones(3, np.float32)
        sl = [settings_for(c, bg, D, DEV, score_flag=True) for c in cams]
        gd = {k: torch.tensor(v, device=DEV) for k, v in g.items()}
        args = dict(means3D=gd["means3D"], opacities=gd["opacities"], shs=gd["shs"], scales=gd["scales"], rotations=gd["rotations"])
        rc = RasterContext(score_mode=mode)
        fails = []
        try:
            ref = np.zeros(P, np.float64)
            for c in cams:
                v = oracle_view(CO, c, P, K, D, bg, score_mode=mode)
                ref += CO.forward(v, g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"],
                                  score=True)["important_score"].astype(np.float64)
            scale = max(1e-6, float(np.abs(ref).max()))
            for rep in range(2):
                tot = views.importance_scores(sl, context=rc, chunk=chunk, **args).cpu().numpy().astype(np.float64)
                if not np.abs(tot - ref).max() <= 1e-5 * scale:
                    fails.append(f"importance_scores rep {rep}: {np.abs(tot - ref).max() / scale:.2e}")
            loop = torch.zeros(P, device=DEV)
            with torch.no_grad():
                for s in sl:
                    sc, img, radii, da = GaussianRasterizer(raster_settings=s, context=rc)(means2D=None, **args)
                    loop += sc
                    img2, radii2, da2 = GaussianRasterizer(raster_settings=s._replace(score_flag=False), context=rc)(means2D=None, **args)
                    if not (torch.equal(img, img2) and torch.equal(radii, radii2) and torch.equal(da, da2)):
                        fails.append("score_flag changes the image")
            if not np.abs(loop.cpu().numpy() - ref).max() <= 1e-5 * scale:
                fails.append(f"per-camera loop: {np.abs(loop.cpu().numpy() - ref).max() / scale:.2e}")
        except Exception as e:
            fails.append(f"exception {e!r}"[:300])
        if fails:
            bad += 1
            print(f"seed {seed} P={P} K={K} D={D} {H}x{W} cams={ncam} mode={mode} chunk={chunk}: " + "; ".join(fails[:4]), flush=True)
    print(f"fuzz_score: {n - bad} of {n} configurations clean (seeds {first}..{first + n - 1})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
