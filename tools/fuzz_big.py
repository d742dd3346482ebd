"""One-off widening of tests/test_full_size.py on the GPU box: random configurations between C2 and C3 in size (Gaussians,
image shape, SH stride / degree, camera distance / elevation / field of view, opacity state, scale range, object or room)
through tests/test_gpu_parity.py's bars -- radii, tiles touched, pair count, sorted list, keys, ranges, n_contrib and the bits of
final_T identical to the scalar C oracle; images and every gradient entry within 1e-5 x max|ref| of their own tensor (tests/util.py: rel_scale).
usage: python tools/fuzz_big.py [n_configs] [first_seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamscene_amd import build, synth  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402
from tests.test_gpu_parity import _check_forward, _grad_check, _run_hip  # noqa: E402
from tests.util import oracle_view  # noqa: E402


def config(seed):
    rng = np.random.default_rng(91_000 + seed)
    room = rng.random() < 0.2
    K = int(rng.choice([4, 16]))
    D = int(rng.integers(0, int(np.sqrt(K))))
    H, W = int(rng.integers(200, 1100)), int(rng.integers(200, 1100))
    if room:
        per_wall = int(rng.choice([20_000, 60_000, 120_000]))
        g = synth.g_indoor(seed=seed, per_wall=per_wall, K=K)
        cam = synth.indoor_cameras(4, H, W, fovx=float(rng.uniform(0.7, 1.1)))[int(rng.integers(0, 4))]
        what = f"room 5 x {per_wall}"
    else:
        P = int(rng.choice([50_000, 100_000, 200_000, 400_000]))
        init = bool(rng.random() < 0.25)
        g = synth.g_object(P, seed=seed, K=K, init_opacity=init)
        g["scales"] = (g["scales"] * float(rng.choice([0.5, 1.0, 1.0, 2.5]))).astype(np.float32)
        radius, theta, fov = float(rng.uniform(1.5, 6.0)), float(rng.uniform(40.0, 100.0)), float(rng.uniform(0.3, 0.9))
        cam = synth.object_cameras(8, H, W, radius=radius, theta=theta, fovx=fov)[int(rng.integers(0, 8))]
        what = f"object {P}{' init' if init else ''} r={radius:.1f} theta={theta:.0f} fov={fov:.2f}"
    bg = rng.random(3).astype(np.float32)
    return g, cam, bg, K, D, f"{what} K={K} D={D} {H}x{W}"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    build.build()
    CO.build()
    bad = 0
    for seed in range(first, first + n):
        g, cam, bg, K, D, what = config(seed)
        P = g["means3D"].shape[0]
        t0 = time.time()
        try:
            out, _ = _run_hip(g, cam, bg, D)
            v = oracle_view(CO, cam, P, K, D, bg)
            f = CO.forward(v, g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
            _check_forward(out, f, P)
            rep = _grad_check(g, cam, bg, D, CO, seed=seed, tol=1e-5)
            worst = max(rep.items(), key=lambda kv: kv[1][0] / max(1.0, kv[1][1]))
            print(f"seed {seed} {what}: N={out['N']} ok, worst {worst[0]} {worst[1][0] / max(1.0, worst[1][1]):.2e} "
                  f"({time.time() - t0:.1f} s)", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"seed {seed} {what}: FAILED {str(e)[:300]}", flush=True)
        torch.cuda.empty_cache()
    print(f"fuzz_big: {n - bad} of {n} configurations clean (seeds {first}..{first + n - 1})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
