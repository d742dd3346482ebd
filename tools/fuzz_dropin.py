"""One-off widening of tests/test_side_streams.py / tests/test_dropin_graphs.py on the GPU box: the reference's per-view interface
(one GaussianRasterizer call per view) under random configurations and call patterns, with the internal streams
(RasterContext.side_streams 2 / 4) and / or the captured ring (dropin_graphs) on, against the same calls on the caller's stream
with both off -- outputs bit-equal, gradients the same bits (tests/util.same_bits). Includes views that see nothing (N = 0),
P = 1, single-tile images, forward-only calls between differentiable ones, in-place parameter edits between steps.
usage: python tools/fuzz_dropin.py [n_configs] [first_seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dreamscene_amd import dropin, synth  # noqa: E402
from dreamscene_amd.rasterizer import GaussianRasterizer, RasterContext  # noqa: E402
from tests.util import same_bits, settings_for  # noqa: E402

DEV = torch.device("cuda:0")
NAMES = ("means3D", "shs", "opacities", "scales", "rotations")
TOTALS = dict(calls=0, replays=0, eager=0, no_slot=0)        # the captured ring over the whole run


def config(seed):
    rng = np.random.default_rng(55_000 + seed)
    P = int(rng.choice([1, 3, 64, 200, 1000, 5000, 20_000]))
    K = int(rng.choice([1, 4, 16]))
    D = int(rng.integers(0, int(np.sqrt(K))))
    H, W = int(rng.integers(8, 260)), int(rng.integers(8, 260))
    V = int(rng.integers(1, 7))
    g = synth.g_object(max(P, 64), seed=seed, K=K)
    g = {k: np.ascontiguousarray(v[:P]) for k, v in g.items()}
    g["scales"] = (g["scales"] * float(rng.choice([0.5, 2.0, 8.0]))).astype(np.float32)
    cams = synth.object_cameras(V + 1, H, W, radius=float(rng.choice([2.0, 3.5, 8.0])))[1:]
    blind = [bool(rng.random() < 0.15) for _ in range(V)]       # views that see nothing: the camera's copy of the scene is far away
    sets = [settings_for(c, rng.random(3).astype(np.float32), D, DEV) for c in cams]
    pattern = str(rng.choice(["fb", "ffbb", "one_backward", "mixed"]))
    return dict(P=P, K=K, D=D, H=H, W=W, V=V, g=g, sets=sets, blind=blind, pattern=pattern, steps=int(rng.integers(2, 5)),
                seed=seed)


def run(cfg, ctx):
    """The pattern over cfg['steps'] steps with an in-place parameter edit between the steps -> list of per-step (outs, grads)"""
    P, V, H, W = cfg["P"], cfg["V"], cfg["H"], cfg["W"]
    t = {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in cfg["g"].items()}
    far = torch.tensor([[1.0e4, 0.0, 0.0]], device=DEV)
    ups = [tuple(torch.tensor(x, device=DEV) for x in synth.upstream_grads(H, W, seed=k)) for k in range(V)]
    leaves = [t[k] for k in NAMES]
    record = []
    for step in range(cfg["steps"]):
        outs, m2ds = [], []
        for k, s in enumerate(cfg["sets"]):
            m2d = torch.zeros((P, 3), device=DEV, requires_grad=True)
            xyz = t["means3D"] + far if cfg["blind"][k] else t["means3D"]
            if cfg["pattern"] == "mixed" and k % 2 == 1:
                with torch.no_grad():                # an evaluation render between the training views
                    o = GaussianRasterizer(s, context=ctx)(means3D=xyz, means2D=None, shs=t["shs"], opacities=t["opacities"],
                                                           scales=t["scales"], rotations=t["rotations"])
                outs.append(o); m2ds.append(None)
                continue
            o = GaussianRasterizer(s, context=ctx)(means3D=xyz, means2D=m2d, shs=t["shs"], opacities=t["opacities"],
                                                   scales=t["scales"], rotations=t["rotations"])
            outs.append(o); m2ds.append(m2d)
            if cfg["pattern"] == "fb":
                gr = torch.autograd.grad([o[0], o[2]], leaves + [m2d], list(ups[k]))
                m2ds[-1] = [x.clone() for x in gr]
        grads = []
        live = [k for k in range(V) if m2ds[k] is not None]
        if cfg["pattern"] == "fb":
            grads = [m2ds[k] for k in live]
        elif cfg["pattern"] in ("ffbb", "mixed"):
            for k in reversed(live):
                gr = torch.autograd.grad([outs[k][0], outs[k][2]], leaves + [m2ds[k]], list(ups[k]))
                grads.append([x.clone() for x in gr])
        elif live:
            loss = sum((outs[k][0] * ups[k][0]).sum() + (outs[k][2] * ups[k][1]).sum() for k in live)
            grads = [[x.clone() for x in torch.autograd.grad(loss, leaves + [m2ds[k] for k in live])]]
        record.append(([tuple(x.clone() for x in o) for o in outs], grads))
        with torch.no_grad():                         # the optimizer's in-place update (bumps the version counters)
            t["means3D"].add_(0.001 * (step + 1))
            t["opacities"].mul_(0.99)
    torch.cuda.synchronize()
    return record


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for seed in range(first, first + n):
        cfg = config(seed)
        fails = []
        try:
            dropin.reset()
            ref = run(cfg, RasterContext(side_streams=0, dropin_graphs=False))
            for tag, ctx in (("streams 2", RasterContext(side_streams=2, dropin_graphs=False)),
                             ("streams 4", RasterContext(side_streams=4, dropin_graphs=False)),
                             ("ring", RasterContext(side_streams=0, dropin_graphs=True))):     # (both at once: an error since round 6)
                dropin.reset()
                got = run(cfg, ctx)
                for r_ in dropin.stats().values():
                    for k_ in TOTALS:
                        TOTALS[k_] += r_.get(k_, 0)
                for step, ((o_r, g_r), (o_g, g_g)) in enumerate(zip(ref, got)):
                    for k, (a, b) in enumerate(zip(o_g, o_r)):
                        if not all(torch.equal(x, y) for x, y in zip(a, b)):
                            fails.append(f"{tag}: step {step} view {k} outputs differ")
                    for j, (ga, gb) in enumerate(zip(g_g, g_r)):
                        for i, (x, y) in enumerate(zip(ga, gb)):
                            try:
                                same_bits(x, y, f"{tag}: step {step} backward {j} tensor {i}")
                            except AssertionError as e:
                                fails.append(str(e)[:160])
            dropin.reset()
        except Exception as e:
            fails.append(f"exception {e!r}"[:300])
        if fails:
            bad += 1
            print(f"seed {seed} P={cfg['P']} K={cfg['K']} D={cfg['D']} {cfg['H']}x{cfg['W']} V={cfg['V']} {cfg['pattern']} "
                  f"blind={cfg['blind']}: {len(fails)} failures: " + "; ".join(fails[:3]), flush=True)
    from dreamscene_amd import rasterizer as R
    st = R.side_stream_stats()
    print(f"fuzz_dropin: {n - bad} of {n} configurations clean (seeds {first}..{first + n - 1}); captured ring {TOTALS}; internal "
          f"streams {dict(calls=sum(v['calls'] for v in st.values()), reused_forks=sum(v['reused_forks'] for v in st.values()))}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
