"""Optimizer step of one GaussianModel (6 groups, gs_renderer.py:615-653): torch.optim.Adam (default and fused=True)
vs dreamscene_amd.optim.FusedAdam; and the densification statistics: trainer indexing ops vs fused into K8 (cost ~0).
usage: python tools/bench_epilogue.py [--gaussians 500000] [--K 16]"""
import argparse, json, os, sys, time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=500000)
    ap.add_argument("--K", type=int, default=16)
    a = ap.parse_args()
    from dreamscene_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    P, K = a.gaussians, a.K
    shapes = [(P, 3), (P, 1, 3), (P, K - 1, 3), (P, 1), (P, 3), (P, 4)]
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-2, 5e-3, 1e-3]
    res = {}
    for name, make in (("torch_adam", lambda g: torch.optim.Adam(g, lr=0.0, eps=1e-15)),
                       ("torch_adam_fused", lambda g: torch.optim.Adam(g, lr=0.0, eps=1e-15, fused=True)),
                       ("gsr_fused_adam", lambda g: FusedAdam(g, lr=0.0, eps=1e-15))):
        ps = [torch.randn(s, device=dev).requires_grad_(True) for s in shapes]
        for p in ps:
            p.grad = torch.randn_like(p) * 1e-3
        try:
            opt = make([{"params": [p], "lr": lr} for p, lr in zip(ps, lrs)])
            res[name + "_us"] = round(timeit(opt.step), 1)
        except Exception as e:      # fused=True may be unavailable in this build
            res[name + "_us"] = f"unavailable: {type(e).__name__}"
    n_el = sum(int(torch.tensor(s).prod()) for s in shapes)
    res["elements"] = n_el
    res["gsr_GBps"] = round(n_el * 28 / (res["gsr_fused_adam_us"] * 1e-6) / 1e9, 1)
    # densification statistics the way the trainer does them
    radii = torch.randint(0, 30, (P,), device=dev, dtype=torch.int32)
    vsp_grad = torch.randn(P, 3, device=dev)
    mr, acc, den = torch.zeros(P, device=dev), torch.zeros(P, 1, device=dev), torch.zeros(P, 1, device=dev)

    def trainer_stats():
        vis = radii > 0
        mr[vis] = torch.max(mr[vis], radii[vis])
        acc[vis] += torch.norm(vsp_grad[vis, :2], dim=-1, keepdim=True)
        den[vis] += 1
    res["trainer_stats_us"] = round(timeit(trainer_stats), 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
