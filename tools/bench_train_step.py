"""Thin caller of tools/train_step.py (the function bench.py's `trainer_step` entry runs): one object-training step shaped like
training/object_trainer.py:293-400, three legs.
usage: python tools/bench_train_step.py [--gaussians 500000] [--res 1024] [--views 4] [--seconds 2] [--init-opacity]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=500000)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--init-opacity", action="store_true")
    ap.add_argument("--legs", default="as_imported,views_fused,raw_leaves")
    a = ap.parse_args()
    from tools import train_step as TS
    res = TS.measure(a.gaussians, a.res, a.res, a.views, 16, 3, init_opacity=a.init_opacity, seconds=a.seconds,
                     legs=tuple(a.legs.split(",")))
    res["workload"] = (f"{a.gaussians} Gaussians, K=16, {a.views} views @{a.res}x{a.res}"
                       f"{', every opacity 0.1' if a.init_opacity else ''}: tools/train_step.py")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
