#!/bin/bash
# SURVEY.md 8(d) sweep on one GPU: {512, 800, 1024}^2 x {100 k, 500 k, 1 M, 2 M} G-object, 4 views per step, fwd+bwd views/s with
# the roofline entry of the dominant kernel per cell -> gpurun_out/<tag>_sweep.jsonl (one bench line per cell)
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_sweep.jsonl
mkdir -p gpurun_out; : > $OUT
for P in 100000 500000 1000000 2000000; do
  for R in 512 800 1024; do
    timeout 240 python bench.py --no-cpu-baseline --steps 60 --warmup 10 --rotate-seconds 0 --train-seconds 0 --sustain-seconds 1 --gaussians $P --res $R 2>/dev/null | tail -1 >> $OUT
    tail -1 $OUT | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['config']['gaussians'], d['config']['resolution'][0], d['value'], 'dropin', d['dropin_views_per_s'], d['config']['batched_through'][:22], r['kernel'], r['frac'], (r.get('valu') or {}).get('issue_frac'))"
  done
done
