#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/$1; mkdir -p $O; cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_fuzz.py -m gpu -q -x </dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for cfg in "" "--gaussians 2000000 --res 512"; do
  echo "== $cfg"
  for r in 1 2; do for v in new old; do
    if [ $v = old ]; then export GSR_LIB=$ROOT/dreamscene_amd/libgsrast_prev.so; else unset GSR_LIB; fi
    timeout 300 python bench.py --no-cpu-baseline --no-dropin --sustain-seconds 0 --rotate-seconds 0 $cfg </dev/null > $O/b.json 2>$O/b.err
    python -c "import json; d=json.load(open('$O/b.json')); print('$v $r', d['value'], d['roofline']['stage_us_per_view'])"
  done; done
done
