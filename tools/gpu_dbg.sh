mkdir -p gpurun_out/dbg
for v in "" "DBG_EAGER_REPS=2" "DBG_EAGER_REPS=6"; do
  echo "=== variant: $v"
  env $v timeout 100 python tools/debug_graph2.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -8
done
timeout 300 python -m pytest tests/test_graph.py -m gpu -q 2>&1 | tail -5
for i in 1 2 3; do timeout 300 python -m pytest tests/test_graph.py -m gpu -q 2>&1 | tail -1; done
