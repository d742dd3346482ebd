cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/kt -o kt -- python $ROOT/tools/bench_scene.py --scene object --models 4 --per-model 125000 --K 16 --views 20 > $ROOT/gpurun_out/kt.log 2>&1
python $ROOT/tools/rocprof_summary.py $ROOT/gpurun_out/kt 2>&1 | head -45
