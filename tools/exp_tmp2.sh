cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/kt -o kt -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline "$@" > $ROOT/gpurun_out/kt.log 2>&1
python $ROOT/tools/rocprof_summary.py $ROOT/gpurun_out/kt 2>&1 | head -40
