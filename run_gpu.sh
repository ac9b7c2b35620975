set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -o r01 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 16 --no-cpu-baseline --no-kernel-timer > $GRAFT_REPO_ROOT/gpurun_out/prof_r01_bench.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/prof_r01_bench.log
find gpurun_out/prof_r01 -type f | head -20
