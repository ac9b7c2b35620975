cd $GRAFT_REPO_ROOT
( time timeout 1500 python bench.py ) > gpurun_out/bench_default.log 2>&1
tail -5 gpurun_out/bench_default.log | cut -c1-3000
