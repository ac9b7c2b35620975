cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "step" 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline --no-kernel-timer 2>&1 | tail -1 | cut -c1-200
