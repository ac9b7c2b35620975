cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "ddp_path" 2>&1 | tail -5
timeout 900 python bench.py --no-cpu-baseline --precision bf16x3 2>&1 | tail -1 | cut -c1-230
