cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "bf16x3" 2>&1 | tail -15
timeout 600 python tools/kbench.py --only conv --mode bf16x3 2>&1 | grep -v amdgpu.ids | grep -v wgrad
