cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "bf16x3" 2>&1 | tail -4
timeout 600 python tools/kbench.py --only " s2 " --mode bf16x3 2>&1 | grep -v amdgpu.ids | grep wgrad
