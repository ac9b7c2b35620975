cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -4
timeout 600 python tools/kbench.py --only "conv3x3" --mode f32 2>&1 | grep -v amdgpu.ids | grep -E "s1 (512->512 @(4|16|64)|128|32->32)|s2 (64|128)"
