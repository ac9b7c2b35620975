cd $GRAFT_REPO_ROOT
GC_T_OC32=1 timeout 600 python tools/kbench.py --only "convT" --mode bf16x3 2>&1 | grep -v amdgpu.ids
