cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-kernel-timer 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200; done
