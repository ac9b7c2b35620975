cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3
for f in 1 1; do timeout 600 python bench.py --no-cpu-baseline --no-kernel-timer 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-140; done
