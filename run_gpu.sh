cd $GRAFT_REPO_ROOT
python tools/kbench.py --mode bf16x3 --reps 20 --only "conv3x3 s2" 2>&1 | grep "wgrad"
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3
for f in 1 1; do timeout 600 python bench.py --no-cpu-baseline --no-kernel-timer 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-140; done
