cd $GRAFT_REPO_ROOT
for e in 0 30 0 30; do
  if [ $e = 0 ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$GRAFT_REPO_ROOT/gan-control_amd/csrc/build/exp/lib_exp$e.so; fi
  echo "== EXP $e"
  python tools/convt_bench.py 2>&1 | grep "convT"
done
unset GANCONTROL_HIP_LIB
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3
for f in 1 1; do timeout 600 python bench.py --no-cpu-baseline --no-kernel-timer 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-140; done
