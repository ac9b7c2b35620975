cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "upfirdn" 2>&1 | tail -2
for v in new old new old; do
  if [ $v = new ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$GRAFT_REPO_ROOT/gan-control_amd/csrc/build/exp/lib_old.so; fi
  echo == $v; python tools/fir_bench.py 2>&1 | grep "fir\|rgb" | head -8
done
