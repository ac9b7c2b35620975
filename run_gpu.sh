cd $GRAFT_REPO_ROOT
for v in new old new old; do
  if [ $v = new ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$GRAFT_REPO_ROOT/gan-control_amd/csrc/build/exp/lib_old.so; fi
  echo == $v; python tools/fir_odd_bench.py 2>&1 | grep blur
done
