# time and fabric traffic of the wave-specialised forward kernel, in-tree library against csrc/alt/libalt_<name>.so: tools/ws_xcd_ab.sh <name>
R=$PWD; out=gpurun_out/ws_xcd_ab.log; rm -f $out
for rep in 1 2; do for b in 4 8; do for lib in new $1; do
  if [ $lib = new ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$R/gan-control_amd/csrc/alt/libalt_$lib.so; fi
  echo "== $lib B=$b" >> $out
  python tools/kbench.py --mode bf16x3 --batch $b --reps 20 --only "conv3x3 s1" 2>&1 | grep "^conv3x3" >> $out
done; done; done
for lib in new $1; do
  if [ $lib = new ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$R/gan-control_amd/csrc/alt/libalt_$lib.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/tools/pmc_mix.py > /dev/null 2>&1)
  done
  python tools/pmc_mix.py --parse gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_xcd_${lib}_traffic.json | tail -12 >> $out; rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
done
