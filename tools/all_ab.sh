# same-box A/B of library builds over every convolution / weight-gradient shape of kbench.py: tools/all_ab.sh "<batches>" <lib> <lib> ...
bs=$1; shift
out=gpurun_out/all_ab.log; rm -f $out
for rep in 1 2; do for b in $bs; do for lib in "$@"; do
  if [ $lib = new ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$PWD/gan-control_amd/csrc/alt/libalt_$lib.so; fi
  echo "== $lib B=$b" >> $out
  python tools/kbench.py --mode bf16x3 --batch $b --reps 20 --only "conv" 2>&1 | grep "conv" >> $out
done; done; done
