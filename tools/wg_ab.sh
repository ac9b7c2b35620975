# same-box A/B of library builds on the stride-1 3x3 weight gradients: tools/wg_ab.sh <lib> <lib> ...
out=gpurun_out/wg_ab.log; rm -f $out
for rep in 1 2; do
for b in 4 8 2; do
  for lib in "$@"; do
    if [ $lib = new ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$PWD/gan-control_amd/csrc/alt/libalt_$lib.so; fi
    echo "== $lib B=$b" >> $out
    python tools/kbench.py --mode bf16x3 --batch $b --reps 20 --only "conv3x3 s" 2>&1 | grep "wgrad" >> $out
  done
done
done
