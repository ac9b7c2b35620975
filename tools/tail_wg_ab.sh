# same-box A/B of library builds on the weight gradients of the low-resolution tail: tools/tail_wg_ab.sh <lib> <lib> ...
out=gpurun_out/tail_wg_ab.log; rm -f $out
for rep in 1 2; do
for b in 2 4 8; do
  for lib in "$@"; do
    if [ $lib = new ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$PWD/gan-control_amd/csrc/alt/libalt_$lib.so; fi
    echo "== $lib B=$b" >> $out
    python tools/kbench.py --mode bf16x3 --batch $b --reps 30 --only "512->512 @" 2>&1 | grep "wgrad" >> $out
  done
done
done
