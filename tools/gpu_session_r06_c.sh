#!/bin/bash
# Round 6, session C: per-shape profile of the step; knob sweep of the stride-2 ws kernel; the ws transposed kernel on the H x W main region + edge form.
O=gpurun_out/r06_c; mkdir -p $O
timeout 600 python tools/shape_profile.py > $O/shape_profile.log 2>&1
head -60 $O/shape_profile.log
A=$PWD/gan-control_amd/csrc/alt
for lib in main s2nt1 s2nt2 s2cons s2sp1 s2sp3 s2a1 s2a2 s2a4 s2a8; do
  if [ $lib = main ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$A/libalt_$lib.so; fi
  echo "== lib=$lib B=8" >> $O/kbench_s2.log
  timeout 300 python tools/kbench.py --mode bf16x3 --batch 8 --only "conv3x3 s2" --reps 20 2>&1 | grep "s2ws" >> $O/kbench_s2.log
done
cat $O/kbench_s2.log
for b in 4 8; do
  for lib in main ctws2; do
    if [ $lib = main ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$A/libalt_$lib.so; fi
    echo "== lib=$lib B=$b" >> $O/kbench_ct.log
    timeout 300 python tools/kbench.py --mode bf16x3 --batch $b --only "convT3x3" --reps 20 2>&1 | grep convT >> $O/kbench_ct.log
  done
done
unset GANCONTROL_HIP_LIB
cat $O/kbench_ct.log
