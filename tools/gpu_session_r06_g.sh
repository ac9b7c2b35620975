#!/bin/bash
# Round 6, session G: the 1x1 stride-1 convolutions (D's skip paths): shipped vs ws from K = 32 vs ablations; 32-channel transposed at 3 workgroups per CU
O=gpurun_out/r06_g; mkdir -p $O
A=$PWD/gan-control_amd/csrc/alt
for lib in main wsmink32 wsabl1 wsabl8 wsabl2; do
  if [ $lib = main ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$A/libalt_$lib.so; fi
  echo "== lib=$lib B=8" >> $O/kbench_1x1.log
  timeout 300 python tools/kbench.py --mode bf16x3 --batch 8 --only "conv1x1 s1" --reps 20 2>&1 | grep "conv1x1" | grep -v wgrad >> $O/kbench_1x1.log
done
cat $O/kbench_1x1.log
for lib in main ctocc3; do
  if [ $lib = main ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$A/libalt_$lib.so; fi
  echo "== lib=$lib B=4" >> $O/kbench_ct32.log
  timeout 300 python tools/kbench.py --mode bf16x3 --batch 4 --only "convT3x3 up2 64->32" --reps 20 2>&1 | grep "convT" >> $O/kbench_ct32.log
done
unset GANCONTROL_HIP_LIB
cat $O/kbench_ct32.log
