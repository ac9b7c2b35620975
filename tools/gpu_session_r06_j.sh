#!/bin/bash
# Round 6, session J (review item 3): the 32- / 64-channel variants of the ws stride-1 kernel under ablation -- no patch staging (1), no weight DMA (2), no stores (8),
# no conversion (16: the pre-split bound), no staging and no stores (9) -- on 32 -> 32 @1024^2 and 64 -> 64 @512^2, B = 4 / 8
O=gpurun_out/r06_j; mkdir -p $O
A=$PWD/gan-control_amd/csrc/alt
for b in 4 8; do
for lib in main wsabl16 wsabl1 wsabl2 wsabl8 wsabl9; do
  if [ $lib = main ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$A/libalt_$lib.so; fi
  echo "== lib=$lib B=$b" >> $O/kbench_lowchan.log
  timeout 300 python tools/kbench.py --mode bf16x3 --batch $b --only "conv3x3 s1 32->32" --reps 20 2>&1 | grep "conv3x3" | grep -v wgrad >> $O/kbench_lowchan.log
  timeout 300 python tools/kbench.py --mode bf16x3 --batch $b --only "conv3x3 s1 64->64" --reps 20 2>&1 | grep "conv3x3" | grep -v wgrad >> $O/kbench_lowchan.log
done
done
unset GANCONTROL_HIP_LIB
cat $O/kbench_lowchan.log
