#!/bin/bash
# Round 6, session I: deep prefetch in the 64-channel stride-2 ws variant; 64 -> 128 on that variant
O=gpurun_out/r06_i; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "conv2d_bf16x3_kernel or fused_epilogue or row_pitched" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
A=$PWD/gan-control_amd/csrc/alt
for b in 4 8; do
for lib in main s2nodeep s2w1k64; do
  if [ $lib = main ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$A/libalt_$lib.so; fi
  echo "== lib=$lib B=$b" >> $O/kbench_s2.log
  timeout 300 python tools/kbench.py --mode bf16x3 --batch $b --only "conv3x3 s2" --reps 20 2>&1 | grep "s2ws" >> $O/kbench_s2.log
done
done
unset GANCONTROL_HIP_LIB
cat $O/kbench_s2.log
