#!/bin/bash
# Round 6, session L: pixel-major accumulators + 16-byte stores in the ws stride-1 and stride-2 kernels: tests, kbench A/B against the dword-store builds
O=gpurun_out/r06_l; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "conv2d_bf16x3_kernel or fused_epilogue or row_pitched or modulated_conv or equal_conv or resblock" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
A=$PWD/gan-control_amd/csrc/alt
for b in 4 8; do
for lib in main notstore olds2; do
  if [ $lib = main ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$A/libalt_$lib.so; fi
  echo "== lib=$lib B=$b" >> $O/kbench.log
  timeout 600 python tools/kbench.py --mode bf16x3 --batch $b --only "conv3x3 s" --reps 20 2>&1 | grep "conv3x3" | grep -v wgrad | grep "_ws_\|s2ws" >> $O/kbench.log
done
done
unset GANCONTROL_HIP_LIB
cat $O/kbench.log
