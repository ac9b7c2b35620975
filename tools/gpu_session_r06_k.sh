#!/bin/bash
# Round 6, session K: deferred stores in the 32-output-channel ws variant: tests of its shapes, kbench A/B
O=gpurun_out/r06_k; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "conv2d_bf16x3_kernel or fused_epilogue" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
A=$PWD/gan-control_amd/csrc/alt
for b in 2 4 8; do
for lib in main nodefer; do
  if [ $lib = main ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$A/libalt_$lib.so; fi
  echo "== lib=$lib B=$b" >> $O/kbench.log
  timeout 300 python tools/kbench.py --mode bf16x3 --batch $b --only "conv3x3 s1 32->32" --reps 20 2>&1 | grep "conv3x3" | grep -v wgrad >> $O/kbench.log
  timeout 300 python tools/kbench.py --mode bf16x3 --batch $b --only "conv1x1 s1 64->32" --reps 20 2>&1 | grep "conv1x1" | grep -v wgrad >> $O/kbench.log
done
done
unset GANCONTROL_HIP_LIB
cat $O/kbench.log
