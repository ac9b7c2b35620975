#!/bin/bash
# Round 6, session Q: stride-2 weight gradient on 32k x 128n tiles: tests, fuzz, kbench A/B
O=gpurun_out/r06_q; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "conv2d_bf16x3_kernel or row_pitched or scale_grads or modulated_conv or equal_conv or transposed_conv_grads" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
timeout 600 python tools/fuzz_conv.py 120 11 s2ws,s2,s2 2>&1 | tail -4
A=$PWD/gan-control_amd/csrc/alt
for b in 4 8; do
for lib in main non128 main non128; do
  if [ $lib = main ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$A/libalt_$lib.so; fi
  echo "== lib=$lib B=$b" >> $O/kbench.log
  timeout 600 python tools/kbench.py --mode bf16x3 --batch $b --only "conv3x3 s2" --reps 20 2>&1 | grep "^wgrad" >> $O/kbench.log
done
done
unset GANCONTROL_HIP_LIB
cat $O/kbench.log
