#!/bin/bash
# Round 6, session N: weight-slab DMA issued by one multiplying wave per SIMD (GC_WS_DMA_HALF): tests of the ws shapes, kbench A/B, trace
O=gpurun_out/r06_n; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "conv2d_bf16x3_kernel or fused_epilogue" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
A=$PWD/gan-control_amd/csrc/alt
for b in 4 8; do
for lib in main dmamid main dmamid; do
  if [ $lib = main ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$A/libalt_$lib.so; fi
  echo "== lib=$lib B=$b" >> $O/kbench.log
  timeout 600 python tools/kbench.py --mode bf16x3 --batch $b --only "conv3x3 s1" --reps 20 2>&1 | grep "conv3x3" | grep -v wgrad | grep "_ws_" >> $O/kbench.log
done
done
unset GANCONTROL_HIP_LIB
cat $O/kbench.log
GANCONTROL_HIP_LIB=$A/libalt_dmamid.so timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "conv2d_bf16x3_kernel or fused_epilogue" 2>&1 | tail -2
