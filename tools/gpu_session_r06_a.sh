#!/bin/bash
# Round 6, session A: the wave-specialised stride-2 kernel -- parity on its shapes, then same-box kbench against the one-role kernel and two ablations.
O=gpurun_out/r06_a; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "conv2d_bf16x3_kernel or fused_epilogue or row_pitched" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
A=gan-control_amd/csrc/alt
for b in 4 8; do
  for lib in main nos2ws s2a1 s2a4; do
    if [ $lib = main ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$PWD/$A/libalt_$lib.so; fi
    echo "== lib=$lib B=$b" >> $O/kbench.log
    timeout 300 python tools/kbench.py --mode bf16x3 --batch $b --only "conv3x3 s2" --reps 20 2>&1 | grep -v "^wgrad" >> $O/kbench.log
  done
done
unset GANCONTROL_HIP_LIB
cat $O/kbench.log
