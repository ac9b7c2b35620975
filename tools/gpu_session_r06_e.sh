#!/bin/bash
# Round 6, session E: re-measure the parity table on this build (ratchet), MFMA counters of the shipped matrix kernels, 512->512 @65^2 on the stride-2 ws kernel.
O=gpurun_out/r06_e; mkdir -p $O
timeout 900 python tools/headline_parity_probe.py --write step_1024_b4 step_512_b16 > $O/parity_probe.log 2>&1; tail -3 $O/parity_probe.log
cp tests/golden/parity_measured.json $O/parity_measured.json
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_baseline_configs.py -q -x -m gpu -k "headline_iteration_against or config2_512" -s > $O/ratchet_tests.log 2>&1; tail -3 $O/ratchet_tests.log
bash tools/pmc_mfma.sh $O/pmc_mfma > $O/pmc_mfma.log 2>&1; tail -12 $O/pmc_mfma.log
A=$PWD/gan-control_amd/csrc/alt
for lib in main s2min128 s2min64; do
  if [ $lib = main ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$A/libalt_$lib.so; fi
  for b in 4 8; do
    echo "== lib=$lib B=$b" >> $O/kbench_s2_min.log
    timeout 300 python tools/kbench.py --mode bf16x3 --batch $b --only "conv3x3 s2 512->512 @65" --reps 20 2>&1 | grep "conv3x3" | grep -v wgrad >> $O/kbench_s2_min.log
  done
done
unset GANCONTROL_HIP_LIB
cat $O/kbench_s2_min.log
