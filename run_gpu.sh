#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 300 python bench.py 2>&1 | tail -1 > gpurun_out/bench_i_bf16x3.json
timeout -k 10 300 python bench.py --precision f32 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_i_f32.json
timeout -k 10 300 python bench.py --size 512 --batch-per-gpu 16 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_i_config1.json
for f in bf16x3 f32 config1; do cut -c1-170 gpurun_out/bench_i_$f.json; echo; done
rm -rf gpurun_out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r01 --output-format csv -- python3 bench.py --steps 16 --warmup 16 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/kernel_stats_bf16x3.csv
find gpurun_out/prof -name '*kernel_trace.csv' -delete
