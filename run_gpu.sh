#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout -k 10 300 python bench.py --steps 32 --warmup 16 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_latest.json
cut -c1-200 gpurun_out/bench_latest.json
timeout -k 10 300 python tools/train_sanity.py 2>&1 | tail -3
