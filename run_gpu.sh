#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python bench.py --steps 32 --warmup 16 2>&1 | tail -1 > gpurun_out/bench_latest.json
cut -c1-200 gpurun_out/bench_latest.json
