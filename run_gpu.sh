#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_ops_gpu.py 2>&1 | tail -3
timeout -k 10 300 python bench.py --steps 32 --warmup 16 2>&1 | tail -1 > gpurun_out/bench_latest.json
cut -c1-200 gpurun_out/bench_latest.json
