#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "wgrad or bf16" 2>&1 | tail -2
