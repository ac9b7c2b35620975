#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "upfirdn2d" 2>&1 | tail -4
timeout -k 10 300 python tools/fir_profile.py 2>&1 | grep -E "x\(., 512, 3|total"
