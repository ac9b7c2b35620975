#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "=== iter 20 (path)"; timeout -k 10 300 python tools/aten_profile.py --iter 20 --top 40 2>&1 | grep -v "amdgpu.ids\|Warn\|warn"
echo "=== iter 16 (r1 + path)"; timeout -k 10 300 python tools/aten_profile.py --iter 16 --top 25 2>&1 | grep -v "amdgpu.ids\|Warn\|warn"
