#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo no-loads; GANCONTROL_HIP_LIB=$PWD/gan-control_amd/csrc/build/libabl3.so timeout 300 python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids
