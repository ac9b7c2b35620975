#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/path_profile.py 2>&1 | grep -v amdgpu > gpurun_out/path_profile_r05.log; head -50 gpurun_out/path_profile_r05.log
TOP=45 python tools/host_profile.py iter inline 2>&1 | grep -v amdgpu > gpurun_out/host_profile_iter_r05.log; head -75 gpurun_out/host_profile_iter_r05.log
