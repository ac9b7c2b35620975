#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
tag=${1:-r05_e}
A=$GRAFT_REPO_ROOT/gan-control_amd/csrc/alt
timeout 900 python -m pytest tests -q -m gpu -x -k "wgrad" 2>&1 | tail -3
{
for B in 4 8; do
echo "== kbench bf16x3 B=$B stride 1: ws weight gradient, lean staging"; python tools/kbench.py --mode bf16x3 --batch $B --only "3x3 s1" 2>&1 | grep -v amdgpu | grep wgrad | grep -v "@4 \|@8 \|@16 "
echo "== kbench bf16x3 B=$B stride 1: + staging waves at priority 3"; GANCONTROL_HIP_LIB=$A/libalt_wgprio.so python tools/kbench.py --mode bf16x3 --batch $B --only "3x3 s1" 2>&1 | grep -v amdgpu | grep wgrad | grep -v "@4 \|@8 \|@16 "
echo "== kbench bf16x3 B=$B stride 1: one-role weight gradient (GC_WG_WS=0)"; GANCONTROL_HIP_LIB=$A/libalt_nowgws.so python tools/kbench.py --mode bf16x3 --batch $B --only "3x3 s1" 2>&1 | grep -v amdgpu | grep wgrad | grep -v "@4 \|@8 \|@16 "
done
} > gpurun_out/kernel_ab_$tag.log 2>&1
cat gpurun_out/kernel_ab_$tag.log
