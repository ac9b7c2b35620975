#!/bin/bash
# round-5 GPU session 4 ($1 = tag): the wave-specialised weight-gradient kernel (wgrad_bf16x3_ws_kernel) against the one-role kernel (alt/libalt_nowgws.so)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
tag=${1:-r05_d}
export OMP_NUM_THREADS=32
R=$GRAFT_REPO_ROOT
A=$R/gan-control_amd/csrc/alt
timeout 900 python -m pytest tests -q -m gpu -x -k "wgrad" 2>&1 | tail -8 > gpurun_out/gputests_wgrad_$tag.log; cat gpurun_out/gputests_wgrad_$tag.log
{
for B in 4 8; do
echo "== kbench bf16x3 B=$B stride 1: new library (ws weight gradient)"; python tools/kbench.py --mode bf16x3 --batch $B --only "3x3 s1" 2>&1 | grep -v amdgpu | grep wgrad
echo "== kbench bf16x3 B=$B stride 1: one-role weight gradient (GC_WG_WS=0)"; GANCONTROL_HIP_LIB=$A/libalt_nowgws.so python tools/kbench.py --mode bf16x3 --batch $B --only "3x3 s1" 2>&1 | grep -v amdgpu | grep wgrad
done
} > gpurun_out/kernel_ab_$tag.log 2>&1
cat gpurun_out/kernel_ab_$tag.log
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/gputests_$tag.log; cat gpurun_out/gputests_$tag.log
timeout 900 python tools/race_screen.py 100 2>&1 | grep -v amdgpu > gpurun_out/race_screen_$tag.log; tail -8 gpurun_out/race_screen_$tag.log
Q="--steps 32 --warmup 16 --no-cpu-baseline --no-fp32-leg --no-families --no-host-issue"
for i in 1 2 3; do
  timeout 600 python bench.py $Q > gpurun_out/bench_${tag}_new_$i.json 2>/dev/null
  GANCONTROL_HIP_LIB=$A/libalt_nowgws.so timeout 600 python bench.py $Q > gpurun_out/bench_${tag}_nowgws_$i.json 2>/dev/null
done
GANCONTROL_HIP_LIB=$A/libalt_r04.so timeout 600 python bench.py $Q > gpurun_out/bench_${tag}_r04_1.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_${tag}_*.json')):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(b['value'],2), round(b['ms_per_step'],2), (b.get('roofline') or {}).get('achieved'))
    except Exception as e: print(f, 'failed', e)
PY
