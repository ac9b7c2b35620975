#!/bin/bash
# round-5 session: early weight DMA / counted vmcnt in the ws kernel (alt/libalt_noearly.so = without)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
tag=${1:-r05_g}
export OMP_NUM_THREADS=32
R=$GRAFT_REPO_ROOT
A=$R/gan-control_amd/csrc/alt
timeout 900 python tools/race_screen.py 300 2>&1 | grep -v amdgpu > gpurun_out/race_screen_$tag.log; tail -16 gpurun_out/race_screen_$tag.log
{
echo "== epilogue probe: early weight DMA"; python tools/epilogue_probe.py 2>&1 | grep -v amdgpu
echo "== epilogue probe: without (GC_WS_EARLY_DMA=0)"; GANCONTROL_HIP_LIB=$A/libalt_noearly.so python tools/epilogue_probe.py 2>&1 | grep -v amdgpu
echo "== epilogue probe B=8: early weight DMA"; python tools/epilogue_probe.py 8 2>&1 | grep -v amdgpu
echo "== epilogue probe B=8: without (GC_WS_EARLY_DMA=0)"; GANCONTROL_HIP_LIB=$A/libalt_noearly.so python tools/epilogue_probe.py 8 2>&1 | grep -v amdgpu
} > gpurun_out/kernel_ab_$tag.log 2>&1
cat gpurun_out/kernel_ab_$tag.log
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > gpurun_out/gputests_$tag.log; cat gpurun_out/gputests_$tag.log
Q="--steps 32 --warmup 16 --no-cpu-baseline --no-fp32-leg --no-families --no-host-issue"
for i in 1 2 3; do
  timeout 600 python bench.py $Q > gpurun_out/bench_${tag}_new_$i.json 2>/dev/null
  GANCONTROL_HIP_LIB=$A/libalt_noearly.so timeout 600 python bench.py $Q > gpurun_out/bench_${tag}_noearly_$i.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_${tag}_*.json')):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(b['value'],2), round(b['ms_per_step'],2), (b.get('roofline') or {}).get('achieved'))
    except Exception as e: print(f, 'failed', e)
PY
