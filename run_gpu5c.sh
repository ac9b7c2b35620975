#!/bin/bash
# round-5 GPU session 3 ($1 = tag): residual-as-template epilogue + pipelined MFMA phases of the one-role / transposed kernels against round-4 kernels
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
tag=${1:-r05_c}
export OMP_NUM_THREADS=32
R=$GRAFT_REPO_ROOT
A=$R/gan-control_amd/csrc/alt
OLD=$A/libalt_r04.so
{
echo "== epilogue probe: new library"; python tools/epilogue_probe.py 2>&1 | grep -v amdgpu
echo "== epilogue probe: round-4 kernels"; GANCONTROL_HIP_LIB=$OLD python tools/epilogue_probe.py 2>&1 | grep -v amdgpu
for B in 4 8; do
for sel in "3x3 s2" "convT3x3"; do
echo "== kbench bf16x3 B=$B $sel: new library"; python tools/kbench.py --mode bf16x3 --batch $B --only "$sel" 2>&1 | grep -v "amdgpu\|wgrad"
echo "== kbench bf16x3 B=$B $sel: no fragment pipeline (GC_FRAG_PIPE=0)"; GANCONTROL_HIP_LIB=$A/libalt_nopipe.so python tools/kbench.py --mode bf16x3 --batch $B --only "$sel" 2>&1 | grep -v "amdgpu\|wgrad"
echo "== kbench bf16x3 B=$B $sel: round-4 kernels"; GANCONTROL_HIP_LIB=$OLD python tools/kbench.py --mode bf16x3 --batch $B --only "$sel" 2>&1 | grep -v "amdgpu\|wgrad"
done
echo "== kbench bf16x3 B=$B convT3x3: weight slab by LDS-DMA + pipeline (GC_CT_DMA=1)"; GANCONTROL_HIP_LIB=$A/libalt_ctdma.so python tools/kbench.py --mode bf16x3 --batch $B --only "convT3x3" 2>&1 | grep -v "amdgpu\|wgrad"
echo "== kbench bf16x3 B=$B small stride-1 planes: new"; python tools/kbench.py --mode bf16x3 --batch $B --only "s1 512->512 @" 2>&1 | grep -v "amdgpu\|wgrad"
echo "== kbench bf16x3 B=$B small stride-1 planes: round-4"; GANCONTROL_HIP_LIB=$OLD python tools/kbench.py --mode bf16x3 --batch $B --only "s1 512->512 @" 2>&1 | grep -v "amdgpu\|wgrad"
done
} > gpurun_out/kernel_ab_$tag.log 2>&1
cat gpurun_out/kernel_ab_$tag.log
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/gputests_$tag.log; cat gpurun_out/gputests_$tag.log
timeout 900 python tools/race_screen.py 100 2>&1 | grep -v amdgpu > gpurun_out/race_screen_$tag.log; tail -4 gpurun_out/race_screen_$tag.log
Q="--steps 32 --warmup 16 --no-cpu-baseline --no-fp32-leg --no-families --no-host-issue"
for i in 1 2 3; do
  timeout 600 python bench.py $Q > gpurun_out/bench_${tag}_new_$i.json 2>/dev/null
  GANCONTROL_HIP_LIB=$OLD timeout 600 python bench.py $Q > gpurun_out/bench_${tag}_old_$i.json 2>/dev/null
done
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o bench -- python3 $R/bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-fp32-leg --no-families --no-host-issue > $R/gpurun_out/bench_${tag}_under_rocprof.json 2> $R/gpurun_out/bench_${tag}_prof.err
cd $R
DB=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py $DB > gpurun_out/rocprof_${tag}_kernel_stats_bf16x3_1024_b4.csv && rm -rf gpurun_out/prof_$tag
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/tools/pmc_mix.py > /dev/null 2>&1)
done
python tools/pmc_mix.py --parse gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_${tag}_traffic.json | tail -3; rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
cp gpurun_out/pmc_${tag}_traffic.json profiles/pmc_r05_traffic.json
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${tag}_bf16x3_default.json 2> gpurun_out/bench_$tag.err; echo "bench rc=$?"
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_${tag}_*.json')):
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(b['value'],2), round(b['ms_per_step'],2), (b.get('roofline') or {}).get('achieved'), (b.get('roofline') or {}).get('traffic'), (b.get('fp32_exact') or {}).get('value'), b.get('host_issue_ms_per_step'))
    except Exception as e: print(f, 'failed', e)
b=json.loads(open('gpurun_out/bench_${tag}_bf16x3_default.json').read().strip().splitlines()[-1])
for k,v in b['families'].items(): print(k, v['ms_per_step'], v['launches_per_step'], v['achieved'])
PY
head -24 gpurun_out/rocprof_${tag}_kernel_stats_bf16x3_1024_b4.csv | cut -c1-150
