#!/bin/bash
# round-5 GPU session ($1 = tag): kernel A/Bs first (short, the round's decisions hang on them), then smoke, -m gpu tests, driver-style bench, forced
# single-rank DDP bench, fp32 rocprof, training sanity, PMC traffic
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
tag=${1:-r05_a}
export OMP_NUM_THREADS=32
R=$GRAFT_REPO_ROOT
A=$R/gan-control_amd/csrc/alt
{
echo "== epilogue probe: new library"; python tools/epilogue_probe.py 2>&1 | grep -v amdgpu
echo "== epilogue probe: old epilogue (HEAD~ conv_bf16x3.hip)"; GANCONTROL_HIP_LIB=$A/libalt_oldepi.so python tools/epilogue_probe.py 2>&1 | grep -v amdgpu
for B in 4 8; do
echo "== stride 2 / transposed, B=$B: shipped"; python tools/kbench.py --mode bf16x3 --batch $B --only "3x3 s2" 2>&1 | grep -v "amdgpu\|wgrad"; python tools/kbench.py --mode bf16x3 --batch $B --only "convT3x3" 2>&1 | grep -v amdgpu
echo "== stride 2, B=$B: no conversion (GC_S2_ABL=1)"; GANCONTROL_HIP_LIB=$A/libalt_s2a1.so python tools/kbench.py --mode bf16x3 --batch $B --only "3x3 s2" 2>&1 | grep -v "amdgpu\|wgrad"
echo "== stride 2, B=$B: no patch staging at all (GC_S2_ABL=2)"; GANCONTROL_HIP_LIB=$A/libalt_s2a2.so python tools/kbench.py --mode bf16x3 --batch $B --only "3x3 s2" 2>&1 | grep -v "amdgpu\|wgrad"
echo "== transposed, B=$B: no conversion (GC_CT_ABL=8)"; GANCONTROL_HIP_LIB=$A/libalt_cta8.so python tools/kbench.py --mode bf16x3 --batch $B --only "convT3x3" 2>&1 | grep -v amdgpu
done
} > gpurun_out/kernel_ab_$tag.log 2>&1
cat gpurun_out/kernel_ab_$tag.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/gputests_$tag.log; cat gpurun_out/gputests_$tag.log
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${tag}_bf16x3_default.json 2> gpurun_out/bench_$tag.err; echo "bench rc=$?"
timeout 600 python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-fp32-leg --no-families > gpurun_out/bench_${tag}_plain.json 2>/dev/null
GANCONTROL_HIP_LIB=$A/libalt_oldepi.so timeout 600 python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-fp32-leg --no-families > gpurun_out/bench_${tag}_plain_oldepi.json 2>/dev/null
timeout 600 python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-fp32-leg --no-families > gpurun_out/bench_${tag}_plain2.json 2>/dev/null
timeout 600 python bench.py --force-ddp --steps 32 --warmup 16 --no-cpu-baseline --no-fp32-leg --no-families > gpurun_out/bench_${tag}_force_ddp.json 2> gpurun_out/bench_${tag}_force_ddp.err; echo "ddp rc=$?"
PROBE_MODES=f32,bf16x3,bf16 timeout 900 python tools/headline_parity_probe.py step_512_b16 2>&1 | grep -v amdgpu > gpurun_out/parity_512_b16_$tag.json; tail -80 gpurun_out/parity_512_b16_$tag.json
timeout 1500 python tools/train_sanity.py --out gpurun_out/train_sanity_$tag.json > gpurun_out/train_sanity_$tag.log 2>&1; echo "sanity rc=$?"; tail -30 gpurun_out/train_sanity_$tag.log
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o bench -- python3 $R/bench.py --precision f32 --steps 32 --warmup 16 --no-cpu-baseline --no-fp32-leg --no-families --no-host-issue > $R/gpurun_out/bench_${tag}_f32_under_rocprof.json 2> $R/gpurun_out/bench_${tag}_f32_prof.err
cd $R
DB=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py $DB > gpurun_out/rocprof_${tag}_kernel_stats_f32_1024_b4.csv && rm -rf gpurun_out/prof_$tag
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/tools/pmc_mix.py > /dev/null 2>&1)
done
python tools/pmc_mix.py --parse gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_${tag}_traffic.json | tail -3; rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
python - <<PY
import json
for f in ['bench_${tag}_bf16x3_default','bench_${tag}_plain','bench_${tag}_plain_oldepi','bench_${tag}_plain2','bench_${tag}_force_ddp','bench_${tag}_f32_under_rocprof']:
    try:
        b=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1])
        print(f, round(b['value'],2), round(b['ms_per_step'],2), (b.get('roofline') or {}).get('achieved'), (b.get('roofline') or {}).get('avg_launch_us'), (b.get('fp32_exact') or {}).get('value'), (b.get('cpu_baseline') or {}).get('value'), b.get('host_issue_ms_per_step'), b.get('comm'), b.get('rccl'))
    except Exception as e: print(f, 'failed', e)
b=json.loads(open('gpurun_out/bench_${tag}_bf16x3_default.json').read().strip().splitlines()[-1])
for k,v in b['families'].items(): print(k, v['ms_per_step'], v['launches_per_step'], v['achieved'])
print(json.dumps(b.get('cpu_baseline'), indent=0)[:1200])
PY
head -16 gpurun_out/rocprof_${tag}_kernel_stats_f32_1024_b4.csv | cut -c1-160
