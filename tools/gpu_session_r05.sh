#!/bin/bash
# round-5 final measurement session ($1 = tag): smoke, -m gpu tests, driver-style bench, config-2 lines, phase times, rocprofv3 kernel stats, PMC traffic, race screen
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
tag=${1:-r05_f}
export OMP_NUM_THREADS=32
R=$GRAFT_REPO_ROOT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/gputests_$tag.log; cat gpurun_out/gputests_$tag.log
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${tag}_bf16x3_default.json 2> gpurun_out/bench_$tag.err; echo "bench rc=$?"
timeout 600 python bench.py --size 512 --batch-per-gpu 16 --steps 16 --warmup 8 --no-cpu-baseline --no-fp32-leg --no-families > gpurun_out/bench_${tag}_bf16x3_512_b16.json 2>/dev/null
timeout 600 python bench.py --size 512 --batch-per-gpu 16 --steps 16 --warmup 8 --precision bf16 --no-cpu-baseline --no-fp32-leg --no-families > gpurun_out/bench_${tag}_bf16_512_b16.json 2>/dev/null
timeout 600 python bench.py --batch-per-gpu 2 --steps 32 --warmup 16 --no-cpu-baseline --no-fp32-leg --no-families > gpurun_out/bench_${tag}_bf16x3_b2.json 2>/dev/null
python tools/phase_times.py 2>&1 | grep -v amdgpu > gpurun_out/phase_times_$tag.log; cat gpurun_out/phase_times_$tag.log
timeout 600 python tools/race_screen.py 100 2>&1 | grep -v amdgpu > gpurun_out/race_screen_$tag.log; tail -2 gpurun_out/race_screen_$tag.log
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o bench -- python3 $R/bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-fp32-leg --no-families --no-host-issue > $R/gpurun_out/bench_${tag}_under_rocprof.json 2> $R/gpurun_out/bench_${tag}_prof.err
cd $R
DB=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py $DB > gpurun_out/rocprof_${tag}_kernel_stats_bf16x3_1024_b4.csv && rm -rf gpurun_out/prof_$tag
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/tools/pmc_mix.py > /dev/null 2>&1)
done
python tools/pmc_mix.py --parse gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_${tag}_traffic.json | tail -10; rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
# the line once more with this build's own counters next to it (roofline.traffic is only quoted for a matching source hash)
cp gpurun_out/pmc_${tag}_traffic.json profiles/pmc_r05_traffic.json
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${tag}2_bf16x3_default.json 2> gpurun_out/bench_${tag}2.err; echo "bench2 rc=$?"
python - <<PY
import json
for f in ['bench_${tag}_bf16x3_default','bench_${tag}2_bf16x3_default','bench_${tag}_bf16x3_512_b16','bench_${tag}_bf16_512_b16','bench_${tag}_bf16x3_b2','bench_${tag}_under_rocprof']:
    try:
        b=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1])
        print(f, round(b['value'],2), round(b['ms_per_step'],2), (b.get('roofline') or {}).get('achieved'), (b.get('roofline') or {}).get('traffic'), (b.get('fp32_exact') or {}).get('value'), (b.get('cpu_baseline') or {}).get('value'), b.get('host_issue_ms_per_step'))
    except Exception as e: print(f, 'failed', e)
b=json.loads(open('gpurun_out/bench_${tag}2_bf16x3_default.json').read().strip().splitlines()[-1])
for k,v in b['families'].items(): print(k, v['ms_per_step'], v['launches_per_step'], v['achieved'])
PY
head -30 gpurun_out/rocprof_${tag}_kernel_stats_bf16x3_1024_b4.csv | cut -c1-150
