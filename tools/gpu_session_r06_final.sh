#!/bin/bash
# round-6 final measurement session ($1 = tag): smoke, -m gpu tests, race screen, driver-style bench, config-2 lines, rocprofv3 kernel stats (split-bf16 and exact fp32),
# PMC traffic + MFMA counters of THIS build (stamped with its source hash), then the driver's line once more with those counters next to it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
tag=${1:-r06_final}
O=gpurun_out/$tag; mkdir -p $O
export OMP_NUM_THREADS=32
R=$GRAFT_REPO_ROOT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1800 python -m pytest tests -q -m gpu > $O/gputests.log 2>&1; tail -3 $O/gputests.log
timeout 900 python tools/race_screen.py 100 2>&1 | grep -v amdgpu > $O/race_screen.log; tail -7 $O/race_screen.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_bf16x3_default.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --size 512 --batch-per-gpu 16 --steps 16 --warmup 8 --no-cpu-baseline --no-fp32-leg --no-families > $O/bench_bf16x3_512_b16.json 2>/dev/null
timeout 600 python bench.py --size 512 --batch-per-gpu 16 --steps 16 --warmup 8 --precision bf16 --no-cpu-baseline --no-fp32-leg --no-families > $O/bench_bf16_512_b16.json 2>/dev/null
timeout 600 python bench.py --precision f32 --steps 16 --warmup 8 --no-cpu-baseline --no-fp32-leg --no-families > $O/bench_f32.json 2>/dev/null
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python3 $R/bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-fp32-leg --no-families --no-host-issue > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_prof.err
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py $DB > $O/rocprof_kernel_stats_bf16x3_1024_b4.csv && rm -rf $O/prof
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof_f32 -o bench -- python3 $R/bench.py --precision f32 --steps 16 --warmup 8 --no-cpu-baseline --no-fp32-leg --no-families --no-host-issue > $R/$O/bench_f32_under_rocprof.json 2> $R/$O/bench_f32_prof.err
cd $R
DB=$(find $O/prof_f32 -name "*.db" | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py $DB > $O/rocprof_kernel_stats_f32_1024_b4.csv && rm -rf $O/prof_f32
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/pmc_$c -- python3 $R/tools/pmc_mix.py > /dev/null 2>&1)
done
python tools/pmc_mix.py --parse $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_traffic.json | tail -10; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
bash tools/pmc_mfma.sh $O/pmc_mfma > $O/pmc_mfma.log 2>&1; tail -10 $O/pmc_mfma.log | cut -c1-200
# the line once more with this build's own counters next to it (quoted only for a matching source hash)
cp $O/pmc_traffic.json profiles/pmc_r06_traffic.json
cp $O/pmc_mfma/pmc_mfma.json profiles/pmc_r06_mfma.json
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench2_bf16x3_default.json 2> $O/bench2.err; echo "bench2 rc=$?"
python - $O <<'PY'
import json, sys
O = sys.argv[1]
for f in ['bench_bf16x3_default', 'bench2_bf16x3_default', 'bench_bf16x3_512_b16', 'bench_bf16_512_b16', 'bench_f32', 'bench_under_rocprof']:
    try:
        b = json.loads(open('%s/%s.json' % (O, f)).read().strip().splitlines()[-1])
        r = b.get('roofline') or {}
        print(f, round(b['value'], 2), round(b['ms_per_step'], 2), r.get('achieved'), r.get('traffic'), r.get('mfma_busy'), (b.get('fp32_exact') or {}).get('value'), (b.get('cpu_baseline') or {}).get('value'), b.get('host_issue_ms_per_step'))
    except Exception as e:
        print(f, 'failed', e)
b = json.loads(open(O + '/bench2_bf16x3_default.json').read().strip().splitlines()[-1])
for k, v in b['families'].items():
    print(k, v['ms_per_step'], v['launches_per_step'], v['achieved'], v['unit'])
PY
head -12 $O/rocprof_kernel_stats_bf16x3_1024_b4.csv | cut -c1-150
