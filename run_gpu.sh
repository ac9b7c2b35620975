cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for P in bf16x3 f32; do
timeout 1500 python bench.py --no-cpu-baseline --precision $P 2>&1 | tail -1 > gpurun_out/bench_v5_$P.json
python - $P <<'PY'
import json,sys
d=json.load(open('gpurun_out/bench_v5_%s.json'%sys.argv[1]))
print(sys.argv[1],'value',d['value'],'ms/step',d['ms_per_step'], d['losses'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['total_ms'])[:8]: print('%-58s %6d launches avg %9.1f us total %8.1f ms  %8.2f %s'%(k,v['launches'],v['avg_us'],v['total_ms'],v['achieved'],v['unit']))
print('timed kernels total ms', sum(v['total_ms'] for v in d['kernels'].values()))
PY
done
