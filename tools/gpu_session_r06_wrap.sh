#!/bin/bash
# round-6 wrap-up on the committed build: full GPU suite, race screen, PMC (traffic + MFMA counters), the driver's line twice
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_wrap; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu > $O/gputests.log 2>&1; tail -2 $O/gputests.log
timeout 900 python tools/race_screen.py 100 2>&1 | grep -v amdgpu > $O/race_screen.log; tail -1 $O/race_screen.log
bash tools/gpu_session_r06_pmc.sh r06_wrap_pmc | tail -4
cp gpurun_out/r06_wrap_pmc/pmc_traffic.json gpurun_out/r06_wrap_pmc/pmc_mfma/pmc_mfma.json gpurun_out/r06_wrap_pmc/pmc_mfma/pmc_mfma.md gpurun_out/r06_wrap_pmc/bench_bf16x3_default.json $O/ 2>/dev/null
mkdir -p $O/pmc_pass; cp gpurun_out/r06_wrap_pmc/pmc_mfma/pass*.csv $O/pmc_pass/
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench2_bf16x3_default.json 2> $O/bench2.err
python - $O <<'PY'
import json, sys
for f in ('bench_bf16x3_default', 'bench2_bf16x3_default'):
    b = json.loads(open('%s/%s.json' % (sys.argv[1], f)).read().strip().splitlines()[-1]); r = b['roofline']
    print(f, round(b['value'], 2), round(b['ms_per_step'], 2), round(r['achieved'], 1), r.get('traffic'), r.get('mfma_busy'))
PY
