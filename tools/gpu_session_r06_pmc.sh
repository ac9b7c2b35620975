#!/bin/bash
# PMC passes (traffic + MFMA counters) on the committed build, then the driver's line with them next to it.  $1 = tag
cd $GRAFT_REPO_ROOT; tag=${1:-r06_pmc}; O=gpurun_out/$tag; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp OMP_NUM_THREADS=32
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/pmc_$c -- python3 $R/tools/pmc_mix.py > /dev/null 2>&1)
done
python tools/pmc_mix.py --parse $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_traffic.json | tail -2; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
bash tools/pmc_mfma.sh $O/pmc_mfma > $O/pmc_mfma.log 2>&1; tail -9 $O/pmc_mfma.log | cut -c1-120
cp $O/pmc_traffic.json profiles/pmc_r06_traffic.json
cp $O/pmc_mfma/pmc_mfma.json profiles/pmc_r06_mfma.json
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_bf16x3_default.json 2> $O/bench.err; echo "bench rc=$?"
python - $O <<'PY'
import json, sys
b = json.loads(open(sys.argv[1] + '/bench_bf16x3_default.json').read().strip().splitlines()[-1])
r = b['roofline']
print(round(b['value'], 2), round(b['ms_per_step'], 2), r.get('achieved'), r.get('traffic'), r.get('mfma_busy'), r.get('mfma_busy_source', '')[:80])
PY
