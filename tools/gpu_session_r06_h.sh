#!/bin/bash
# Round 6, session H: the driver's bench command on the current build, then the full GPU suite.
O=gpurun_out/r06_h; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
python - <<'PY'
import json
b = json.loads(open('gpurun_out/r06_h/bench_default.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b['roofline'])
for k, v in b['families'].items(): print(k, v)
PY
timeout 1500 python -m pytest tests -q -x -m gpu > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log; tail -4 $O/gputests.log
