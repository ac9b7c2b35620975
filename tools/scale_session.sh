#!/bin/bash
# The first session on an 8-GPU node, as ONE command (nothing here has ever run on more than one GPU: DESIGN.md section 5).
#   tools/scale_session.sh [out_dir]        -> one JSON line per run in <out_dir>/scale.jsonl, stderr of every run next to it
# 1. the scaling points the driver measures: bench.py --gpus 1 / 2 / 4 / 8 (weak scaling, 4 images per GPU);
# 2. at 8 GPUs, the knobs of the gradient all-reduce: RCCL algorithm / protocol, bucket size, size of the LAST bucket of a backward pass
#    (xGMI is point-to-point, 7 links x ~153 GB/s per GPU: a ring all-reduce is per-link bound, small tail buckets shorten the exposed part).
# bench.py starts the N-rank job itself (torch.distributed.run, 127.0.0.1 rendezvous); the flags below only set the ranks' environment.
O=${1:-gpurun_out/scale}; mkdir -p $O
: > $O/scale.jsonl
run() {      # run <tag> <bench.py args...>
    tag=$1; shift
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg "$@" > $O/$tag.json 2> $O/$tag.err
    line=$(tail -n 1 $O/$tag.json)
    [ -n "$line" ] && echo "{\"tag\": \"$tag\", \"line\": $line}" >> $O/scale.jsonl || echo "{\"tag\": \"$tag\", \"line\": null}" >> $O/scale.jsonl
}
N=$(python - <<'PY'
import importlib.util, sys
sys.argv = ['x']
spec = importlib.util.spec_from_file_location('b', 'bench.py'); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
print(m.visible_gpu_count())
PY
)
for n in 1 2 4 8; do [ $n -le $N ] && run gpus$n --gpus $n; done
top=$N; [ $top -gt 8 ] && top=8
if [ $top -gt 1 ]; then
    for algo in Ring Tree; do run g${top}_algo_$algo --gpus $top --rccl-algo $algo; done
    for proto in Simple LL128; do run g${top}_proto_$proto --gpus $top --rccl-proto $proto; done
    for mb in 8 16 64; do run g${top}_bucket_$mb --gpus $top --bucket-mb $mb; done
    for mb in 1 4; do run g${top}_last_$mb --gpus $top --last-bucket-mb $mb; done
fi
python - $O <<'PY'
import json, sys
base = None
for l in open(sys.argv[1] + '/scale.jsonl'):
    r = json.loads(l); b = r['line']
    if not b: print('%-24s failed' % r['tag']); continue
    if r['tag'] == 'gpus1': base = b['value']
    c = b.get('comm') or {}
    print('%-24s n=%d %8.2f images/s %7.2f ms/step  x%.2f vs 1 GPU  exposed comm %s ms' % (r['tag'], b['n_gpus'], b['value'], b['ms_per_step'], b['value'] / base if base else float('nan'), c.get('exposed_ms_per_step')))
PY
