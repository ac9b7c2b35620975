# same-box A/B of library builds on the whole step: [BENCH_ARGS="--batch-per-gpu 2"] tools/step_ab.sh <tag> <lib> <lib> ...   ("new" = the in-tree library, else csrc/alt/libalt_<name>.so); three alternations
tag=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    if [ $lib = new ]; then unset GANCONTROL_HIP_LIB; else export GANCONTROL_HIP_LIB=$PWD/gan-control_amd/csrc/alt/libalt_$lib.so; fi
    python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-fp32-leg --no-host-issue --no-families $BENCH_ARGS > gpurun_out/bench_${tag}_${lib}_$rep.json 2> gpurun_out/bench_${tag}_${lib}_$rep.err
  done
done
python - $tag <<'PY'
import glob, json, sys
for f in sorted(glob.glob('gpurun_out/bench_%s_*.json' % sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/bench_*.json')):
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1]); print(f, b['value'], b['ms_per_step'])
    except Exception as e: print(f, 'failed', e)
PY
