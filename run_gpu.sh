cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -2
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/bench_g_bf16x3.json
timeout 900 python bench.py --precision f32 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/bench_g_f32.json
timeout 900 python bench.py --size 512 --batch-per-gpu 16 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/bench_g_config1.json
for m in bf16x3 f32; do
  rm -rf gpurun_out/prof_$m && mkdir -p gpurun_out/prof_$m
  timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$m -o $m --output-format csv -- python3 bench.py --precision $m --no-cpu-baseline > gpurun_out/prof_bench_$m.log 2>&1
  cp $(find gpurun_out/prof_$m -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_$m.csv
  find gpurun_out/prof_$m -name "*kernel_trace.csv" -delete
done
cut -c1-200 gpurun_out/bench_g_bf16x3.json; echo; cut -c1-200 gpurun_out/bench_g_f32.json; echo; cut -c1-200 gpurun_out/bench_g_config1.json
