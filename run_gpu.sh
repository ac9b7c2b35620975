cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/bench_e_bf16x3.json
timeout 900 python bench.py --precision f32 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/bench_e_f32.json
timeout 900 python bench.py --size 512 --batch-per-gpu 16 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/bench_e_config1.json
for m in bf16x3 f32; do
  rm -rf gpurun_out/prof_$m && mkdir -p gpurun_out/prof_$m
  timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$m -o $m --output-format csv -- python3 bench.py --precision $m --no-cpu-baseline > gpurun_out/prof_bench_$m.log 2>&1
  cp $(find gpurun_out/prof_$m -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_$m.csv
  find gpurun_out/prof_$m -name "*kernel_trace.csv" -delete
  tail -1 gpurun_out/prof_bench_$m.log | cut -c1-200
done
cut -c1-220 gpurun_out/bench_e_bf16x3.json; echo; cut -c1-220 gpurun_out/bench_e_f32.json; echo; cut -c1-220 gpurun_out/bench_e_config1.json
