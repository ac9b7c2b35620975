cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "weight_layout" 2>&1 | tail -3
export TMPDIR=/tmp
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bf16x3 --output-format csv -- python3 bench.py --no-cpu-baseline --steps 16 --warmup 16 > gpurun_out/prof_bench.log 2>&1
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/kernel_stats_bf16x3.csv
find gpurun_out/prof -name "*kernel_trace.csv" -delete
grep -i "weight_layout\|pack_weights" gpurun_out/kernel_stats_bf16x3.csv | cut -c1-200
for f in 1 1; do timeout 600 python bench.py --no-cpu-baseline --no-kernel-timer 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-140; done
