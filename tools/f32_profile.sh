cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f32 -o bench -- python3 $R/bench.py --precision f32 --steps 16 --warmup 8 --no-cpu-baseline --no-fp32-leg --no-families --no-host-issue > $R/gpurun_out/bench_r05_final_f32_under_rocprof.json 2> $R/gpurun_out/bench_f32_prof.err
cd $R
DB=$(find gpurun_out/prof_f32 -name "*.db" | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py $DB > gpurun_out/rocprof_r05_final_kernel_stats_f32_1024_b4.csv && rm -rf gpurun_out/prof_f32
head -6 gpurun_out/rocprof_r05_final_kernel_stats_f32_1024_b4.csv | cut -c1-160
python bench.py --precision f32 --steps 16 --warmup 8 --no-cpu-baseline --no-fp32-leg --no-families > gpurun_out/bench_r05_final_f32.json 2>/dev/null; tail -c 300 gpurun_out/bench_r05_final_f32.json
