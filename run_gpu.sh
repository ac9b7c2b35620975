cd $GRAFT_REPO_ROOT
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_r01_d.json
timeout 900 python bench.py --precision f32 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_r01_d_f32.json
timeout 900 python bench.py --size 512 --batch-per-gpu 16 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_r01_d_c2.json
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01e -o r01e -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/prof_r01e_bench.log 2>&1
rm -f $R/gpurun_out/prof_r01e/*kernel_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01f -o r01f -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timer --precision f32 > $R/gpurun_out/prof_r01f_bench.log 2>&1
rm -f $R/gpurun_out/prof_r01f/*kernel_trace.csv
