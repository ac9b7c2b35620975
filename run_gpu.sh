cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_f -o p -- python3 $R/tools/pmc_traffic.py bf16x3 > $R/gpurun_out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_w -o p -- python3 $R/tools/pmc_traffic.py bf16x3 > $R/gpurun_out/pmc_w.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_f32 -o p -- python3 $R/tools/pmc_traffic.py f32 > $R/gpurun_out/pmc_f32.log 2>&1
rm -f $R/gpurun_out/pmc_*/p_kernel_trace.csv
