#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmcf gpurun_out/pmcw
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmcf -o f --output-format csv -- python3 tools/pmc_mix.py 2>&1 | tail -1 | cut -c1-160
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmcw -o w --output-format csv -- python3 tools/pmc_mix.py 2>&1 | tail -1 | cut -c1-160
timeout -k 10 120 python3 tools/pmc_mix.py --parse gpurun_out/pmcf gpurun_out/pmcw gpurun_out/pmc_r01_traffic.json
