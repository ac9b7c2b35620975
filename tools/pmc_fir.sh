#!/bin/bash
# PMC passes over the 4 x 4 FIR tile kernel (one --pmc set per run, --kernel-trace only: the pool's rules); run from the repo root on a GPU box.
# usage: tools/pmc_fir.sh <variant>   -> gpurun_out/pmc_fir_<variant>/pass*.csv
v=${1:-plain}
out=gpurun_out/pmc_fir_$v
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/raw$i -- python3 tools/one_fir.py 4 32 1025 $v > $out/log$i.txt 2>&1
  f=$(find $out/raw$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then grep -E "Counter_Name|fir44_tile" "$f" > $out/pass$i.csv; fi
  rm -rf $out/raw$i
done
ls -la $out
