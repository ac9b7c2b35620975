#!/bin/bash
# PMC passes over the shipped matrix kernels (one --pmc set per run, --kernel-trace only, the program directly after `--`: the pool's rules); from the repo root on a GPU box.
#   tools/pmc_mfma.sh [out_dir]   -> <out_dir>/pass*.csv, pmc_mfma.md, pmc_mfma.json
out=${1:-gpurun_out/pmc_r06_mfma}
mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
i=0
python3 tools/pmc_mfma.py --sets | while read -r set; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/raw$i -- python3 tools/pmc_mfma.py > $out/log$i.txt 2>&1
  f=$(find $out/raw$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then grep -E "Counter_Name|conv_bf16x3_ws_kernel|conv_s2ws_bf16x3_kernel|convt_fused_bf16x3_kernel|wgrad_bf16x3_ws2_kernel|wgrad_bf16x3_s2_kernel" "$f" > $out/pass$i.csv; else tail -5 $out/log$i.txt; fi
  rm -rf $out/raw$i
done
python3 tools/pmc_mfma.py --parse $out $out/pmc_mfma.md $out/pmc_mfma.json
