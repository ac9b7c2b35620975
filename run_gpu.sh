#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/sq1 gpurun_out/sq2 gpurun_out/sq3
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d gpurun_out/sq1 -o a --output-format csv -- python3 tools/one_conv.py 4 256 256 128 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d gpurun_out/sq2 -o b --output-format csv -- python3 tools/one_conv.py 4 256 256 128 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_CVT --kernel-trace -d gpurun_out/sq3 -o c --output-format csv -- python3 tools/one_conv.py 4 256 256 128 > /dev/null 2>&1
for d in sq1 sq2 sq3; do python3 tools/pmc_summary.py gpurun_out/$d conv_bf16x3_kernel; done
