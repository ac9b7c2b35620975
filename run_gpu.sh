cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc1 -o p -- python3 $R/tools/kbench.py --only "conv3x3 s1 128->128" --reps 3 > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc2 -o p -- python3 $R/tools/kbench.py --only "conv3x3 s1 128->128" --reps 3 > $R/gpurun_out/pmc2.log 2>&1
ls -R $R/gpurun_out/pmc1 | head; tail -3 $R/gpurun_out/pmc1.log
