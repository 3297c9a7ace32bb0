cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]+|GRBM_[A-Z_]+|TCP_[A-Z0-9_]+|TA_[A-Z0-9_]+)\b" | sort -u > gpurun_out/counters.txt; wc -l gpurun_out/counters.txt
bash tools/pmc_run.sh "conv_gemm_x3_kernel<[012], [24], 1, 2, true" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" \
  2>&1 | tee gpurun_out/pmc_gate.txt
cp /tmp/pmc1.csv gpurun_out/pmc1.csv; cp /tmp/pmc2.csv gpurun_out/pmc2.csv; cp /tmp/pmc3.csv gpurun_out/pmc3.csv
