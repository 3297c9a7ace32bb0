#!/bin/bash
# PMC comparison of the conv_loop variants (separate passes; counters only, no tracing)
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/tools/ubench/conv_loop
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1)); rm -rf /tmp/pl$i
  timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pl$i -o p -- $B 64 > /tmp/pl$i.log 2>&1 || echo "pass $i failed: $(tail -2 /tmp/pl$i.log)"
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pl$i | cut -c1-400
done
