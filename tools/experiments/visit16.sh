cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for args in "--workload c5 --bf16" "--workload c5"; do
timeout 300 python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('$args: ms/step %.3f | gate kernel avg %.1f us  loss %s' % (d['ms_per_step'], 1e3 * r['avg_launch_ms'], d['losses_last_step']))"
done
bash tools/pmc_run.sh "conv_gemm_x3_kernel<[012], [24], 1, 2, true" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" 2>&1 | tee gpurun_out/pmc_gate_now.txt
