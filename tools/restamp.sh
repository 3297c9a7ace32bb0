#!/bin/bash
# After `bash tools/profile_round.sh <tag>` came back: copy the round's summaries into profiles/<tag>/ and
# re-stamp profiles/roofline_traffic.json (bench.py's `roofline.traffic`) against the current kernel sources.
TAG=${1:-r6}
cd "$(dirname "$0")/.."
P=gpurun_out/prof_$TAG
mkdir -p profiles/$TAG
cp $P/*.csv $P/*_bench_line.json profiles/$TAG/
[ -f $P/bench_default.json ] && cp $P/bench_default.json profiles/$TAG/
python tools/pmc_traffic.py c2_B16 'conv_gemm_x3_kernel<1, 4, 1, 2, true, 3, 0>|conv_gemm_x3_kernel<1, 4, 1, 2, true, 0, 0>' \
  $P/c2_pmc_FETCH_SIZE.csv $P/c2_pmc_WRITE_SIZE.csv "round 6, profile_round.sh $TAG, final sources; float32x2 gate kernel, 256 x 128 tiles, two workgroups per CU, condition as a K step, weights by LDS-DMA (all 20 gate launches: 19 read the pre-split x, the first an fp32 x); FETCH x2 calibrated on this access pattern (profiles/r3/calib_*.csv)" | grep hbm_traffic
python tools/pmc_traffic.py c4_N1048560 'vq_mfma_x3_kernel<128, false, 2>' \
  $P/c4_pmc_FETCH_SIZE.csv $P/c4_pmc_WRITE_SIZE.csv "round 6, final sources, sweep kernel only (three fp16 products, |w|^2 per tile as 16-byte loads)" | grep hbm_traffic
python tools/pmc_traffic.py c5_bf16_B16 'conv_gemm_x3_kernel<1, 4, 1, 1, true, 3, 0>' \
  $P/c5_bf16_pmc_FETCH_SIZE.csv $P/c5_bf16_pmc_WRITE_SIZE.csv "round 6, final sources (x, z, gates and gh stored as bf16; the 39 gate launches that read a bf16 x)" | grep hbm_traffic
