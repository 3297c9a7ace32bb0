#!/bin/bash
# One GPU visit that produces every profile artefact of the round under gpurun_out/prof_<tag>/:
#   kernel-trace stats of bench.py for c2 (default), c4, c5 --bf16, and separate --pmc passes
#   (FETCH_SIZE / WRITE_SIZE / MFMA busy / instruction mix) for the dominant kernel of each.
# usage: bash tools/profile_round.sh <tag>
TAG=${1:-r5}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_stats() {   # name, bench args...
  local name=$1; shift
  rm -rf /tmp/ps_$name
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$name -o s -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/${name}_bench.log 2>&1
  f=$(find /tmp/ps_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv || echo "no stats for $name"
  grep "^{" $OUT/${name}_bench.log | tail -1 > $OUT/${name}_bench_line.json
}
run_pmc() {     # name, counters, bench args...
  local name=$1; local ctr=$2; shift; shift
  rm -rf /tmp/pp_$name
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pp_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > /tmp/pp_$name.log 2>&1 || echo "pmc pass $name failed"
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pp_$name > $OUT/${name}.csv
}
# (--no-graph: eager launches, so that bench.py's own per-launch events and rocprofv3 time the same dispatches;
#  --no-fresh-input: one timed region = `steps + warmup` steps in the stats)
run_stats c2 --steps 10 --warmup 3 --no-graph --no-fresh-input
run_stats c2_graph --steps 10 --warmup 3 --no-fresh-input
run_stats c2_x3 --steps 10 --warmup 3 --no-graph --no-fresh-input --matmul float32x3
run_stats c2_fp32mfma --steps 10 --warmup 3 --no-graph --no-fresh-input --matmul float32
run_stats c4 --workload c4 --steps 5 --warmup 2
run_stats c4_N1920 --workload c4 --vq-rows 1920 --steps 20 --warmup 3
run_stats c5_bf16 --workload c5 --bf16 --steps 5 --warmup 2
run_stats c5_fp32 --workload c5 --steps 5 --warmup 2
run_pmc c2_pmc_FETCH_SIZE FETCH_SIZE --steps 2 --warmup 1 --no-graph --no-fresh-input
run_pmc c2_pmc_WRITE_SIZE WRITE_SIZE --steps 2 --warmup 1 --no-graph --no-fresh-input
run_pmc c2_pmc_MFMA "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" --steps 2 --warmup 1 --no-graph --no-fresh-input
run_pmc c2_pmc_INSTS "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" --steps 2 --warmup 1 --no-graph --no-fresh-input
run_pmc c4_pmc_FETCH_SIZE FETCH_SIZE --workload c4 --steps 2 --warmup 1
run_pmc c4_pmc_WRITE_SIZE WRITE_SIZE --workload c4 --steps 2 --warmup 1
run_pmc c4_pmc_MFMA "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --workload c4 --steps 2 --warmup 1
run_pmc c5_bf16_pmc_FETCH_SIZE FETCH_SIZE --workload c5 --bf16 --steps 2 --warmup 1
run_pmc c5_bf16_pmc_WRITE_SIZE WRITE_SIZE --workload c5 --bf16 --steps 2 --warmup 1
run_pmc c5_bf16_pmc_MFMA "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --workload c5 --bf16 --steps 2 --warmup 1
ls $OUT | head -60
for n in c2 c4 c5_bf16; do echo "== $n"; head -6 $OUT/${n}_kernel_stats.csv | cut -c1-150; done
