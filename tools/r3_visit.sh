#!/bin/bash
# round-3 GPU visit helper: bash tools/r3_visit.sh <step> ...   (steps run in order; output under gpurun_out/r3/)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3
bench_line() {   # label, env..., -- bench args
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3/bench_$label.json 2> gpurun_out/r3/bench_$label.err
  python - "$label" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open('gpurun_out/r3/bench_%s.json' % sys.argv[1]) if l.startswith('{')][-1]); r = d['roofline']
    print('bench[%s]: ms/step %.3f  %.4g samples/s | gate kernel %.1f TF frac %.3f avg %.1f us' % (sys.argv[1], d['ms_per_step'], d['value'], r['achieved'], r['frac'], 1e3 * r['avg_launch_ms']))
except Exception as e:
    print('bench[%s] failed: %r' % (sys.argv[1], e)); print(open('gpurun_out/r3/bench_%s.err' % sys.argv[1]).read()[-2000:])
PY
}
kstats() {       # label, env...
  local label=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$label && env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$label -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/b_$label.log 2>&1 )
  f=$(find /tmp/prof_$label -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] || { echo "no stats for $label"; tail -5 /tmp/b_$label.log; return; }
  cp $f gpurun_out/r3/kstats_$label.csv
  python - "$f" "$label" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('[%s] sum of kernel time per step: %.2f ms, launches/step %.0f' % (sys.argv[2], tot / 1e6 / 13, sum(int(r['Calls']) for r in rows) / 13))
for r in rows[:14]:
    print('  %-66s n/step %5.1f avg %8.1f us  ms/step %6.2f' % (r['Name'][:66], int(r['Calls']) / 13, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6 / 13))
PY
}
for step in "$@"; do
  case $step in
    newtests) timeout 1500 python -m pytest tests/test_gpu_bench_shapes.py "tests/test_gpu_kernels.py::test_vq_near_ties_around_the_certainty_band" "tests/test_gpu_kernels.py::test_conv1d_fwd_bwd" tests/test_gpu_kernels.py::test_vq_golden tests/test_gpu_kernels.py::test_vq_golden_stress -m gpu -q -x --no-header -p no:cacheprovider -s 2>&1 | tail -15 ;;
    alltests) timeout 2400 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -8 ;;
    lin128) for v in 0 32 64; do kstats lin$v VQVAE_LIN128=$v; done ;;
    quick) timeout 900 python -m pytest tests/test_gpu_bench_shapes.py::test_resstack_b16_vs_oracle tests/test_gpu_kernels.py::test_condition_assemble tests/test_gpu_kernels.py::test_upsample_constant_known_answer "tests/test_gpu_kernels.py::test_conv1d_fwd_bwd" tests/test_gpu_kernels.py::test_resblock_fwd_bwd -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -5 ;;
    bench) bench_line default A=1 ;;
    convtests) timeout 1200 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -5 ;;
    w3nc) kstats w3nc2 VQVAE_W3_NC=2 ;;
    modeltests) timeout 1500 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_model.py tests/test_gpu_configs.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -5 ;;
    wintests) timeout 1500 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_fullsize.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -5 ;;
    winab) kstats win1 VQVAE_X3_WIN=1; kstats win0 VQVAE_X3_WIN=0; kstats win1b VQVAE_X3_WIN=1; kstats win0b VQVAE_X3_WIN=0 ;;
    occ) VQVAE_X3_NB=3 python tools/occ_scaling.py 2>&1 | grep '^B' ;;
    pp) timeout 900 python -m pytest tests/test_gpu_bench_shapes.py::test_resblock_b16_vs_oracle tests/test_gpu_kernels.py::test_conv1d_fwd_bwd tests/test_gpu_kernels.py::test_resblock_fwd_bwd -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -5; VQVAE_X3_NB=3 python tools/occ_scaling.py 2>&1 | grep '^B'; kstats pp1 VQVAE_X3_PP=1; kstats pp0 VQVAE_X3_PP=0 ;;
    kstats) kstats default A=1 ;;
    *) echo "unknown step $step" ;;
  esac
done
