#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration (tools/ubench/fetch_calib.hip): separate --pmc passes, per-kernel means
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/cal_$c -o p -- $GRAFT_REPO_ROOT/tools/ubench/fetch_calib > /tmp/cal_$c.log 2>&1 || echo "pass $c failed"
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/cal_$c > $OUT/calib_$c.csv
  cat $OUT/calib_$c.csv
done
grep bytes /tmp/cal_FETCH_SIZE.log
