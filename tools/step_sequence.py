"""Reads a rocprofv3 kernel trace of bench.py (..._kernel_trace.csv) and prints the launch sequence of the LAST profiled step,
run-length encoded (consecutive repeats of one pattern collapsed), with each launch's duration and the gap in front of it.
usage: python tools/step_sequence.py <kernel_trace.csv> [min_gap_us]"""
import csv
import re
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda t: t[0])
adam = [i for i, k in enumerate(ks) if 'adam_kernel' in k[2]]
step = ks[adam[-2] + 1:adam[-1] + 1]


def short(n):
    n = n.split('(')[0].replace('void ', '').replace('vq::', '')
    n = re.sub(r'conv_gemm_x3_kernel', 'x3', n)
    return n[:44]


print('%d launches, %.3f ms wall, %.3f ms of kernels' % (len(step), (step[-1][1] - step[0][0]) / 1e6, sum(e - s for s, e, _ in step) / 1e6))
prev_end = step[0][0]
out = []
for s, e, n in step:
    out.append((short(n), (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
i = 0
while i < len(out):
    # collapse period-p repeats (p = 1..6)
    best = (1, 1)
    for p in range(1, 7):
        reps = 1
        while i + (reps + 1) * p <= len(out) and [o[0] for o in out[i + reps * p:i + (reps + 1) * p]] == [o[0] for o in out[i:i + p]]:
            reps += 1
        if reps > 1 and reps * p > best[0] * best[1]:
            best = (p, reps)
    p, reps = best
    if reps > 1:
        print('  x%d {' % reps)
        for j in range(p):
            durs = [out[i + r * p + j][1] for r in range(reps)]
            gaps = [out[i + r * p + j][2] for r in range(reps)]
            print('      %-46s avg %7.1f us   gap before %5.1f us' % (out[i + j][0], sum(durs) / reps, sum(gaps) / reps))
        print('  }')
        i += p * reps
    else:
        print('  %-50s %7.1f us   gap before %5.1f us' % out[i])
        i += 1
