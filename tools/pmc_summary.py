#!/usr/bin/env python
"""Reduces rocprofv3 counter_collection CSVs to a per-kernel mean table.
usage: python tools/pmc_summary.py <dir> > summary.csv"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row['Kernel_Name'].split('(')[0]
                a = acc[name][row['Counter_Name']]
                a[0] += float(row['Counter_Value'])
                a[1] += 1
    counters = sorted({c for k in acc for c in acc[k]})
    w = csv.writer(sys.stdout)
    w.writerow(['kernel', 'dispatches'] + ['mean_' + c for c in counters])
    for k in sorted(acc, key=lambda k: -max(v[1] for v in acc[k].values())):
        n = max(v[1] for v in acc[k].values())
        w.writerow([k, n] + ['%.1f' % (acc[k][c][0] / acc[k][c][1]) if acc[k][c][1] else '' for c in counters])


if __name__ == '__main__':
    main()
