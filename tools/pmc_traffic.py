#!/usr/bin/env python
"""Folds rocprofv3 --pmc summaries (tools/pmc_summary.py output of SEPARATE FETCH_SIZE and
WRITE_SIZE passes of one bench.py command) into profiles/roofline_traffic.json, the file
bench.py reads `roofline.traffic` from.  Each entry is stamped with the hash of the kernel
sources it was measured on; bench.py reports null once the sources change.

usage: python tools/pmc_traffic.py <key> <kernel-substring> <fetch_summary.csv> <write_summary.csv> [note]

HBM bytes per launch = 2 x FETCH_SIZE(KiB) x 1024 + WRITE_SIZE(KiB) x 1024: on gfx950 FETCH_SIZE
tallies a wide coalesced read at half its bytes (/opt/skills/guides/MI355X_MICROARCH.md, HBM)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mean_of(path, pat, col):
    """Dispatch-weighted mean of `col` over the kernels matching any of the '|'-separated patterns."""
    pats = pat.split('|')
    tot, n = 0.0, 0
    with open(path) as f:
        for row in csv.DictReader(f):
            if any(p in row['kernel'] for p in pats):
                d = int(row['dispatches'])
                tot += float(row[col]) * d
                n += d
    if n == 0:
        raise SystemExit('%s: no kernel matching %r' % (path, pat))
    return tot / n, n


def main():
    key, pat, fetch_csv, write_csv = sys.argv[1:5]
    note = sys.argv[5] if len(sys.argv) > 5 else ''
    import bench
    fetch_kib, n1 = mean_of(fetch_csv, pat, 'mean_FETCH_SIZE')
    write_kib, n2 = mean_of(write_csv, pat, 'mean_WRITE_SIZE')
    out = os.path.join(ROOT, 'profiles', 'roofline_traffic.json')
    try:
        with open(out) as f:
            db = json.load(f)
    except Exception:
        db = {}
    db[key] = {
        'kernel': pat,
        'kernel_source_sha256_16': bench.kernel_source_hash(),
        'FETCH_SIZE_KiB_raw_mean': fetch_kib, 'WRITE_SIZE_KiB_mean': write_kib,
        'dispatches_averaged': [n1, n2],
        'hbm_traffic_bytes_per_launch': int(2 * fetch_kib * 1024 + write_kib * 1024),
        'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_run.sh), '
                  'FETCH_SIZE x2 per MI355X_MICROARCH.md, per-launch mean; ' + note,
    }
    with open(out, 'w') as f:
        json.dump(db, f, indent=1, sort_keys=True)
    print(json.dumps(db[key], indent=1))


if __name__ == '__main__':
    main()
