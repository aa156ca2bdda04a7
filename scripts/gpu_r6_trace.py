# -*- coding: utf-8 -*-
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name calls / avg / min us, the steady-state half only (the second half
of the dispatches by time: past prefill and graph capture).  python scripts/gpu_r6_trace.py <dir> <out.txt>"""
import collections
import csv
import glob
import os
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    files = glob.glob(os.path.join(d, '**', '*kernel_trace*.csv'), recursive=True)
    lines = []
    for f in files:
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
        rows = rows[len(rows) // 2:]
        agg = collections.defaultdict(list)
        for r in rows:
            agg[r['Kernel_Name'][:64]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        tot = sum(sum(v) for v in agg.values())
        lines.append(f'== {os.path.basename(f)}: steady-state half, {len(rows)} dispatches, {tot / 1e6:.3f} ms in kernels')
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            lines.append(f'{k:66s} n={len(v):6d} avg {sum(v) / len(v) / 1e3:7.2f} us  min {min(v) / 1e3:7.2f}  total {sum(v) / 1e6:8.3f} ms')
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:16]))


if __name__ == '__main__':
    main()
