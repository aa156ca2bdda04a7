# -*- coding: utf-8 -*-
"""Turn the summaries scripts/gpu_profile.sh leaves in gpurun_out/ (rocprofv3 --kernel-trace --stats + separate --pmc passes:
FETCH_SIZE | WRITE_SIZE | SQ counters) into the committed profiles/<tag>_profile_raw.txt and profiles/pmc_latest.json (per-kernel
HBM bytes per launch with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md, MFMA busy fractions).

    python scripts/make_pmc_json.py r02
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'gpurun_out')
tag = sys.argv[1] if len(sys.argv) > 1 else 'rXX'


def tsv(path):
    out = {}
    for line in open(path):
        parts = line.rstrip('\n').split('\t')
        if len(parts) == 3:
            out[parts[0].strip()] = (int(parts[1]), float(parts[2]))
    return out


stats = {}
for r in csv.DictReader(open(os.path.join(G, 'run_kernel_stats.csv'))):
    if r['Name'].startswith(('k_', 'void k_')):
        stats[r['Name']] = r
fetch, write = tsv(os.path.join(G, 'pmc_FETCH_SIZE_summary.txt')), tsv(os.path.join(G, 'pmc_WRITE_SIZE_summary.txt'))
sq = {}
for line in open(os.path.join(G, 'pmc_SQ_summary.txt')):
    name, _, rest = line.rstrip('\n').partition('   ')
    vals = dict(kv.split('=') for kv in rest.split() if '=' in kv)
    sq[name.strip()] = {k: float(v) for k, v in vals.items()}


def find(d, name):
    for k, v in d.items():
        if name.startswith(k[:50]) or k.startswith(name[:50]):
            return v
    return None


raw = [f'== rocprofv3 --kernel-trace --stats (bench.py --steps 10 --warmup 2 --no-cpu-baseline), {tag}']
kern = {}
for name, r in sorted(stats.items(), key=lambda kv: -int(kv[1]['TotalDurationNs'])):
    avg = float(r['AverageNs']) / 1e3
    raw.append(f"{name[:70]:72s} calls {r['Calls']:>6s} avg {avg:8.2f} us min {int(r['MinNs']) / 1e3:8.2f} max {int(r['MaxNs']) / 1e3:8.2f} "
               f"total {int(r['TotalDurationNs']) / 1e6:9.2f} ms")
    f, w, s = find(fetch, name), find(write, name), find(sq, name)
    e = {'dispatches': int(r['Calls']), 'avg_us': round(avg, 2)}
    if f:
        e['fetch_size_kb'] = f[1]
    if w:
        e['write_size_kb'] = w[1]
    if f or w:
        e['hbm_bytes_per_launch'] = int(2 * (f[1] if f else 0) * 1024 + (w[1] if w else 0) * 1024)
    if s:
        e['sq'] = s
        if s.get('SQ_VALU_MFMA_BUSY_CYCLES'):
            e['mfma_busy_frac'] = round(s['SQ_VALU_MFMA_BUSY_CYCLES'] / (avg * 1e-6 * 2.4e9 * 1024), 4)
    kern[name] = e
raw.append('== FETCH_SIZE (KB per dispatch, mean; separate --pmc pass)')
raw += [f'{k[:70]:72s} n={v[0]:6d} mean={v[1]:14.1f}' for k, v in fetch.items() if k.startswith(('k_', 'void k_'))]
raw.append('== WRITE_SIZE (KB per dispatch, mean; separate --pmc pass)')
raw += [f'{k[:70]:72s} n={v[0]:6d} mean={v[1]:14.1f}' for k, v in write.items() if k.startswith(('k_', 'void k_'))]
raw.append('== SQ counters (per dispatch, mean; separate --pmc pass)')
raw += [f'{k[:52]:54s} ' + ' '.join(f'{c}={v:.0f}' for c, v in sorted(s.items())) for k, s in sq.items()]
src = f'profiles/{tag}_profile_raw.txt'
open(os.path.join(ROOT, src), 'w').write('\n'.join(raw) + '\n')
json.dump({'source': f'{src} (rocprofv3 --kernel-trace --stats, then separate --pmc passes: FETCH_SIZE | WRITE_SIZE | SQ counters; '
                     'scripts/gpu_profile.sh, bench.py --steps 10)',
           'correction': 'gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> read bytes = 2 x FETCH_SIZE x 1024 '
                         '(MI355X_MICROARCH.md, HBM); WRITE_SIZE x 1024 uncorrected',
           'mfma_note': 'mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (avg duration x 2.4 GHz x 1024 SIMDs)',
           'kernels': kern}, open(os.path.join(ROOT, 'profiles', 'pmc_latest.json'), 'w'), indent=1)
print('wrote', src, 'and profiles/pmc_latest.json;', len(kern), 'kernels')
