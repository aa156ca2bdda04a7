#!/usr/bin/env python3
"""profiles/pmc_latest.json from the summaries scripts/gpu_profile.sh leaves in gpurun_out/ (kernel stats + the three --pmc passes).
Run in the build container after a gpurun call of scripts/gpu_record.sh; argument = the name of the raw text file it belongs to."""
import csv
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, 'gpurun_out')
src = sys.argv[1] if len(sys.argv) > 1 else 'profiles/r03b_profile_raw.txt'
kern = {}
for r in csv.DictReader(open(os.path.join(G, 'run_kernel_stats.csv'))):
    if r['Name'].startswith(('k_', 'void k_')):
        kern[r['Name'][:70]] = {'dispatches': int(r['Calls']), 'avg_us': round(float(r['AverageNs']) / 1e3, 2)}
for c, key in (('FETCH_SIZE', 'fetch_size_kb'), ('WRITE_SIZE', 'write_size_kb')):
    for l in open(os.path.join(G, f'pmc_{c}_summary.txt')):
        name, n, v = l.rstrip('\n').split('\t')
        if name in kern:
            kern[name][key] = float(v)
for l in open(os.path.join(G, 'pmc_SQ_summary.txt')):
    name = l[:54].rstrip()
    hit = [k for k in kern if k.startswith(name)]
    if hit:
        kern[hit[0]]['sq'] = {kv.split('=')[0]: float(kv.split('=')[1]) for kv in l[54:].split()}
for k, d in kern.items():
    if 'fetch_size_kb' in d:
        d['hbm_bytes_per_launch'] = int(2 * d['fetch_size_kb'] * 1024 + d.get('write_size_kb', 0.0) * 1024)
    if 'sq' in d and d['avg_us'] > 0:
        d['mfma_busy_frac'] = round(d['sq'].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (d['avg_us'] * 1e-6 * 2.4e9 * 1024), 4)
out = {
    'source': f'{src} (rocprofv3 --kernel-trace --stats, then separate --pmc passes: FETCH_SIZE | WRITE_SIZE | SQ counters; scripts/gpu_profile.sh, bench.py --steps 10)',
    'correction': 'gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> read bytes = 2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md, HBM); WRITE_SIZE x 1024 uncorrected',
    'mfma_note': 'mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (avg duration x 2.4 GHz x 1024 SIMDs)',
    'kernels': kern,
}
json.dump(out, open(os.path.join(R, 'profiles', 'pmc_latest.json'), 'w'), indent=1)
g = kern.get([k for k in kern if k.startswith('void k_gemm64r<4, 1, 4, 8, 0>')][0])
print('gate/up:', g['avg_us'], 'us', g['hbm_bytes_per_launch'] / 1e6, 'MB', 'mfma busy', g.get('mfma_busy_frac'))
