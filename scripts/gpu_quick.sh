#!/bin/bash
# quick iteration: gpu tests (no full-size), short bench, kernel-trace stats of our kernels only
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -k "not full_size" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps ${STEPS:-32} --warmup 4 --no-cpu-baseline > gpurun_out/bench.log 2> gpurun_out/bench.err
python - <<'PY'
import json
for l in open('gpurun_out/bench.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print('BENCH value', d['value'], 'ms/step', d['ms_per_step'], 'accept', d['config']['mean_accept_len'], 'T', d['config']['mean_draft_len'],
              'eq_greedy', d['config']['lookahead_equals_greedy'], 'roofline frac', d['roofline']['frac'], 'step frac', d['roofline']['verify_step']['frac'])
        print('  events ms by class', d['roofline']['verify_step']['ms_by_class_events'])
PY
rm -rf /tmp/la_prof; mkdir -p /tmp/la_prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/la_prof/stats -o run -- bash -c "cd $REPO && python bench.py --steps 8 --warmup 2 --no-cpu-baseline --profile-iters 1" > $REPO/gpurun_out/prof_stats.log 2>&1 )
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/la_prof/stats/**/*kernel_stats*.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r['Name'].startswith(('k_', 'void k_'))]
    with open('gpurun_out/kernel_stats_ours.csv', 'w') as fo:
        fo.write('name,calls,avg_us,min_us,max_us,total_ms\n')
        for r in rows:
            fo.write(f"\"{r['Name'][:48]}\",{r['Calls']},{float(r['AverageNs'])/1e3:.2f},{int(r['MinNs'])/1e3:.2f},{int(r['MaxNs'])/1e3:.2f},{int(r['TotalDurationNs'])/1e6:.2f}\n")
            print(f"{r['Name'][:48]:50s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.2f} us  min {int(r['MinNs'])/1e3:7.2f}  total {int(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
