#!/bin/bash
# round 6 record run: the full GPU suite on the product libraries, smoke(), the driver's default bench line, and the rocprofv3 summary of the same command
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
TAG=${TAG:-r06}
timeout 2400 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest_gpu_record.log 2>&1; echo "suite exit $?"; tail -3 $OUT/${TAG}_pytest_gpu_record.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1; echo "smoke exit $?"; tail -1 $OUT/${TAG}_smoke.log
( time python bench.py > $OUT/${TAG}_bench_default.log 2>&1 ) 2> $OUT/${TAG}_bench_default.time; echo "bench exit $?"; tail -2 $OUT/${TAG}_bench_default.time
tail -1 $OUT/${TAG}_bench_default.log > $OUT/${TAG}_bench_default_record.json
python - <<PY
import json
d = json.load(open('$OUT/${TAG}_bench_default_record.json'))
print('HEADLINE', d['value'], d['unit'], d['ms_per_step'], 'ms/step', 'roofline', d['roofline'].get('frac'), 'cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('kind'))
for leg in d.get('secondary', []):
    print('LEG', {k: leg.get(k) for k in ('name', 'ms_per_step', 'value', 'accept_len', 'bound', 'frac', 'traffic_ratio', 'draft_retrieval', 'equals_greedy', 'error')})
print('line bytes', len(open('$OUT/${TAG}_bench_default_record.json').read()))
PY
if [ -n "${PROFILE:-}" ]; then
  RAW=/tmp/la_prof_$TAG; rm -rf $RAW
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o run -- bash -c "cd $REPO && python bench.py --secondary '' --no-cpu-baseline" > $OUT/${TAG}_rocprof_bench.log 2>&1 )
  python - <<PY
import csv, glob
rows = []
for f in glob.glob('$RAW/**/*kernel_stats*.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -int(r['TotalDurationNs']))
with open('$OUT/${TAG}_profile_kernel_stats.txt', 'w') as fo:
    fo.write('# rocprofv3 --kernel-trace --stats -- python bench.py --secondary "" --no-cpu-baseline (HEAD, product library)\n')
    for r in rows[:25]:
        line = '%-90s calls %7s avg %9.2f us total %10.2f ms  %5.1f %%' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, int(r['TotalDurationNs']) / 1e6, float(r['Percentage']))
        print(line); fo.write(line + '\n')
PY
fi
