#!/bin/bash
# round 5, call 13: the expert plan + gather of the gathered MoE step as ONE launch (default) vs the two launches of round 3 (la_lab_set(16, 4)):
# bitwise test + MoE suites, rocprofv3 kernel stats of the Mixtral bs=4 leg, step A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_moe.py tests/test_gpu_mblock.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "moe or mixtral or expert" > $OUT/r5c13_pytest.log 2>&1
echo "pytest exit $?" >> $OUT/r5c13_pytest.log
tail -6 $OUT/r5c13_pytest.log | cut -c1-220
RAW=/tmp/la_c13; rm -rf $RAW; mkdir -p $RAW
CMD="cd $REPO && BENCH_IS_SECONDARY=1 python bench.py --model mixtral --batch 4 --steps 12 --warmup 2 --no-cpu-baseline --profile-iters 1"
( cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o run -- bash -c "$CMD" > $OUT/r5c13_stats.log 2>&1 )
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/la_c13/stats/**/*kernel_stats*.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r['Name'].startswith(('k_', 'void k_'))]
    rows.sort(key=lambda r: -int(r['TotalDurationNs']))
    with open('gpurun_out/r5c13_kernel_stats_mixtral_b4.txt', 'w') as o:
        for r in rows[:26]:
            line = '%-72s calls %6d avg %8.2f us total %9.2f ms' % (r['Name'][:72], int(r['Calls']), float(r['AverageNs']) / 1e3, int(r['TotalDurationNs']) / 1e6)
            o.write(line + '\n')
            if 'moe' in line or 'router' in line: print(line)
PY
for rep in a b; do
  for v in 0 4; do
    LA_DEBUG="16=$v" timeout 500 python bench.py --model mixtral --batch 4 --steps 32 --warmup 4 --no-cpu-baseline > $OUT/r5c13_mixtral_k16_${v}_$rep.json 2> $OUT/r5c13_mixtral_k16_${v}_$rep.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5c13_*.json')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            print(f.split('/')[-1], d['value'], d['ms_per_step'], 'accept', d['config']['mean_accept_len'], 'eq', d['config']['lookahead_equals_greedy'])
PY
