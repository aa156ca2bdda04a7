#!/bin/bash
# round 6, session 3, call 1: lane-parallel router tail — MoE parity tests, the Mixtral bs=4 leg twice, its kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_moe.py tests/test_gpu_mblock.py -m gpu -q -x > $OUT/r6b1_moe_tests.log 2>&1; echo "moe tests exit $?"; tail -3 $OUT/r6b1_moe_tests.log
for i in 1 2; do
  python bench.py --model mixtral --batch 4 --steps 24 --warmup 4 --secondary "" --no-cpu-baseline > $OUT/r6b1_mixtral_$i.log 2>&1
  tail -1 $OUT/r6b1_mixtral_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mixtral:4', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'))"
done
RAW=/tmp/la_prof_mx; rm -rf $RAW
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o run -- bash -c "cd $REPO && python bench.py --model mixtral --batch 4 --steps 24 --warmup 4 --secondary '' --no-cpu-baseline" > $OUT/r6b1_rocprof.log 2>&1 )
python - <<PY
import csv, glob
rows = []
for f in glob.glob('$RAW/**/*kernel_stats*.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -int(r['TotalDurationNs']))
with open('$OUT/r6b1_kernel_stats_mixtral_b4.txt', 'w') as fo:
    for r in rows[:40]:
        line = '%-80s calls %7s avg %9.2f us total %10.2f ms' % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3, int(r['TotalDurationNs']) / 1e6)
        fo.write(line + '\n')
        if 'norm' in line or 'moe' in line: print(line)
PY
