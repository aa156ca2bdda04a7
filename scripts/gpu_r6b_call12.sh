#!/bin/bash
# round 6, session 3, call 12: k_gemm_fatq (knob 36) — bitwise tests, A/B on the lab build (36 = 0 vs 1), kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mblock.py -m gpu -q -x -k "direct_weight or paired_wide" > $OUT/r6b12_tests.log 2>&1; echo "bitwise tests exit $?"; tail -6 $OUT/r6b12_tests.log
run() {  # tag, LA_DEBUG, args
  LA_LAB_BUILD=1 LA_DEBUG="$2" timeout 600 python bench.py $3 --steps 24 --warmup 4 --secondary "" --no-cpu-baseline > $OUT/r6b12_$1.log 2>&1
  tail -1 $OUT/r6b12_$1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'), (d['config'].get('speed_incl_prefill') or {}).get('prefill_ms'))" || tail -5 $OUT/r6b12_$1.log
}
for i in 1 2; do
  run mistral8_off_$i "36=0" "--model mistral --batch 8"
  run mistral8_on_$i "36=1" "--model mistral --batch 8"
done
for i in 1 2; do
  run 13b4_off_$i "36=0" "--model 13b --batch 4"
  run 13b4_on_$i "36=1" "--model 13b --batch 4"
done
run 7b8_off "36=0" "--model 7b --batch 8"
run 7b8_on "36=1" "--model 7b --batch 8"
run 7b4_off "36=0" "--model 7b --batch 4"
run 7b4_on "36=1" "--model 7b --batch 4"
for arm in 0 1; do
  for leg in "mistral 8" "13b 4"; do set -- $leg
  RAW=/tmp/la_prof_fatq$arm$1; rm -rf $RAW
  ( cd /tmp && LA_LAB_BUILD=1 LA_DEBUG="36=$arm" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o run -- bash -c "cd $REPO && python bench.py --model $1 --batch $2 --steps 12 --warmup 2 --secondary '' --no-cpu-baseline" > $OUT/r6b12_rocprof$arm$1.log 2>&1 )
  python - <<PY
import csv, glob
rows = []
for f in glob.glob('$RAW/**/*kernel_stats*.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -int(r['TotalDurationNs']))
for r in rows[:40]:
    if 'gemm_fat' in r['Name']:
        print('arm $arm $1 %-60s calls %6s avg %9.2f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  done
done
