#!/bin/bash
# round 6, session 3, call 13: partner prefetch in k_gemm_fatd (knob 37) — bitwise check, A/B; failure pattern of the k_gemm_fatq bitwise test
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mblock.py -m gpu -q -k "direct_weight_qkv" > $OUT/r6b13_qkv_tests.log 2>&1; echo "qkv tests exit $?"; grep -n "AssertionError: \|passed\|failed" $OUT/r6b13_qkv_tests.log | head -20
timeout 900 python -m pytest tests/test_gpu_mblock.py -m gpu -q -k "direct_weight_gateup" > $OUT/r6b13_gu_tests.log 2>&1; echo "gate/up tests (touch) exit $?"; tail -3 $OUT/r6b13_gu_tests.log
run() {  # tag, LA_DEBUG, args
  LA_LAB_BUILD=1 LA_DEBUG="$2" timeout 600 python bench.py $3 --steps 24 --warmup 4 --secondary "" --no-cpu-baseline > $OUT/r6b13_$1.log 2>&1
  tail -1 $OUT/r6b13_$1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'), (d['config'].get('speed_incl_prefill') or {}).get('prefill_ms'))" || tail -5 $OUT/r6b13_$1.log
}
for i in 1 2; do
  run mistral8_base_$i "36=0,37=0" "--model mistral --batch 8"
  run mistral8_touch_$i "36=0,37=1" "--model mistral --batch 8"
done
run 7b8_base "36=0,37=0" "--model 7b --batch 8"
run 7b8_touch "36=0,37=1" "--model 7b --batch 8"
run 13b8_base "36=0,37=0" "--model 13b --batch 8"
run 13b8_touch "36=0,37=1" "--model 13b --batch 8"
for arm in 0 1; do
  RAW=/tmp/la_prof_touch$arm; rm -rf $RAW
  ( cd /tmp && LA_LAB_BUILD=1 LA_DEBUG="36=0,37=$arm" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o run -- bash -c "cd $REPO && python bench.py --model mistral --batch 8 --steps 12 --warmup 2 --secondary '' --no-cpu-baseline" > $OUT/r6b13_rocprof$arm.log 2>&1 )
  python - <<PY
import csv, glob
rows = []
for f in glob.glob('$RAW/**/*kernel_stats*.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -int(r['TotalDurationNs']))
for r in rows[:40]:
    if 'gemm_fat' in r['Name']:
        print('arm $arm %-60s calls %6s avg %9.2f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
