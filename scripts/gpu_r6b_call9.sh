#!/bin/bash
# round 6, session 3, call 9: k_gemm_fatd<2, SLAB> (knob 35 bit 1) — bitwise tests, then alternating A/B on the lab build (35 = 1 vs 3), kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mblock.py -m gpu -q -x -k "direct_weight" > $OUT/r6b9_tests.log 2>&1; echo "bitwise tests exit $?"; tail -6 $OUT/r6b9_tests.log
run() {  # tag, LA_DEBUG, args
  LA_LAB_BUILD=1 LA_DEBUG="$2" timeout 600 python bench.py $3 --steps 24 --warmup 4 --secondary "" --no-cpu-baseline > $OUT/r6b9_$1.log 2>&1
  tail -1 $OUT/r6b9_$1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'), (d['config'].get('speed_incl_prefill') or {}).get('prefill_ms'))" || tail -5 $OUT/r6b9_$1.log
}
for i in 1 2; do
  run mistral8_gu_$i "35=1" "--model mistral --batch 8"
  run mistral8_gu_slab_$i "35=3" "--model mistral --batch 8"
done
run 7b8_gu "35=1" "--model 7b --batch 8"
run 7b8_gu_slab "35=3" "--model 7b --batch 8"
run 13b8_gu "35=1" "--model 13b --batch 8"
run 13b8_gu_slab "35=3" "--model 13b --batch 8"
for arm in 1 3; do
  RAW=/tmp/la_prof_fats$arm; rm -rf $RAW
  ( cd /tmp && LA_LAB_BUILD=1 LA_DEBUG="35=$arm" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o run -- bash -c "cd $REPO && python bench.py --model mistral --batch 8 --steps 12 --warmup 2 --secondary '' --no-cpu-baseline" > $OUT/r6b9_rocprof$arm.log 2>&1 )
  python - <<PY
import csv, glob
rows = []
for f in glob.glob('$RAW/**/*kernel_stats*.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -int(r['TotalDurationNs']))
for r in rows[:40]:
    if 'gemm_fat' in r['Name'] or 'row_norm_mb' in r['Name']:
        print('arm $arm  %-60s calls %6s avg %9.2f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
