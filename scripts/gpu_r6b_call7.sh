#!/bin/bash
# round 6, session 3, call 7: k_gemm_fatd (lab knob 35: weights straight into MFMA operand registers) — bitwise test, then alternating A/B on the
# Mistral-7B bs=8 and Llama-2-7B bs=8 legs (lab build as the process library), kernel trace of both arms
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_mblock.py -m gpu -q -x -k "direct_weight" > $OUT/r6b7_tests.log 2>&1; echo "bitwise test exit $?"; tail -5 $OUT/r6b7_tests.log
run() {  # tag, LA_DEBUG value, args
  LA_LAB_BUILD=1 LA_DEBUG="$2" timeout 600 python bench.py $3 --steps 24 --warmup 4 --secondary "" --no-cpu-baseline > $OUT/r6b7_$1.log 2>&1
  tail -1 $OUT/r6b7_$1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'))" || tail -5 $OUT/r6b7_$1.log
}
for i in 1 2; do
  run mistral8_base_$i "35=0" "--model mistral --batch 8"
  run mistral8_fatd_$i "35=1" "--model mistral --batch 8"
done
run 7b8_base "35=0" "--model 7b --batch 8"
run 7b8_fatd "35=1" "--model 7b --batch 8"
for arm in 0 1; do
  RAW=/tmp/la_prof_fatd$arm; rm -rf $RAW
  ( cd /tmp && LA_LAB_BUILD=1 LA_DEBUG="35=$arm" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o run -- bash -c "cd $REPO && python bench.py --model mistral --batch 8 --steps 12 --warmup 2 --secondary '' --no-cpu-baseline" > $OUT/r6b7_rocprof$arm.log 2>&1 )
  python - <<PY
import csv, glob
rows = []
for f in glob.glob('$RAW/**/*kernel_stats*.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -int(r['TotalDurationNs']))
for r in rows[:40]:
    if 'gemm_fat' in r['Name']:
        print('arm $arm  %-60s calls %6s avg %9.2f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
