#!/bin/bash
# round 6, session 3, call 24: x DIRECT in the one-row-group fat forms (k_gemm_fat STG = 2, lab knob 36: bit 0 slab, 1 QKV, 2 one-region gate/up) — bitwise
# tests, then alternating A/B per bit and together on Mistral bs=8, 13B bs=4, Mixtral bs=4, 7B bs=8; kernel traces
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mblock.py -m gpu -q -k "x_direct" > $OUT/r6b24_tests.log 2>&1; echo "bitwise tests exit $?"; grep -n "AssertionError: \|passed\|failed" $OUT/r6b24_tests.log | head
run() {  # tag, LA_DEBUG, args
  LA_LAB_BUILD=1 LA_DEBUG="$2" timeout 600 python bench.py $3 --steps 24 --warmup 4 --secondary "" --no-cpu-baseline > $OUT/r6b24_$1.log 2>&1
  tail -1 $OUT/r6b24_$1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'))" || tail -5 $OUT/r6b24_$1.log
}
run mistral8_x0_1 "36=0" "--model mistral --batch 8"
run mistral8_x1 "36=1" "--model mistral --batch 8"
run mistral8_x2 "36=2" "--model mistral --batch 8"
run mistral8_x7 "36=7" "--model mistral --batch 8"
run mistral8_x0_2 "36=0" "--model mistral --batch 8"
run 13b4_x0_1 "36=0" "--model 13b --batch 4"
run 13b4_x1 "36=1" "--model 13b --batch 4"
run 13b4_x2 "36=2" "--model 13b --batch 4"
run 13b4_x4 "36=4" "--model 13b --batch 4"
run 13b4_x7 "36=7" "--model 13b --batch 4"
run 13b4_x0_2 "36=0" "--model 13b --batch 4"
run mixtral4_x0 "36=0" "--model mixtral --batch 4"
run mixtral4_x7 "36=7" "--model mixtral --batch 4"
for arm in 0 7; do
  for leg in "mistral 8" "13b 4"; do set -- $leg
  RAW=/tmp/la_prof_fatx$arm$1; rm -rf $RAW
  ( cd /tmp && LA_LAB_BUILD=1 LA_DEBUG="36=$arm" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o run -- bash -c "cd $REPO && python bench.py --model $1 --batch $2 --steps 12 --warmup 2 --secondary '' --no-cpu-baseline" > $OUT/r6b24_rocprof$arm$1.log 2>&1 )
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
