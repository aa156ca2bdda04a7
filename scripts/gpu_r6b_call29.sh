#!/bin/bash
# round 6, session 3, call 29: down_proj at 256 rows with 6 K splits x all token tiles per workgroup (x direct) instead of 3 splits x token quarters (13B bs=4),
# and 8 splits for 7B bs=4 — lab knob 36 bit 1 + --gemm-cfg down_ks
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
run() {  # tag, LA_DEBUG, args
  LA_LAB_BUILD=1 LA_DEBUG="$2" timeout 600 python bench.py $3 --steps 24 --warmup 4 --secondary "" --no-cpu-baseline > $OUT/r6b29_$1.log 2>&1
  tail -1 $OUT/r6b29_$1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'))" || tail -5 $OUT/r6b29_$1.log
}
for i in 1 2; do
  run 13b4_base_$i "36=1" "--model 13b --batch 4"
  run 13b4_ks6q_$i "36=1" "--model 13b --batch 4 --gemm-cfg 0,0,0,0,0,6"
  run 13b4_ks6_$i "36=3" "--model 13b --batch 4 --gemm-cfg 0,0,0,0,0,6"
done
run 7b4_base "36=1" "--model 7b --batch 4"
run 7b4_ks8 "36=3" "--model 7b --batch 4 --gemm-cfg 0,0,0,0,0,8"
for arm in "36=1|" "36=3|--gemm-cfg 0,0,0,0,0,6"; do
  DBG="${arm%%|*}"; EXTRA="${arm##*|}"
  RAW=/tmp/la_prof_ks$DBG; rm -rf "$RAW"
  ( cd /tmp && LA_LAB_BUILD=1 LA_DEBUG="$DBG" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW" -o run -- bash -c "cd $REPO && python bench.py --model 13b --batch 4 --steps 12 --warmup 2 --secondary '' --no-cpu-baseline $EXTRA" > $OUT/r6b29_rocprof_$DBG.log 2>&1 )
  python - <<PY
import csv, glob
rows = []
for f in glob.glob('$RAW/**/*kernel_stats*.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -int(r['TotalDurationNs']))
for r in rows[:40]:
    if 'gemm_fat' in r['Name'] or 'row_norm_mb' in r['Name']:
        print('arm $DBG %-60s calls %6s avg %9.2f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
