#!/bin/bash
# round 6, session 3, call 21: k_gemm_fatd with x staged through registers (global -> VGPR -> ds_write; lab knob 35 = 3) vs the LDS-DMA form (35 = 1)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_mblock.py -m gpu -q -k "direct_weight" > $OUT/r6b21_tests.log 2>&1; echo "bitwise tests exit $?"; tail -3 $OUT/r6b21_tests.log
run() {  # tag, LA_DEBUG, args
  LA_LAB_BUILD=1 LA_DEBUG="$2" timeout 600 python bench.py $3 --steps 24 --warmup 4 --secondary "" --no-cpu-baseline > $OUT/r6b21_$1.log 2>&1
  tail -1 $OUT/r6b21_$1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'))" || tail -5 $OUT/r6b21_$1.log
}
for i in 1 2; do
  run mistral8_dma_$i "35=1" "--model mistral --batch 8"
  run mistral8_xs_$i "35=3" "--model mistral --batch 8"
done
run 7b6_dma "35=1" "--model 7b --batch 6"
run 7b6_xs "35=3" "--model 7b --batch 6"
for arm in 1 3; do
  RAW=/tmp/la_prof_xs$arm; rm -rf $RAW
  ( cd /tmp && LA_LAB_BUILD=1 LA_DEBUG="35=$arm" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o run -- bash -c "cd $REPO && python bench.py --model mistral --batch 8 --steps 12 --warmup 2 --secondary '' --no-cpu-baseline" > $OUT/r6b21_rocprof$arm.log 2>&1 )
  python - <<PY
import csv, glob
rows = []
for f in glob.glob('$RAW/**/*kernel_stats*.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -int(r['TotalDurationNs']))
for r in rows[:40]:
    if 'gemm_fatd' in r['Name']:
        print('arm $arm %-60s calls %6s avg %9.2f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
