#!/bin/bash
# round 5, call 6: gate/up as ONE planned region x all token blocks per workgroup in fat waves (la_lab_set(6, 113) vs the paired fat default 49)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mblock.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "paired or tiny or mixtral_blocks" > $OUT/r5c6_pytest.log 2>&1
echo "pytest exit $?" >> $OUT/r5c6_pytest.log
tail -4 $OUT/r5c6_pytest.log | cut -c1-220
LA_LAB_SET="6=113" timeout 900 python -m pytest tests/test_gpu_mblock.py tests/test_gpu_batch.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "not paired and not schedules and not merged" > $OUT/r5c6_pytest_113.log 2>&1
echo "pytest(6=113) exit $?" >> $OUT/r5c6_pytest_113.log
tail -3 $OUT/r5c6_pytest_113.log | cut -c1-220
for shp in "11008 4096" "14336 4096" "13824 5120"; do
  set -- $shp
  MB_F=$1 MB_K=$2 timeout 300 python scripts/gpu_mb_gemm.py time > $OUT/r5c6_gemm_$1.log 2>&1
  echo "== F=$1 K=$2"; grep -E "gate" $OUT/r5c6_gemm_$1.log | cut -c1-120
done
for rep in a b; do
 for cfg in "mistral 8" "13b 4" "mixtral 4" "7b 1"; do
  set -- $cfg
  for v in 49 113; do
    X=""; [ "$1" = "7b" ] && X="--secondary \"\""
    LA_DEBUG="6=$v" timeout 500 bash -c "python bench.py --model $1 --batch $2 --steps 24 --warmup 4 --no-cpu-baseline $X" > $OUT/r5c6_${1}_v${v}_$rep.json 2> $OUT/r5c6_${1}_v${v}_$rep.err
  done
 done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5c6_*.json')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            print(f.split('/')[-1], d['value'], d['ms_per_step'], 'accept', d['config']['mean_accept_len'], 'eq', d['config']['lookahead_equals_greedy'],
                  'prefill_ms', d['config']['speed_incl_prefill']['prefill_ms'])
PY
