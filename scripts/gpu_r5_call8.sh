#!/bin/bash
# round 5, call 8: slab launches as ONE 64-row region x 256 rows per workgroup at <= 4 blocks (la_lab_set(6, 1393) vs the default 369); the new
# generate(min_new_tokens) GPU test; whole suite at the current defaults
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mblock.py tests/test_gpu_edges.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "paired or min_new" > $OUT/r5c8_pytest.log 2>&1
echo "pytest exit $?" >> $OUT/r5c8_pytest.log
tail -6 $OUT/r5c8_pytest.log | cut -c1-220
LA_LAB_SET="6=1393" timeout 900 python -m pytest tests/test_gpu_mblock.py tests/test_gpu_batch.py tests/test_gpu_moe.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "not paired and not schedules and not merged" > $OUT/r5c8_pytest_1393.log 2>&1
echo "pytest(6=1393) exit $?" >> $OUT/r5c8_pytest_1393.log
tail -3 $OUT/r5c8_pytest_1393.log | cut -c1-220
for rep in a b; do
 for cfg in "13b 4" "mixtral 4"; do
  set -- $cfg
  for v in 369 1393; do
    LA_DEBUG="6=$v" timeout 500 python bench.py --model $1 --batch $2 --steps 24 --warmup 4 --no-cpu-baseline > $OUT/r5c8_${1}_v${v}_$rep.json 2> $OUT/r5c8_${1}_v${v}_$rep.err
  done
 done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5c8_*.json')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            print(f.split('/')[-1], d['value'], d['ms_per_step'], 'accept', d['config']['mean_accept_len'], 'eq', d['config']['lookahead_equals_greedy'],
                  'prefill_ms', d['config']['speed_incl_prefill']['prefill_ms'])
PY
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $OUT/r5c8_pytest_full.log 2>&1
echo "pytest exit $?" >> $OUT/r5c8_pytest_full.log
tail -5 $OUT/r5c8_pytest_full.log | cut -c1-220
