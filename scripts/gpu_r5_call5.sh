#!/bin/bash
# round 5, call 5: merged-expert launches with two weight regions per workgroup (la_lab_set(25, 4 / 8 / 12)) — agreement test, Mixtral
# layer-shape parity under the new forms, step A/B at Mixtral bs=4 (25 = 1: round-4 default; 5 = paired gate/up; 9 = paired down; 12 = both)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mblock.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "merged_expert or mixtral or moe" -s > $OUT/r5c5_pytest.log 2>&1
echo "pytest exit $?" >> $OUT/r5c5_pytest.log
grep -E "passed|failed|Error|assert|Mixtral-8x7B" $OUT/r5c5_pytest.log | cut -c1-220 | tail -8
LA_LAB_SET="25=12" timeout 900 python -m pytest tests/test_gpu_mblock.py tests/test_gpu_moe.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "mixtral or moe" -s > $OUT/r5c5_pytest_25_12.log 2>&1
echo "pytest(25=12) exit $?" >> $OUT/r5c5_pytest_25_12.log
grep -E "passed|failed|Error|assert|Mixtral-8x7B" $OUT/r5c5_pytest_25_12.log | cut -c1-220 | tail -6
for rep in a b; do
  for v in 1 4 8 12; do
    LA_DEBUG="25=$v" timeout 500 python bench.py --model mixtral --batch 4 --steps 24 --warmup 4 --no-cpu-baseline > $OUT/r5c5_mixtral_v${v}_$rep.json 2> $OUT/r5c5_mixtral_v${v}_$rep.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5c5_*.json')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            print(f.split('/')[-1], d['value'], d['ms_per_step'], 'accept', d['config']['mean_accept_len'], 'eq', d['config']['lookahead_equals_greedy'],
                  'prefill_ms', d['config']['speed_incl_prefill']['prefill_ms'])
PY
tail -3 $OUT/r5c5_mixtral_v12_a.err | cut -c1-300
