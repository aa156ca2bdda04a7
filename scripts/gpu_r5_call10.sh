#!/bin/bash
# round 5, call 10: QKV at <= 4 blocks over <= 128 regions as one region x 128 rows per workgroup (la_lab_set(6, 369 + 4096)) vs the default
# (one region x 256 rows): bitwise tests, Mixtral bs=4 and Mistral bs=4 steps A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_mblock.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "paired" > $OUT/r5c10_pytest.log 2>&1
echo "pytest exit $?" >> $OUT/r5c10_pytest.log
tail -6 $OUT/r5c10_pytest.log | cut -c1-220
for rep in a b; do
 for cfg in "mixtral 4" "mistral 4"; do
  if [ "$rep$cfg" = "bmistral 4" ]; then continue; fi
  set -- $cfg
  for v in 369 4465; do
    LA_DEBUG="6=$v" timeout 500 python bench.py --model $1 --batch $2 --steps 24 --warmup 4 --no-cpu-baseline > $OUT/r5c10_${1}_v${v}_$rep.json 2> $OUT/r5c10_${1}_v${v}_$rep.err
  done
 done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5c10_*.json')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            print(f.split('/')[-1], d['value'], d['ms_per_step'], 'accept', d['config']['mean_accept_len'], 'eq', d['config']['lookahead_equals_greedy'],
                  'prefill_ms', d['config']['speed_incl_prefill']['prefill_ms'])
PY
LA_LAB_SET="6=4465" timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_moe.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "not paired and not schedules and not merged" > $OUT/r5c10_pytest_4465.log 2>&1
echo "pytest(6=4465) exit $?" >> $OUT/r5c10_pytest_4465.log
tail -3 $OUT/r5c10_pytest_4465.log | cut -c1-220
