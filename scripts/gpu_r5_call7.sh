#!/bin/bash
# round 5, call 7: QKV as ONE {lo, hi} region x 256 rows per workgroup in fat waves of 2 x 2 tiles (la_lab_set(6, 369) vs the default 113); smoke with
# the multi-block error printed
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mblock.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "paired" > $OUT/r5c7_pytest.log 2>&1
echo "pytest exit $?" >> $OUT/r5c7_pytest.log
tail -4 $OUT/r5c7_pytest.log | cut -c1-220
LA_LAB_SET="6=369" timeout 900 python -m pytest tests/test_gpu_mblock.py tests/test_gpu_batch.py tests/test_gpu_moe.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "not paired and not schedules and not merged" > $OUT/r5c7_pytest_369.log 2>&1
echo "pytest(6=369) exit $?" >> $OUT/r5c7_pytest_369.log
tail -3 $OUT/r5c7_pytest_369.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r5c7_smoke.log 2>&1; echo "smoke exit $?" >> $OUT/r5c7_smoke.log; tail -2 $OUT/r5c7_smoke.log | cut -c1-300
for rep in a b; do
 for cfg in "mistral 8" "13b 4" "mixtral 4"; do
  set -- $cfg
  for v in 113 369; do
    LA_DEBUG="6=$v" timeout 500 python bench.py --model $1 --batch $2 --steps 24 --warmup 4 --no-cpu-baseline > $OUT/r5c7_${1}_v${v}_$rep.json 2> $OUT/r5c7_${1}_v${v}_$rep.err
  done
 done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5c7_*.json')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            print(f.split('/')[-1], d['value'], d['ms_per_step'], 'accept', d['config']['mean_accept_len'], 'eq', d['config']['lookahead_equals_greedy'],
                  'prefill_ms', d['config']['speed_incl_prefill']['prefill_ms'])
PY
