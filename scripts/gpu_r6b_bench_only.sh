#!/bin/bash
# the driver's default bench line once more on whatever box this call lands on (box-to-box variance record)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=${TAG:-box}
python bench.py > $OUT/${TAG}_bench_default.log 2>&1; echo "bench exit $?"
tail -1 $OUT/${TAG}_bench_default.log > $OUT/${TAG}_bench_default_record.json
python - <<PY
import json
d = json.load(open('$OUT/${TAG}_bench_default_record.json'))
print('HEADLINE', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['config'].get('speed_incl_prefill', {}).get('prefill_ms'))
for leg in d.get('secondary', []):
    print('LEG', leg.get('name'), leg.get('ms_per_step'), leg.get('value'), leg.get('equals_greedy'))
PY
