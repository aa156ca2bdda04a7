#!/bin/bash
# round 5, fourth final record: key 6 default 12657 (XCD K-split mapping of the fat slab launches on, for grids of whole multiples of 256 workgroups):
# whole GPU suite + smoke, Mistral bs=8 A/B against 4465
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $OUT/r5_pytest_final4.log 2>&1
echo "pytest exit $?" >> $OUT/r5_pytest_final4.log
tail -4 $OUT/r5_pytest_final4.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r5_smoke4.log 2>&1; echo "smoke exit $?" >> $OUT/r5_smoke4.log; tail -2 $OUT/r5_smoke4.log | cut -c1-300
for rep in a b; do
  for v in 12657 4465; do
    LA_DEBUG="6=$v" timeout 300 python bench.py --model mistral --batch 8 --steps 32 --warmup 4 --no-cpu-baseline > $OUT/r5f4_mistral8_v${v}_$rep.json 2> $OUT/r5f4_mistral8_v${v}_$rep.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5f4_*.json')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            print(f.split('/')[-1], d['value'], d['ms_per_step'], 'accept', d['config']['mean_accept_len'], 'eq', d['config']['lookahead_equals_greedy'],
                  'prefill_ms', d['config']['speed_incl_prefill']['prefill_ms'])
PY
