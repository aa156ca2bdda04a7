#!/bin/bash
# round 5, second final record: whole GPU suite + smoke + the default bench line at HEAD (key 6 = 4465, plan + gather in one launch)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 --durations=5 > $OUT/r5_pytest_final2.log 2>&1
echo "pytest exit $?" >> $OUT/r5_pytest_final2.log
tail -10 $OUT/r5_pytest_final2.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r5_smoke2.log 2>&1; echo "smoke exit $?" >> $OUT/r5_smoke2.log; tail -2 $OUT/r5_smoke2.log | cut -c1-300
( time timeout 1500 python bench.py > $OUT/bench_default2.log 2> $OUT/bench_default2.err ) 2> $OUT/bench_default2.time
echo "bench exit $?" >> $OUT/bench_default2.err; tail -3 $OUT/bench_default2.time
python - <<'PY'
import json
for l in open('gpurun_out/bench_default2.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print({k: d[k] for k in ('value', 'ms_per_step', 'steps', 'dtype')}, 'accept', d['config']['mean_accept_len'], 'roofline', d['roofline']['frac'], 'step', d['roofline']['verify_step']['frac'])
        for s in d['config'].get('secondary', []):
            print('  secondary', str(s.get('workload', s.get('config', '')))[:70], s.get('ms_per_step'), s.get('value'))
PY
