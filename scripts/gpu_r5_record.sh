#!/bin/bash
# round 5 record run: whole GPU suite + smoke, default bench (cpu_baseline + secondary legs), rocprofv3 stats + PMC passes of the headline
# (scripts/gpu_profile.sh) and of the batch configurations (scripts/gpu_prof_secondary.sh)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 --durations=8 > $OUT/r5_pytest_record.log 2>&1
echo "pytest exit $?" >> $OUT/r5_pytest_record.log
tail -14 $OUT/r5_pytest_record.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r5_smoke.log 2>&1; echo "smoke exit $?" >> $OUT/r5_smoke.log; tail -2 $OUT/r5_smoke.log | cut -c1-300
( time timeout 1500 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
echo "bench exit $?" >> $OUT/bench_default.err; cat $OUT/bench_default.time | tail -3
python - <<'PY'
import json
for l in open('gpurun_out/bench_default.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print({k: d[k] for k in ('value', 'ms_per_step', 'steps', 'dtype')}, 'accept', d['config']['mean_accept_len'], 'roofline', d['roofline']['frac'], 'step', d['roofline']['verify_step']['frac'],
              'cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('ms_per_step'))
        for s in d.get('secondary') or []:
            print('  secondary', str(s.get('workload', s))[:34], s.get('draft_retrieval', '')[:24], s.get('ms_per_step'), s.get('value'), s.get('error'))
PY
STEPS=10 bash scripts/gpu_profile.sh > $OUT/profile.log 2>&1
grep -E "^void k_|^k_" $OUT/profile.log | head -24 | cut -c1-150
STEPS=12 bash scripts/gpu_prof_secondary.sh > $OUT/prof_secondary.log 2>&1
tail -30 $OUT/prof_secondary.log | cut -c1-200
