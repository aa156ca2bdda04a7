#!/bin/bash
# round 5 record run 2 (after the paired expert launches became default): whole GPU suite + smoke, default bench, traffic / kernel stats of the batch configurations
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 --durations=5 > $OUT/r5_pytest_record2.log 2>&1
echo "pytest exit $?" >> $OUT/r5_pytest_record2.log
tail -12 $OUT/r5_pytest_record2.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r5_smoke.log 2>&1; echo "smoke exit $?" >> $OUT/r5_smoke.log; tail -2 $OUT/r5_smoke.log | cut -c1-300
( time timeout 1500 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
echo "bench exit $?" >> $OUT/bench_default.err; tail -3 $OUT/bench_default.time
python - <<'PY'
import json
for l in open('gpurun_out/bench_default.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print({k: d[k] for k in ('value', 'ms_per_step', 'steps', 'dtype')}, 'accept', d['config']['mean_accept_len'], 'roofline', d['roofline']['frac'], 'step', d['roofline']['verify_step']['frac'],
              'floor', d['roofline']['verify_step']['floor_model'].get('frac_of_peak_at_floor'), 'cpu', d['cpu_baseline']['value'])
        for s in d.get('secondary') or []:
            print('  secondary', str(s.get('workload', s))[:34], str(s.get('draft_retrieval', ''))[:24], s.get('ms_per_step'), s.get('value'), s.get('error'))
PY
STEPS=12 bash scripts/gpu_prof_secondary.sh > $OUT/prof_secondary.log 2>&1
grep -E "HBM bytes per step" $OUT/prof_secondary.log
head -8 $OUT/kernel_stats_mixtral_b4.txt | cut -c1-200
