#!/bin/bash
# round 5 final record: whole GPU suite + smoke at HEAD, default bench (cpu_baseline + five secondary legs), rocprofv3 stats + PMC passes of the
# headline and of the batch configurations, then two sanity legs (fp16 headline, Mistral --batch 16 as two passes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 --durations=5 > $OUT/r5_pytest_final.log 2>&1
echo "pytest exit $?" >> $OUT/r5_pytest_final.log
tail -10 $OUT/r5_pytest_final.log | cut -c1-220
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
STEPS=10 bash scripts/gpu_profile.sh > $OUT/profile.log 2>&1
grep -E "^void k_|^k_" $OUT/profile.log | head -12 | cut -c1-150
STEPS=12 bash scripts/gpu_prof_secondary.sh > $OUT/prof_secondary.log 2>&1
grep -E "HBM bytes per step" $OUT/prof_secondary.log
timeout 400 python bench.py --dtype fp16 --steps 48 --warmup 8 --no-cpu-baseline --secondary "" > $OUT/r5_final_fp16.json 2> $OUT/r5_final_fp16.err
timeout 400 python bench.py --model mistral --batch 16 --steps 12 --warmup 2 --no-cpu-baseline > $OUT/r5_final_mistral_b16.json 2> $OUT/r5_final_mistral_b16.err
python - <<'PY'
import json
for f in ('gpurun_out/r5_final_fp16.json', 'gpurun_out/r5_final_mistral_b16.json'):
    try:
        for l in open(f):
            if l.startswith('{'):
                d = json.loads(l)
                print(f.split('/')[-1], d['dtype'], d['value'], d['ms_per_step'], 'accept', d['config']['mean_accept_len'], 'eq', d['config']['lookahead_equals_greedy'], d['config']['workload'][:60])
    except Exception as e:
        print(f, 'FAILED', e)
PY
tail -2 $OUT/r5_final_mistral_b16.err | cut -c1-300
