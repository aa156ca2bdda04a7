#!/bin/bash
# round 4 record run: default bench (cpu_baseline + secondary legs), then rocprofv3 stats + PMC passes of a short bench (scripts/gpu_profile.sh)
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout 1500 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err
echo "bench exit $?" >> gpurun_out/bench_default.err
STEPS=10 bash scripts/gpu_profile.sh > gpurun_out/profile.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/bench_default.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print({k: d[k] for k in ('value', 'ms_per_step', 'steps', 'dtype')}, 'accept', d['config']['mean_accept_len'], 'roofline', d['roofline']['frac'], 'step', d['roofline']['verify_step']['frac'],
              'cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('ms_per_step'))
        for s in d.get('secondary') or []:
            print('  secondary', s.get('workload', s)[:40], s.get('ms_per_step'), s.get('value'), s.get('error'))
PY
grep -E "^== (FETCH|WRITE)|^void k_|^k_" gpurun_out/profile.log | cut -c1-150 | head -40
tail -3 gpurun_out/bench_default.err
