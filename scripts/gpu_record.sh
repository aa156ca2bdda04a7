#!/bin/bash
# record run: default bench (with cpu_baseline), then rocprofv3 stats + PMC passes of a short bench
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout 1200 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err
echo "bench exit $?" >> gpurun_out/bench_default.err
STEPS=10 bash scripts/gpu_profile.sh > gpurun_out/profile.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/bench_default.log'):
    if l.startswith('{'):
        d = json.loads(l); print({k: d[k] for k in ('value', 'ms_per_step', 'steps')}, d['config']['mean_accept_len'], d['roofline']['frac'], d['roofline']['verify_step']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['ms_per_step'])
PY
grep -E "^== (FETCH|WRITE)|^void k_|^k_" gpurun_out/profile.log | cut -c1-150 | head -40
