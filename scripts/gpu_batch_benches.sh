#!/bin/bash
# the batch configurations of BASELINE.json (configs 3-5) through bench.py on one MI355X; one JSON line each -> gpurun_out/
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" 2>gpurun_out/bench_$name.err | tail -1 > gpurun_out/bench_$name.json; python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    r = json.load(open(f'gpurun_out/bench_{n}.json'))
    print(n, r['value'], r['unit'], 'ms/step', r['ms_per_step'], 'accept', r['config'].get('mean_accept_len'), 'roofline', {k: r['roofline'].get(k) for k in ('bound', 'achieved', 'frac')}, 'eq', r['config'].get('lookahead_equals_greedy'))
except Exception as e:
    print(n, 'FAILED', e)
PY
}
run 7b_b4 --model 7b --batch 4 --steps 16 --warmup 4
run 13b_b4 --model 13b --batch 4 --steps 16 --warmup 4
run mistral_b8 --model mistral --batch 8 --steps 16 --warmup 4
run mixtral_b4 --model mixtral --batch 4 --steps 12 --warmup 3
