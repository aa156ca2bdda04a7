#!/bin/bash
# round 4, call 1: the new / changed parity tests + a same-box baseline of the default bench (short)
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -s \
  -k "mixtral8x7b_layer_shape or full_size_llama7b_32_layers or trie_trace_6" > gpurun_out/r4_parity.log 2>&1
echo "pytest exit $?" >> gpurun_out/r4_parity.log
grep -E "^\[|passed|failed|Error|error|assert" gpurun_out/r4_parity.log | cut -c1-260 | tail -40
timeout 600 python bench.py --steps 32 --warmup 4 --no-cpu-baseline --secondary '' > gpurun_out/r4_bench_base.log 2> gpurun_out/r4_bench_base.err
python - <<'PY'
import json
for l in open('gpurun_out/r4_bench_base.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print('BENCH value', d['value'], 'ms/step', d['ms_per_step'], 'accept', d['config']['mean_accept_len'],
              'eq_greedy', d['config']['lookahead_equals_greedy'], 'roofline frac', d['roofline']['frac'], 'step frac', d['roofline']['verify_step']['frac'])
PY
tail -3 gpurun_out/r4_bench_base.err
