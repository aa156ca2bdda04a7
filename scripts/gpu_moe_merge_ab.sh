#!/bin/bash
# gathered multi-block MoE: one launch per stage for all experts (default) vs one launch per expert (la_debug_set(16, 1))
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_mblock.py tests/test_gpu_moe.py -x -q -m gpu 2>&1 | tail -3
run() {
  timeout 300 python bench.py --model mixtral --batch $B --steps 24 --warmup 4 --no-cpu-baseline --secondary "" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('mixtral b$B LA_DEBUG=${LA_DEBUG:-}', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'))"
}
for B in 4 8; do
  for i in 1 2; do
    LA_DEBUG="16=1" run
    LA_DEBUG= run
  done
done
