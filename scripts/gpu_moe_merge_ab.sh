#!/bin/bash
# gathered multi-block MoE: one launch per stage for all experts + fused accumulate/norm (default) vs one launch per expert and
# separate accumulate / norm kernels (la_lab_set(16, 1))
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3; fi
run() {
  timeout 300 python bench.py --model mixtral --batch $B --steps 24 --warmup 4 --no-cpu-baseline --secondary "" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('mixtral b$B LA_DEBUG=${LA_DEBUG:-}', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'))"
}
for B in ${BATCHES:-4 8}; do
  for i in 1 2; do
    LA_DEBUG="16=1" run
    LA_DEBUG= run
  done
done
