#!/bin/bash
# gathered MoE: an expert's last single block through the one-block body (default) vs the padded two-block pass (la_lab_set(16, 2))
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_mblock.py tests/test_gpu_moe.py -x -q -m gpu 2>&1 | tail -3
run() {
  timeout 300 python bench.py --model $M --batch $B --steps 24 --warmup 4 --no-cpu-baseline --secondary "" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$M b$B LA_DEBUG=${LA_DEBUG:-}', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'))"
}
M=mixtral; B=4
for i in 1 2; do
  LA_DEBUG="16=2" run
  LA_DEBUG= run
done
