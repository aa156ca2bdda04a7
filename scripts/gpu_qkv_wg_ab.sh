#!/bin/bash
# the second QKV image of the multi-block path (cfg.qkv_mb_wg: fewer, fuller workgroups x token quarters) on / off (LA_QKV_MB_WG=0)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
run() {
  timeout 300 python bench.py --model $M --batch $B --steps 24 --warmup 4 --no-cpu-baseline --secondary "" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$M b$B LA_QKV_MB_WG=${LA_QKV_MB_WG-auto}', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'))"
}
for leg in "mistral 8" "mixtral 4" "13b 4"; do
  set -- $leg; M=$1; B=$2
  for i in 1 2; do
    LA_QKV_MB_WG=0 run
    unset LA_QKV_MB_WG; run
  done
done
