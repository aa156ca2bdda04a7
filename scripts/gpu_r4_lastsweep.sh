#!/bin/bash
# round 4, last sweep: round-2 form decisions re-checked under the round-4 wide-GEMM schedule (unpaired slab / QKV launches at 13B bs=4,
# 2 K splits over 4 token groups at Mistral bs=8)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
leg() {   # label, LA_DEBUG, model, batch
  LA_DEBUG="$2" BENCH_IS_SECONDARY=1 timeout 600 python bench.py --model $3 --batch $4 --steps 24 --warmup 4 --no-cpu-baseline --profile-iters 1 > /tmp/leg.json 2> /tmp/leg.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/leg.json'))
    print(f"[{sys.argv[1]:44s}] {d['ms_per_step']:.3f} ms/step  tok/s {d['value']:.0f}  eq_greedy={d['config'].get('lookahead_equals_greedy')}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('/tmp/leg.err').read()[-600:])
PY
}
for rep in 1 2; do
  leg "13b bs=4      paired slab / QKV (default)" "" 13b 4
  leg "13b bs=4      unpaired (6=0)" "6=0" 13b 4
  leg "mistral bs=8  4 K splits x 2 token groups (default)" "" mistral 8
  leg "mistral bs=8  2 K splits x 4 token groups (12=1)" "12=1" mistral 8
done | tee $OUT/r4_lastsweep.txt
