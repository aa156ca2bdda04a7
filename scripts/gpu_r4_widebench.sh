#!/bin/bash
# round 4, call 23: the batch legs with wide per-sample trees (bench.py --batch B --decoding-length 128 --branch-length 32) next to the 64-row setting
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
leg() {   # label, model, batch, DL, BL
  BENCH_IS_SECONDARY=1 timeout 600 python bench.py --model $2 --batch $3 --decoding-length $4 --branch-length $5 --steps 24 --warmup 4 --no-cpu-baseline --profile-iters 1 > /tmp/leg.json 2> /tmp/leg.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/leg.json'))
    print(f"[{sys.argv[1]:44s}] {d['ms_per_step']:.3f} ms/step  tok/s {d['value']:.0f}  accept {d['config']['mean_accept_len']}  mean draft {d['config'].get('mean_draft_len')}  eq_greedy={d['config'].get('lookahead_equals_greedy')}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('/tmp/leg.err').read()[-1200:])
PY
}
{
leg "7b bs=4   decoding 64 / branch 12" 7b 4 64 12
leg "7b bs=4   decoding 128 / branch 32 (2 blocks each)" 7b 4 128 32
leg "7b bs=2   decoding 256 / branch 32 (4 blocks each)" 7b 2 256 32
leg "13b bs=4  decoding 128 / branch 32" 13b 4 128 32
} | tee $OUT/r4_widebench.txt
