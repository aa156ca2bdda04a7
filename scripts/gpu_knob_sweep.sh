#!/bin/bash
# ms/step of the default bench leg under la_debug_set knob settings: gpu_knob_sweep.sh "" "1=40" "2=1" ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
for rep in 1 2; do
for kv in "$@"; do
  LA_DEBUG="$kv" timeout 300 python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --secondary "" --profile-iters 1 ${BENCH_ARGS:-} > /tmp/knob.json 2> /tmp/knob.err
  python - "$kv" "$rep" <<'PY'
import json, sys
kv, rep = sys.argv[1], sys.argv[2]
try:
    d = json.load(open('/tmp/knob.json'))
    print(f"knobs [{kv:12s}] rep {rep}: {d['ms_per_step']:.4f} ms/step  equal_greedy={d['config'].get('lookahead_equals_greedy')}")
except Exception as e:
    print(kv, rep, 'FAILED', e, open('/tmp/knob.err').read()[-400:])
PY
done; done | tee -a $OUT/knob_sweep.txt
