#!/bin/bash
# ms/step of the default bench leg for cfg.fuse settings: gpu_fuse_ab.sh 0 4 12 ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
for rep in 1 2; do
for f in "$@"; do
  timeout 300 python bench.py --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline --secondary "" --profile-iters 1 --fuse $f > /tmp/fz.json 2> /tmp/fz.err
  python - "$f" "$rep" <<'PY'
import json, sys
f, rep = sys.argv[1], sys.argv[2]
try:
    d = json.load(open('/tmp/fz.json'))
    print(f"fuse {f:3s} rep {rep}: {d['ms_per_step']:.4f} ms/step  equal_greedy={d['config'].get('lookahead_equals_greedy')} accept {d['config']['mean_accept_len']}")
except Exception as e:
    print(f, rep, 'FAILED', e, open('/tmp/fz.err').read()[-600:])
PY
done; done | tee -a $OUT/fuse_ab.txt
