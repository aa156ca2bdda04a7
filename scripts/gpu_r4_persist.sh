#!/bin/bash
# round 4, call 25: the MLP half of a layer as ONE launch (norm -> gate/up -> down_proj, cfg.fuse 5 / 21 / 23): bitwise test + step A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "role_fused or fused_norm" > $OUT/r4_pytest_persist.log 2>&1
echo "pytest exit $?" >> $OUT/r4_pytest_persist.log
tail -5 $OUT/r4_pytest_persist.log | cut -c1-300
run() {   # label, fuse, gemm cfg
  timeout 600 python bench.py --steps 32 --warmup 6 --no-cpu-baseline --profile-iters 2 --secondary "" --fuse $2 $3 > /tmp/ab.json 2> /tmp/ab.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/ab.json'))
    ev = d['roofline']['verify_step'].get('ms_by_class_events', {})
    print(f"[{sys.argv[1]:58s}] {d['ms_per_step']:.4f} ms/step  tok/s {d['value']:.0f}  eq_greedy={d['config'].get('lookahead_equals_greedy')}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('/tmp/ab.err').read()[-600:])
PY
}
for rep in 1 2; do
  run "separate kernels (default)" 0 ""
  run "separate kernels, down_proj variant of the role form" 0 "--gemm-cfg 0,0,0,0,1026,0,0,0"
  run "gate/up -> down one launch (fuse 4)" 4 "--gemm-cfg 0,0,0,0,1026,0,0,0"
  run "norm -> gate/up one launch (fuse 17)" 17 ""
  run "norm -> gate/up -> down one launch (fuse 21)" 21 "--gemm-cfg 0,0,0,0,1026,0,0,0"
  run "+ input norm -> QKV: 4 launches per layer (fuse 23)" 23 "--gemm-cfg 0,0,0,0,1026,0,0,0"
done | tee $OUT/r4_persist_ab.txt
