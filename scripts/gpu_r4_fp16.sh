#!/bin/bash
# round 4, call 4: the float16 build (tests/test_gpu_fp16.py), the suites touched by the dtype refactor and the two-graph step,
# the attention kernel test, an fp16 bench leg next to the bf16 one
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_fp16.py tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_edges.py tests/test_gpu_batch.py -m gpu -q -p no:cacheprovider --timeout 900 -x -s > $OUT/r4_pytest_fp16.log 2>&1
echo "pytest exit $?" >> $OUT/r4_pytest_fp16.log
grep -E "^\[fp16|passed|failed|Error|assert" $OUT/r4_pytest_fp16.log | cut -c1-300 | tail -20
run() {
  LA_DEBUG="$2" timeout 300 python bench.py --steps ${STEPS:-48} --warmup 6 --no-cpu-baseline --secondary "" --profile-iters 2 $3 > /tmp/ab.json 2> /tmp/ab.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/ab.json'))
    ev = d['roofline']['verify_step'].get('ms_by_class_events', {})
    print(f"[{sys.argv[1]:28s}] {d['ms_per_step']:.4f} ms/step  tok/s {d['value']:.0f}  dtype {d['dtype']} eq_greedy={d['config'].get('lookahead_equals_greedy')}  events {ev}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('/tmp/ab.err').read()[-600:])
PY
}
for rep in 1 2; do
  run "bf16 default" "" ""
  run "fp16" "" "--dtype fp16"
done | tee $OUT/r4_fp16_bench.txt
