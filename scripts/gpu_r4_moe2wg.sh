#!/bin/bash
# round 4, call 30: merged-expert launches as two workgroups per CU (la_lab 25): MoE parity suites under each setting + Mixtral bs=4 A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
for v in 1 3; do
  LA_LAB_SET="25=$v" timeout 900 python -m pytest tests/test_gpu_moe.py tests/test_gpu_mblock.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "moe or mixtral or expert" > $OUT/r4_pytest_moe2wg_$v.log 2>&1
  echo "pytest exit $?" >> $OUT/r4_pytest_moe2wg_$v.log
  tail -2 $OUT/r4_pytest_moe2wg_$v.log | cut -c1-200
done
leg() {
  LA_DEBUG="$2" BENCH_IS_SECONDARY=1 timeout 600 python bench.py --model mixtral --batch 4 --steps 24 --warmup 4 --no-cpu-baseline --profile-iters 1 > /tmp/leg.json 2> /tmp/leg.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/leg.json'))
    print(f"[{sys.argv[1]:44s}] {d['ms_per_step']:.3f} ms/step  tok/s {d['value']:.0f}  accept {d['config']['mean_accept_len']}  eq_greedy={d['config'].get('lookahead_equals_greedy')}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('/tmp/leg.err').read()[-800:])
PY
}
for rep in 1 2; do
  leg "mixtral bs=4  one workgroup per CU (25=0)" "25=0"
  leg "mixtral bs=4  gate/up two per CU (25=1, default)" ""
  leg "mixtral bs=4  gate/up + down two per CU (25=3)" "25=3"
done | tee $OUT/r4_moe2wg.txt
