#!/bin/bash
# round 4, call 17: gate/up of >= 5 blocks as two co-resident 256-row workgroups per CU (la_lab 5 = 1): parity, GEMMs alone, Mistral bs=8 A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
LA_LAB_SET="5=1" timeout 900 python -m pytest tests/test_gpu_mblock.py -m gpu -q -p no:cacheprovider --timeout 600 -x > $OUT/r4_pytest_2wg.log 2>&1
echo "pytest exit $?" >> $OUT/r4_pytest_2wg.log
tail -5 $OUT/r4_pytest_2wg.log | cut -c1-300
timeout 600 python scripts/gpu_mb_gemm.py time 2>&1 | grep -v amdgpu.ids | tee $OUT/r4_wide_2wg_micro.txt
leg() {   # label, LA_DEBUG, model, batch, extra
  LA_DEBUG="$2" BENCH_IS_SECONDARY=1 timeout 600 python bench.py --model $3 --batch $4 --steps 24 --warmup 4 --no-cpu-baseline --profile-iters 1 $5 > /tmp/leg.json 2> /tmp/leg.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/leg.json'))
    print(f"[{sys.argv[1]:38s}] {d['ms_per_step']:.3f} ms/step  tok/s {d['value']:.0f}  accept {d['config']['mean_accept_len']}  eq_greedy={d['config'].get('lookahead_equals_greedy')}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('/tmp/leg.err').read()[-800:])
PY
}
for rep in 1 2; do
  leg "mistral bs=8  default" "" mistral 8 ""
  leg "mistral bs=8  gate/up 2 WGs per CU" "5=1" mistral 8 ""
  leg "mistral bs=8  paired gate/up (6=3)" "6=3" mistral 8 ""
done | tee $OUT/r4_wide_2wg_ab.txt
