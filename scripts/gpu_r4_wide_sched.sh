#!/bin/bash
# round 4, call 11: wide multi-block GEMM schedules (la_lab 24): parity suite under the knob, the GEMMs alone, batch legs A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
LA_LAB_SET="24=2" timeout 900 python -m pytest tests/test_gpu_mblock.py -m gpu -q -p no:cacheprovider --timeout 600 -x > $OUT/r4_pytest_sched.log 2>&1
echo "pytest exit $?" >> $OUT/r4_pytest_sched.log
tail -5 $OUT/r4_pytest_sched.log | cut -c1-300
timeout 600 python scripts/gpu_mb_gemm.py time 2>&1 | grep -v amdgpu.ids | tee $OUT/r4_wide_sched_micro2.txt
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
  leg "mistral bs=8  schedule 0 (default)" "" mistral 8 ""
  leg "mistral bs=8  auto (3 at TW>=3, else 2)" "24=1" mistral 8 ""
  leg "mistral bs=8  schedule 3" "24=3" mistral 8 ""
  leg "13b bs=4      schedule 0 (default)" "" 13b 4 ""
  leg "13b bs=4      schedule 2" "24=2" 13b 4 ""
done | tee $OUT/r4_wide_sched_ab2.txt
