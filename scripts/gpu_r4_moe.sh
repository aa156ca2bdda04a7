#!/bin/bash
# round 4, call 8: from_hf test, MoE suites, K splits of the gathered experts' down projection (la_lab 22) on the Mixtral bs=4 leg
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_from_hf.py tests/test_gpu_moe.py tests/test_gpu_trie.py -m gpu -q -p no:cacheprovider --timeout 900 -x > $OUT/r4_pytest_moe.log 2>&1
echo "pytest exit $?" >> $OUT/r4_pytest_moe.log
tail -5 $OUT/r4_pytest_moe.log | cut -c1-300
for ks in 1 2; do
LA_LAB_SET="22=$ks" timeout 900 python -m pytest tests/test_gpu_mblock.py -m gpu -q -p no:cacheprovider --timeout 900 -x -k "mixtral or moe" > $OUT/r4_pytest_moe_ks$ks.log 2>&1
echo "ks=$ks: $(tail -1 $OUT/r4_pytest_moe_ks$ks.log)"
done
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
  leg "mixtral bs=4  down ks=4 (round 3)" "22=4" mixtral 4 ""
  leg "mixtral bs=4  down ks=2" "22=2" mixtral 4 ""
  leg "mixtral bs=4  down ks=1" "22=1" mixtral 4 ""
done | tee $OUT/r4_moe_down_ks.txt
