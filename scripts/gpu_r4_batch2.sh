#!/bin/bash
# round 4, call 6: multi-block suites (GQA start rotation in k_tree_attn_mb, replay overlap of the chained device trie), then batch legs:
# rotation on / off (la_lab 21), paired gate/up at 256 rows (la_lab 6 = 3), device trie chained vs host trie
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_mblock.py tests/test_gpu_moe.py tests/test_gpu_batch.py tests/test_gpu_trie.py tests/test_gpu_fp16.py -m gpu -q -p no:cacheprovider --timeout 900 -x > $OUT/r4_pytest_mb2.log 2>&1
echo "pytest exit $?" >> $OUT/r4_pytest_mb2.log
tail -6 $OUT/r4_pytest_mb2.log | cut -c1-300
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
  leg "mistral bs=8  rotation (default)" "" mistral 8 ""
  leg "mistral bs=8  no rotation" "21=0" mistral 8 ""
  leg "mistral bs=8  device trie chained" "" mistral 8 "--device-trie"
  leg "13b bs=4      default" "" 13b 4 ""
  leg "13b bs=4      paired gate/up (6=3)" "6=3" 13b 4 ""
done | tee $OUT/r4_batch_ab2.txt
leg "mixtral bs=4  rotation (default)" "" mixtral 4 "" | tee -a $OUT/r4_batch_ab2.txt
leg "mixtral bs=4  no rotation" "21=0" mixtral 4 "" | tee -a $OUT/r4_batch_ab2.txt
leg "mixtral bs=4  paired gate/up (6=3)" "6=3" mixtral 4 "" | tee -a $OUT/r4_batch_ab2.txt
