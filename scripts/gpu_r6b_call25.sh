#!/bin/bash
# round 6, session 3, call 25: x direct in the slab launch at 5-8 blocks as the product default — mblock / e2e / batch / fp16 tests, then product vs lab (36 = 0) legs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_mblock.py tests/test_gpu_batch.py tests/test_gpu_e2e.py tests/test_gpu_fp16.py -m gpu -q > $OUT/r6b25_tests.log 2>&1; echo "tests exit $?"; tail -3 $OUT/r6b25_tests.log
run() {  # tag, env, args
  env $2 timeout 600 python bench.py $3 --steps 24 --warmup 4 --secondary "" --no-cpu-baseline > $OUT/r6b25_$1.log 2>&1
  tail -1 $OUT/r6b25_$1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'), (d['config'].get('speed_incl_prefill') or {}).get('prefill_ms'))" || tail -5 $OUT/r6b25_$1.log
}
for i in 1 2; do
  run mistral8_off_$i "LA_LAB_BUILD=1 LA_DEBUG=36=0" "--model mistral --batch 8"
  run mistral8_on_$i "LA_X=1" "--model mistral --batch 8"
done
run 7b8_off "LA_LAB_BUILD=1 LA_DEBUG=36=0" "--model 7b --batch 8"
run 7b8_on "LA_X=1" "--model 7b --batch 8"
run 13b8_off "LA_LAB_BUILD=1 LA_DEBUG=36=0" "--model 13b --batch 8"
run 13b8_on "LA_X=1" "--model 13b --batch 8"
run 7b6_off "LA_LAB_BUILD=1 LA_DEBUG=36=0" "--model 7b --batch 6"
run 7b6_on "LA_X=1" "--model 7b --batch 6"
