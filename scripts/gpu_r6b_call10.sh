#!/bin/bash
# round 6, session 3, call 10: HEAD after the slab variant was dropped — mblock / e2e / batch / fp16 GPU tests, Mistral bs=8 + 7B bs=6 sanity
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_mblock.py tests/test_gpu_batch.py tests/test_gpu_e2e.py tests/test_gpu_fp16.py -m gpu -q > $OUT/r6b10_tests.log 2>&1; echo "tests exit $?"; tail -4 $OUT/r6b10_tests.log
for leg in "mistral 8" "7b 6" "13b 8"; do set -- $leg
  python bench.py --model $1 --batch $2 --steps 24 --warmup 4 --secondary "" --no-cpu-baseline > $OUT/r6b10_$1_$2.log 2>&1
  tail -1 $OUT/r6b10_$1_$2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1:$2', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'))"
done
