#!/bin/bash
# round 6, session 3, call 17: the second (full-block) QKV image for plans with a pair count that is not a multiple of 4 (13B) as the engine default — full GPU suite, 13B legs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q > $OUT/r6b17_pytest_gpu.log 2>&1; echo "suite exit $?"; tail -4 $OUT/r6b17_pytest_gpu.log
for leg in "13b 4" "13b 8" "13b 1"; do set -- $leg
  python bench.py --model $1 --batch $2 --steps 24 --warmup 4 --secondary "" --no-cpu-baseline > $OUT/r6b17_$1_$2.log 2>&1
  tail -1 $OUT/r6b17_$1_$2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1:$2', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'))"
done
