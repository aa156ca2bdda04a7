#!/bin/bash
# round 6, session 3, call 26: the 7B bs=6 outlier of call 25 re-measured (x direct slab launch: product vs lab build with knob 36 = 0), three alternations
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
run() {  # tag, env, args
  env $2 timeout 600 python bench.py $3 --steps 32 --warmup 4 --secondary "" --no-cpu-baseline > $OUT/r6b26_$1.log 2>&1
  tail -1 $OUT/r6b26_$1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'))" || tail -5 $OUT/r6b26_$1.log
}
for i in 1 2 3; do
  run 7b6_off_$i "LA_LAB_BUILD=1 LA_DEBUG=36=0" "--model 7b --batch 6"
  run 7b6_on_$i "LA_X=1" "--model 7b --batch 6"
done
run 7b5_off "LA_LAB_BUILD=1 LA_DEBUG=36=0" "--model 7b --batch 5"
run 7b5_on "LA_X=1" "--model 7b --batch 5"
run 7b7_off "LA_LAB_BUILD=1 LA_DEBUG=36=0" "--model 7b --batch 7"
run 7b7_on "LA_X=1" "--model 7b --batch 7"
