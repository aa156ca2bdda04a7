#!/bin/bash
# round 6, session 3, call 18: slab launches with 2 K splits x 4 token groups at >= 5 blocks (lab knob 12) re-measured on top of the fat / fatd kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
run() {  # tag, env, args
  env LA_LAB_BUILD=1 $2 timeout 600 python bench.py $3 --steps 24 --warmup 4 --secondary "" --no-cpu-baseline > $OUT/r6b18_$1.log 2>&1
  tail -1 $OUT/r6b18_$1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'))" || tail -5 $OUT/r6b18_$1.log
}
for i in 1 2; do
  run mistral8_ks4_$i "LA_MB_KS2=0" "--model mistral --batch 8"
  run mistral8_ks2_$i "LA_MB_KS2=1" "--model mistral --batch 8"
done
run 7b8_ks4 "LA_MB_KS2=0" "--model 7b --batch 8"
run 7b8_ks2 "LA_MB_KS2=1" "--model 7b --batch 8"
