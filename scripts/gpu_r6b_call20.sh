#!/bin/bash
# round 6, session 3, call 20: the paired direct-weight gate/up launch at 3-4 blocks (lab knob 35 bit 1: k_gemm_fatd<2>, two regions x 4 token tiles, grid.z = 2)
# vs the one-region fat form that is the default there — 13B bs=4, Mistral bs=4, 7B bs=4, alternating; bitwise check through equals_greedy + the mblock tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
run() {  # tag, LA_DEBUG, args
  LA_LAB_BUILD=1 LA_DEBUG="$2" timeout 600 python bench.py $3 --steps 24 --warmup 4 --secondary "" --no-cpu-baseline > $OUT/r6b20_$1.log 2>&1
  tail -1 $OUT/r6b20_$1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'))" || tail -5 $OUT/r6b20_$1.log
}
for i in 1 2; do
  run 13b4_one_$i "35=1" "--model 13b --batch 4"
  run 13b4_pairD_$i "35=3" "--model 13b --batch 4"
done
run mistral4_one "35=1" "--model mistral --batch 4"
run mistral4_pairD "35=3" "--model mistral --batch 4"
run 7b4_one "35=1" "--model 7b --batch 4"
run 7b4_pairD "35=3" "--model 7b --batch 4"
