#!/bin/bash
# round 6, session 3, call 6: gate/up planned WITHOUT padded MFMA rows (fewer workgroups) vs the one-workgroup-per-CU plan, alternating
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
run() {  # tag, env, args
  env $2 python bench.py $3 --steps 24 --warmup 4 --secondary "" --no-cpu-baseline > $OUT/r6b6_$1.log 2>&1
  tail -1 $OUT/r6b6_$1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'), d['config'].get('speed_incl_prefill'))"
}
for i in 1 2; do
  run mistral8_gu256_$i "LA_X=0" "--model mistral --batch 8"
  run mistral8_gu224_$i "LA_GU_WG=224" "--model mistral --batch 8"
done
for i in 1 2; do
  run 13b4_gu256_$i "LA_X=0" "--model 13b --batch 4"
  run 13b4_gu216_$i "LA_GU_WG=216" "--model 13b --batch 4"
done
run 7b8_gu256 "LA_X=0" "--model 7b --batch 8"
run 7b8_gu172 "LA_GU_WG=172" "--model 7b --batch 8"
