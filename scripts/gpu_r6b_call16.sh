#!/bin/bash
# round 6, session 3, call 16: PMC passes of the three batch legs at HEAD (pmc_secondary.json), then 13B bs=4 with the QKV image planned for 240 full
# workgroups (32 RoPE pairs each: 8-byte epilogue stores, no padded MFMA rows) vs the 256 x 30 plan, alternating
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
bash scripts/gpu_prof_secondary.sh > $OUT/r06d_prof_secondary.log 2>&1; tail -3 $OUT/r06d_prof_secondary.log | cut -c1-160
run() {  # tag, env, args
  env $2 timeout 600 python bench.py $3 --steps 24 --warmup 4 --secondary "" --no-cpu-baseline > $OUT/r6b16_$1.log 2>&1
  tail -1 $OUT/r6b16_$1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'))" || tail -5 $OUT/r6b16_$1.log
}
for i in 1 2; do
  run 13b4_plan256_$i "LA_X=0" "--model 13b --batch 4"
  run 13b4_plan240_$i "LA_QKV_MB_WG=240" "--model 13b --batch 4"
done
run 13b8_plan256 "LA_X=0" "--model 13b --batch 8"
run 13b8_plan240 "LA_QKV_MB_WG=240" "--model 13b --batch 8"
