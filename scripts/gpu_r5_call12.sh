#!/bin/bash
# round 5, call 12: the fuller QKV image of the GQA models over 128 regions (24 RoPE pairs per region, 32-row blocks 75 % full) instead of 96
# (LA_QKV_MB_WG): 128 regions x 2 token groups = 256 workgroups = one per CU at 4 blocks (bit 12 form) and at 8 blocks (bit 9 form)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
run() {   # tag model batch wg pair
  if [ "$4" = "def" ]; then
    LA_DEBUG="6=$5" timeout 500 python bench.py --model $2 --batch $3 --steps 32 --warmup 4 --no-cpu-baseline > $OUT/r5c12_$1.json 2> $OUT/r5c12_$1.err
  else
    LA_QKV_MB_WG=$4 LA_DEBUG="6=$5" timeout 500 python bench.py --model $2 --batch $3 --steps 32 --warmup 4 --no-cpu-baseline > $OUT/r5c12_$1.json 2> $OUT/r5c12_$1.err
  fi
}
for rep in a b; do
  run mistral8_wg96_p4465_$rep mistral 8 def 4465
  run mistral8_wg128_p4465_$rep mistral 8 128 4465
  run mistral8_wg128_p4977_$rep mistral 8 128 4977
  run mistral8_wg96_p4977_$rep mistral 8 def 4977
  run mixtral4_wg96_$rep mixtral 4 def 4465
  run mixtral4_wg128_$rep mixtral 4 128 4465
done
run mistral4_wg96 mistral 4 def 4465
run mistral4_wg128 mistral 4 128 4465
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5c12_*.json')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            print(f.split('/')[-1], d['value'], d['ms_per_step'], 'accept', d['config']['mean_accept_len'], 'eq', d['config']['lookahead_equals_greedy'])
PY
