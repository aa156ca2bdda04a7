#!/bin/bash
# round 5, call 11: more repetitions of call 10's A/B (Mixtral bs=4 varies by +-0.3 ms run to run)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
for rep in c d e f; do
 for cfg in "mixtral 4" "mistral 4"; do
  set -- $cfg
  for v in 369 4465; do
    LA_DEBUG="6=$v" timeout 500 python bench.py --model $1 --batch $2 --steps 32 --warmup 4 --no-cpu-baseline > $OUT/r5c11_${1}_v${v}_$rep.json 2> $OUT/r5c11_${1}_v${v}_$rep.err
  done
 done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5c11_*.json')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            print(f.split('/')[-1], d['value'], d['ms_per_step'], 'accept', d['config']['mean_accept_len'], 'eq', d['config']['lookahead_equals_greedy'])
PY
