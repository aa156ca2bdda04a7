#!/bin/bash
# round 5, call 9: the register-staged form of the paired gate/up fat launch (la_lab_set(6, 369 + 2048)) against the LDS-DMA form (369):
# bitwise test, the gate/up microbenchmark at the three shapes, Mistral bs=8 / 13B bs=8-row / prefill steps A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_mblock.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "paired" > $OUT/r5c9_pytest.log 2>&1
echo "pytest exit $?" >> $OUT/r5c9_pytest.log
tail -6 $OUT/r5c9_pytest.log | cut -c1-220
for shape in "11008 4096" "14336 4096" "13824 5120"; do
  set -- $shape
  echo "== F=$1 K=$2"
  MB_F=$1 MB_K=$2 timeout 300 python scripts/gpu_mb_gemm.py time 2>&1 | grep "gate/up" | tee -a $OUT/r5c9_gemm.log
done
for rep in a b; do
 for cfg in "mistral 8" "13b 4"; do
  set -- $cfg
  for v in 369 2417; do
    LA_DEBUG="6=$v" timeout 500 python bench.py --model $1 --batch $2 --steps 24 --warmup 4 --no-cpu-baseline > $OUT/r5c9_${1}_v${v}_$rep.json 2> $OUT/r5c9_${1}_v${v}_$rep.err
  done
 done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5c9_*.json')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            print(f.split('/')[-1], d['value'], d['ms_per_step'], 'accept', d['config']['mean_accept_len'], 'eq', d['config']['lookahead_equals_greedy'],
                  'prefill_ms', d['config']['speed_incl_prefill']['prefill_ms'])
PY
