#!/bin/bash
# round 5, call 14: the fat slab launches (o_proj / down, 4 K splits) with ONE K split per XCD (la_lab_set(6, 4465 + 8192)) vs the plain grid order:
# bitwise test, down microbenchmark at the three shapes, Mistral bs=8 / Mixtral bs=4 steps and the 512-token prefill of the bs=1 run
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_mblock.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "paired" > $OUT/r5c14_pytest.log 2>&1
echo "pytest exit $?" >> $OUT/r5c14_pytest.log
tail -4 $OUT/r5c14_pytest.log | cut -c1-220
for shape in "11008 4096" "14336 4096" "13824 5120"; do
  set -- $shape
  echo "== F=$1 K=$2" | tee -a $OUT/r5c14_gemm.log
  MB_F=$1 MB_K=$2 timeout 300 python scripts/gpu_mb_gemm.py time 2>&1 | grep "^down" | tee -a $OUT/r5c14_gemm.log
done
for rep in a b; do
  for v in 4465 12657; do
    LA_DEBUG="6=$v" timeout 500 python bench.py --model mistral --batch 8 --steps 32 --warmup 4 --no-cpu-baseline > $OUT/r5c14_mistral8_v${v}_$rep.json 2> $OUT/r5c14_mistral8_v${v}_$rep.err
  done
done
for v in 4465 12657; do
  LA_DEBUG="6=$v" timeout 500 python bench.py --model mixtral --batch 4 --steps 32 --warmup 4 --no-cpu-baseline > $OUT/r5c14_mixtral4_v${v}_a.json 2> $OUT/r5c14_mixtral4_v${v}_a.err
  LA_DEBUG="6=$v" timeout 500 python bench.py --steps 32 --warmup 4 --no-cpu-baseline --secondary "" > $OUT/r5c14_7b1_v${v}_a.json 2> $OUT/r5c14_7b1_v${v}_a.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5c14_*.json')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            print(f.split('/')[-1], d['value'], d['ms_per_step'], 'accept', d['config']['mean_accept_len'], 'eq', d['config']['lookahead_equals_greedy'],
                  'prefill_ms', d['config']['speed_incl_prefill']['prefill_ms'])
PY
