#!/bin/bash
# round 5, call 3: fat waves for the paired slab / QKV launches too (la_lab_set(6, 49)) — bitwise tests, per-launch A/B, step A/B at
# Mistral bs=8 / 13B bs=4 / Mixtral bs=4 (default 17 = fat gate/up only; 1 = the round-4 kernels; 49 = all fat)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mblock.py tests/test_gpu_moe.py -m gpu -q -p no:cacheprovider --timeout 600 -x > $OUT/r5c3_pytest.log 2>&1
echo "pytest exit $?" >> $OUT/r5c3_pytest.log
tail -6 $OUT/r5c3_pytest.log | cut -c1-220
LA_LAB_SET="6=49" timeout 900 python -m pytest tests/test_gpu_mblock.py tests/test_gpu_batch.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "not paired and not schedules" > $OUT/r5c3_pytest_fat49.log 2>&1
echo "pytest(6=49) exit $?" >> $OUT/r5c3_pytest_fat49.log
tail -4 $OUT/r5c3_pytest_fat49.log | cut -c1-220
for shp in "14336 4096" "13824 5120"; do
  set -- $shp
  MB_F=$1 MB_K=$2 timeout 300 python scripts/gpu_mb_gemm.py time > $OUT/r5c3_gemm_$1.log 2>&1
  echo "== F=$1 K=$2"; grep -E "down|gate" $OUT/r5c3_gemm_$1.log | grep -v k_gemm_mb | cut -c1-120
done
for rep in a b; do
 for cfg in "mistral 8" "13b 4" "mixtral 4"; do
  set -- $cfg
  for v in 1 17 49; do
    LA_DEBUG="6=$v" timeout 500 python bench.py --model $1 --batch $2 --steps 24 --warmup 4 --no-cpu-baseline > $OUT/r5c3_${1}_v${v}_$rep.json 2> $OUT/r5c3_${1}_v${v}_$rep.err
  done
 done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5c3_*.json')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            print(f.split('/')[-1], d['value'], d['ms_per_step'], 'accept', d['config']['mean_accept_len'], 'eq', d['config']['lookahead_equals_greedy'],
                  'prefill_ms', d['config']['speed_incl_prefill']['prefill_ms'])
PY
