#!/bin/bash
# round 5, call 4: fat waves after the waterfall fix (uniform piece strides) — bitwise tests, per-launch A/B, step A/B (1 = round-4 kernels,
# 17 = fat gate/up, 49 = fat gate/up + slab + QKV) at Mistral bs=8 / 13B bs=4 / Mixtral bs=4; the 32-layer parity test with planted rows
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mblock.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "paired or schedules" > $OUT/r5c4_pytest.log 2>&1
echo "pytest exit $?" >> $OUT/r5c4_pytest.log
tail -4 $OUT/r5c4_pytest.log | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "full_size" -s > $OUT/r5c4_pytest_32layers.log 2>&1
echo "pytest exit $?" >> $OUT/r5c4_pytest_32layers.log
grep -E "decisive|passed|failed|Error|assert" $OUT/r5c4_pytest_32layers.log | cut -c1-250 | tail -8
LA_LAB_SET="6=49" timeout 900 python -m pytest tests/test_gpu_mblock.py tests/test_gpu_batch.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "not paired and not schedules" > $OUT/r5c4_pytest_fat49.log 2>&1
echo "pytest(6=49) exit $?" >> $OUT/r5c4_pytest_fat49.log
tail -3 $OUT/r5c4_pytest_fat49.log | cut -c1-220
for shp in "11008 4096" "14336 4096" "13824 5120"; do
  set -- $shp
  MB_F=$1 MB_K=$2 timeout 300 python scripts/gpu_mb_gemm.py time > $OUT/r5c4_gemm_$1.log 2>&1
  echo "== F=$1 K=$2"; grep -E "down|gate" $OUT/r5c4_gemm_$1.log | grep -v "k_gemm_mb\|paired" | cut -c1-120
done
for rep in a b; do
 for cfg in "mistral 8" "13b 4" "mixtral 4"; do
  set -- $cfg
  for v in 1 17 49; do
    LA_DEBUG="6=$v" timeout 500 python bench.py --model $1 --batch $2 --steps 24 --warmup 4 --no-cpu-baseline > $OUT/r5c4_${1}_v${v}_$rep.json 2> $OUT/r5c4_${1}_v${v}_$rep.err
  done
 done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5c4_*.json')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            print(f.split('/')[-1], d['value'], d['ms_per_step'], 'accept', d['config']['mean_accept_len'], 'eq', d['config']['lookahead_equals_greedy'],
                  'prefill_ms', d['config']['speed_incl_prefill']['prefill_ms'])
PY
