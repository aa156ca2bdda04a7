#!/bin/bash
# round 5, call 2: the fat-wave gate/up kernel (k_gemm_fat, la_lab_set(6, 17)) — bitwise test, per-launch A/B at the 7B / Mistral / 13B
# shapes, step A/B at Mistral bs=8 / 13B bs=4; the attention with K AND V of the first tile requested early (A/B vs la_lab_set(18, 1))
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mblock.py tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider --timeout 600 -x > $OUT/r5c2_pytest.log 2>&1
echo "pytest exit $?" >> $OUT/r5c2_pytest.log
tail -8 $OUT/r5c2_pytest.log | cut -c1-220
for shp in "11008 4096" "14336 4096" "13824 5120"; do
  set -- $shp
  MB_F=$1 MB_K=$2 timeout 300 python scripts/gpu_mb_gemm.py time > $OUT/r5c2_gemm_$1.log 2>&1
  echo "== F=$1 K=$2"; grep -E "gate/up" $OUT/r5c2_gemm_$1.log | cut -c1-120
done
B1="python bench.py --steps 64 --warmup 8 --no-cpu-baseline --secondary \"\""
for tag in specv_a nospec_a specv_b nospec_b; do
  case $tag in nospec*) export LA_DEBUG="18=1";; *) unset LA_DEBUG;; esac
  timeout 400 bash -c "$B1" > $OUT/r5c2_ab_$tag.json 2> $OUT/r5c2_ab_$tag.err
done
unset LA_DEBUG
for tag in mistral_def_a mistral_fat_a mistral_def_b mistral_fat_b 13b_def_a 13b_fat_a; do
  case $tag in *fat*) export LA_DEBUG="6=17";; *) unset LA_DEBUG;; esac
  case $tag in mistral*) M="--model mistral --batch 8";; *) M="--model 13b --batch 4";; esac
  timeout 400 python bench.py $M --steps 24 --warmup 4 --no-cpu-baseline > $OUT/r5c2_$tag.json 2> $OUT/r5c2_$tag.err
done
unset LA_DEBUG
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5c2_*.json')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            vs = d['roofline'].get('verify_step', {})
            print(f.split('/')[-1], d['value'], d['ms_per_step'], 'accept', d['config']['mean_accept_len'], 'eq', d['config']['lookahead_equals_greedy'],
                  'attn_ms', vs.get('ms_by_class_events', {}).get('attn'), 'prefill_ms', d['config']['speed_incl_prefill']['prefill_ms'])
PY
tail -3 $OUT/r5c2_mistral_fat_a.err
