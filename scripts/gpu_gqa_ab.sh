#!/bin/bash
# GPU tests touching the attention kernels + the GQA batch lines + kernel stats (XCD-aware head map A/B: compare with profiles/r03_kernel_stats_*)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mblock.py tests/test_gpu_kernels.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -4
for leg in "mistral 8" "mixtral 4" "mistral 1"; do
  set -- $leg
  for i in 1 2; do
    timeout 300 python bench.py --model $1 --batch $2 --steps 24 --warmup 4 --no-cpu-baseline --secondary "" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 b$2', d['ms_per_step'], d.get('roofline',{}).get('frac'))"
  done
done
bash scripts/gpu_prof_batch.sh > /dev/null 2>&1
grep -h "tree_attn" $OUT/kernel_stats_*.txt
