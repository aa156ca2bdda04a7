#!/bin/bash
# staged-attention session: kernel + e2e parity tests, kernel A/B by context length, step A/B at 512 and 3500 tokens of context
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
OUT=gpurun_out; mkdir -p $OUT
( timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -k "tree_attention or staged_attention or prefetch" 2>&1 | tail -15 ) > $OUT/pytest_attn.log
timeout 200 python scripts/gpu_ab.py attn > $OUT/attn_ab.log 2>&1
timeout 200 python scripts/gpu_pf_ab.py --settings 0:0:0:0,0:0:0:1,0:0:0:0,0:0:0:1 --out $OUT/attn_step_ab_512.json > $OUT/attn_step_ab_512.log 2>&1
timeout 200 python scripts/gpu_pf_ab.py --layers 8 --prompt-len 3500 --settings 0:0:0:0,0:0:0:1,0:0:0:0,0:0:0:1 --out $OUT/attn_step_ab_3500.json > $OUT/attn_step_ab_3500.log 2>&1
tail -5 $OUT/pytest_attn.log; cat $OUT/attn_ab.log | grep -v amdgpu; grep -h "ms_per_step" $OUT/attn_step_ab_512.log $OUT/attn_step_ab_3500.log | cut -c1-260
