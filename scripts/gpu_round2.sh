#!/bin/bash
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout 900 python scripts/gpu_diag.py > gpurun_out/diag.log 2>&1
echo "diag exit $?" >> gpurun_out/diag.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider --timeout 600 -k "gemm" > gpurun_out/kernels.log 2>&1
echo "kernels exit $?" >> gpurun_out/kernels.log
timeout 1200 python bench.py --steps 32 --warmup 4 > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
tail -30 gpurun_out/diag.log; tail -3 gpurun_out/kernels.log; tail -2 gpurun_out/bench.log
