#!/bin/bash
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
DIAG_LAYERS=32 timeout 900 python scripts/gpu_diag.py > gpurun_out/diag32.log 2>&1
echo "diag exit $?" >> gpurun_out/diag32.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider --timeout 600 -k "gemm" > gpurun_out/kernels.log 2>&1
echo "kernels exit $?" >> gpurun_out/kernels.log
timeout 900 python scripts/gpu_tune.py small gemm > gpurun_out/tune.log 2>&1
echo "tune exit $?" >> gpurun_out/tune.log
BENCH_DEBUG=1 timeout 900 python bench.py --steps 24 --warmup 2 --no-cpu-baseline > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
tail -20 gpurun_out/diag32.log; tail -3 gpurun_out/kernels.log; tail -12 gpurun_out/bench.log
