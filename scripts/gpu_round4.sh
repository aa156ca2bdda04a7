#!/bin/bash
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 600 python scripts/gpu_tune.py gemm > gpurun_out/tune.log 2>&1
echo "tune exit $?" >> gpurun_out/tune.log
BENCH_DEBUG=1 timeout 900 python bench.py --steps 48 --warmup 4 > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "bench exit $?" >> gpurun_out/bench.log
STEPS=10 bash scripts/gpu_profile.sh > gpurun_out/profile.log 2>&1
tail -15 gpurun_out/pytest_gpu.log; grep -E "rb=2 ks=(1|4) var=0|gateup|lm_head" gpurun_out/tune.log | head -20; tail -8 gpurun_out/bench.err; tail -3 gpurun_out/bench.log; tail -60 gpurun_out/profile.log
