#!/bin/bash
# One gpurun call: kernel parity, e2e parity, smoke, short bench. Logs land in gpurun_out/.
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/dev.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider --timeout 600 > gpurun_out/kernels.log 2>&1
echo "kernels exit $?" >> gpurun_out/kernels.log
timeout 1800 python -m pytest tests/test_gpu_e2e.py -q -p no:cacheprovider --timeout 900 -k "not full_size" > gpurun_out/e2e.log 2>&1
echo "e2e exit $?" >> gpurun_out/e2e.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 1200 python bench.py --steps 32 --warmup 4 > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
tail -5 gpurun_out/kernels.log; tail -5 gpurun_out/e2e.log; tail -3 gpurun_out/smoke.log; tail -3 gpurun_out/bench.log
