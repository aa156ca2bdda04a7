#!/bin/bash
# round 6, session 3, call 2: output_scores + narrow-head tests first, then the full GPU suite on the product libraries
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_batch.py -m gpu -q -x -k "narrow or scores" > $OUT/r6b2_new_tests.log 2>&1; echo "new tests exit $?"; tail -25 $OUT/r6b2_new_tests.log
timeout 2400 python -m pytest tests -m gpu -q > $OUT/r6b2_pytest_gpu.log 2>&1; echo "suite exit $?"; tail -15 $OUT/r6b2_pytest_gpu.log
