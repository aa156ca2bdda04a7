#!/bin/bash
# round 6, session 3, call 3: narrow-head tests again (tail tolerance + fp32 triangulation), wrapper tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_from_hf.py -m gpu -q -k "narrow or wrapper" -s > $OUT/r6b3_new_tests.log 2>&1; echo "new tests exit $?"; tail -25 $OUT/r6b3_new_tests.log
