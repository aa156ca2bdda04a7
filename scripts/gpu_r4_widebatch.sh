#!/bin/bash
# round 4, call 20: new parity tests — several wide trees per pass, the batch loop with wide per-sample trees, bitwise wide-GEMM schedules
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_mblock.py tests/test_gpu_batch.py -m gpu -q -p no:cacheprovider --timeout 900 -x -k "several_wide or wide_per_sample or wide_schedules or wide_tree" > $OUT/r4_pytest_widebatch.log 2>&1
echo "pytest exit $?" >> $OUT/r4_pytest_widebatch.log
tail -40 $OUT/r4_pytest_widebatch.log | cut -c1-300
