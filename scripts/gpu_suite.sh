#!/bin/bash
# full -m gpu suite (no -x), summary in gpurun_out/pytest_gpu.log
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
( timeout ${SUITE_TIMEOUT:-600} python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} 2>&1 | tail -60 ) > gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
