#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python scripts/gpu_wide_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_wide_stamps.txt
