#!/bin/bash
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probe_stream.hip -o /tmp/la_probe_bin > gpurun_out/probe.log 2>&1
timeout 300 /tmp/la_probe_bin >> gpurun_out/probe.log 2>&1
echo "probe exit $?" >> gpurun_out/probe.log
cat gpurun_out/probe.log
