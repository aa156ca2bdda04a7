#!/bin/bash
# round 4 final: the whole GPU suite, smoke, then the record run (default bench with secondaries + rocprofv3 stats / PMC passes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
bash scripts/gpu_r4_full.sh
bash scripts/gpu_r4_record.sh
