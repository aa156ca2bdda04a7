#!/bin/bash
# round 6, call 1: forked-graph-branch weight prefetch (la_lab_set keys 26-30) vs the default step, same box, alternating
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python scripts/gpu_r6_knob_ab.py --steps 48 --reps 2 --out $OUT/r6c1_fork_ab.json \
  --settings "base:|o32:26=32|o64:26=64|o128:26=128|og64:26=64;27=64|og128:26=128;27=128|n2_64:28=64|n2_128:28=128|q64:29=64|q128:29=128|d64:30=64|all64:26=64;28=64;29=64|all128:26=128;28=128;29=128" \
  > $OUT/r6c1_fork_ab.log 2>&1
echo "exit $?" >> $OUT/r6c1_fork_ab.log
grep -E "SUMMARY|exit|Error|error" $OUT/r6c1_fork_ab.log | cut -c1-300
