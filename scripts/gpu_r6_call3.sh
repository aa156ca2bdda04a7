#!/bin/bash
# round 6, call 3: kernel traces of the default step, the attention riders (key 31) and the forked branch (key 26): what the o_proj
# and attention launches take in each form
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
for s in "base:" "ride64:31=64" "ride128:31=128" "fork64:26=64"; do
  n=${s%%:*}
  rm -rf /tmp/r6c3_$n
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/r6c3_$n -o run -- python $REPO/scripts/gpu_r6_knob_ab.py --steps 32 --reps 1 --out $OUT/r6c3_prof_$n.json --settings "$s" > $OUT/r6c3_prof_$n.log 2>&1)
  echo "== $n"; python scripts/gpu_r6_trace.py /tmp/r6c3_$n $OUT/r6c3_trace_$n.txt
done
