#!/bin/bash
# round 6, call 2: weight-prefetch RIDERS on the single-launch attention (la_lab_set key 31) vs default; kernel trace of both
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python scripts/gpu_r6_knob_ab.py --steps 48 --reps 3 --out $OUT/r6c2_ride_ab.json \
  --settings "base:|ride16:31=16|ride32:31=32|ride64:31=64|ride128:31=128" > $OUT/r6c2_ride_ab.log 2>&1
echo "exit $?" >> $OUT/r6c2_ride_ab.log
grep -E "SUMMARY|exit|Error|error" $OUT/r6c2_ride_ab.log | cut -c1-300
for s in "base:" "ride64:31=64" "ride128:31=128"; do
  n=${s%%:*}
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/r6c2_prof_$n -o r6c2_$n -- python $GRAFT_REPO_ROOT/scripts/gpu_r6_knob_ab.py --steps 32 --reps 1 --out $GRAFT_REPO_ROOT/$OUT/r6c2_prof_$n.json --settings "$s" > $GRAFT_REPO_ROOT/$OUT/r6c2_prof_$n.log 2>&1)
  f=$(find $OUT/r6c2_prof_$n -name "*kernel_stats.csv" | head -1)
  echo "== $n ($f)"; head -12 "$f" | cut -c1-200
done
