#!/bin/bash
# round 6, call 5: review item 1(b) — key-split attention without combine + o_proj merging the partials on load (lab knob 33) vs default,
# and vs the key-split pair with its combine launch (knob 17 = 0)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 900 python scripts/gpu_r6_knob_ab.py --steps 48 --reps 3 --out $OUT/r6c5_merge_ab.json \
  --settings "base:|merge4:33=4|merge2:33=2|pair8:17=0" > $OUT/r6c5_merge_ab.log 2>&1
echo "exit $?" >> $OUT/r6c5_merge_ab.log
grep -E "SUMMARY|exit|Error|error" $OUT/r6c5_merge_ab.log | cut -c1-400
for s in "merge4:33=4" "merge2:33=2"; do
  n=${s%%:*}
  rm -rf /tmp/r6c5_$n
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/r6c5_$n -o run -- python $REPO/scripts/gpu_r6_knob_ab.py --steps 32 --reps 1 --out $OUT/r6c5_prof_$n.json --settings "$s" > $OUT/r6c5_prof_$n.log 2>&1)
  echo "== $n"; python scripts/gpu_r6_trace.py /tmp/r6c5_$n $OUT/r6c5_trace_$n.txt | head -9
done
