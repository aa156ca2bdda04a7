#!/bin/bash
# round 6, call 4: attention riders with a start delay (key 32) so that the attention's own HBM phase goes first
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 900 python scripts/gpu_r6_knob_ab.py --steps 48 --reps 3 --out $OUT/r6c4_ride_delay_ab.json \
  --settings "base:|r64d2:31=64;32=2|r64d4:31=64;32=4|r64d6:31=64;32=6|r32d4:31=32;32=4|r128d4:31=128;32=4|r64d9:31=64;32=9" > $OUT/r6c4_ride_delay_ab.log 2>&1
echo "exit $?" >> $OUT/r6c4_ride_delay_ab.log
grep -E "SUMMARY|exit|Error|error" $OUT/r6c4_ride_delay_ab.log | cut -c1-300
for s in "r64d4:31=64;32=4"; do
  n=${s%%:*}
  rm -rf /tmp/r6c4_$n
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/r6c4_$n -o run -- python $REPO/scripts/gpu_r6_knob_ab.py --steps 32 --reps 1 --out $OUT/r6c4_prof_$n.json --settings "$s" > $OUT/r6c4_prof_$n.log 2>&1)
  echo "== $n"; python scripts/gpu_r6_trace.py /tmp/r6c4_$n $OUT/r6c4_trace_$n.txt
done
