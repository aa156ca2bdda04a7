#!/bin/bash
# round 6, call 6: review item 1(c) priced by a TIMING PROBE (lab knob 34; numerics garbage): o_proj at full K (1 split) / 2 splits with one
# row-block per workgroup, post-attention norm launch skipped = upper bound of what a norm folded into gate/up could buy
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 900 python scripts/gpu_r6_knob_ab.py --steps 48 --reps 3 --out $OUT/r6c6_fullk_probe_ab.json \
  --settings "base:|nonorm:34=32|fullk128wg:34=49|fullk128wg_norm:34=17|ks2_256wg:34=50|ks2_256wg_norm:34=18|ks2_rb2:34=34" > $OUT/r6c6_fullk_probe_ab.log 2>&1
echo "exit $?" >> $OUT/r6c6_fullk_probe_ab.log
grep -E "SUMMARY|exit|Error|error" $OUT/r6c6_fullk_probe_ab.log | cut -c1-330
for s in "fullk128wg:34=49" "ks2_256wg:34=50"; do
  n=${s%%:*}
  rm -rf /tmp/r6c6_$n
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/r6c6_$n -o run -- python $REPO/scripts/gpu_r6_knob_ab.py --steps 32 --reps 1 --out $OUT/r6c6_prof_$n.json --settings "$s" > $OUT/r6c6_prof_$n.log 2>&1)
  echo "== $n"; python scripts/gpu_r6_trace.py /tmp/r6c6_$n $OUT/r6c6_trace_$n.txt | head -9
done
