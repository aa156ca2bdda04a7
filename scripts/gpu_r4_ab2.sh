#!/bin/bash
# round 4, call 3: attention microbench (variants + stamps), GPU suite on the new build (norm4 default, unrolled tail), A/B of norm4
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
REPO=$PWD
timeout 600 python scripts/gpu_attn1.py > $OUT/r4_attn1_micro.txt 2>&1; cat $OUT/r4_attn1_micro.txt | grep -v amdgpu.ids
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -x > $OUT/r4_pytest_gpu2.log 2>&1
echo "pytest exit $?" >> $OUT/r4_pytest_gpu2.log
tail -12 $OUT/r4_pytest_gpu2.log | cut -c1-300
run() {   # label, LA_DEBUG, extra bench args
  LA_DEBUG="$2" timeout 300 python bench.py --steps ${STEPS:-48} --warmup 6 --no-cpu-baseline --secondary "" --profile-iters 2 $3 > /tmp/ab.json 2> /tmp/ab.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/ab.json'))
    ev = d['roofline']['verify_step'].get('ms_by_class_events', {})
    print(f"[{sys.argv[1]:28s}] {d['ms_per_step']:.4f} ms/step  tok/s {d['value']:.0f}  eq_greedy={d['config'].get('lookahead_equals_greedy')}  events {ev}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('/tmp/ab.err').read()[-600:])
PY
}
for rep in 1 2; do
  run "default (attn1+norm4)" "" ""
  run "norm4 off" "19=0" ""
  run "attn1 off" "17=0" ""
  run "both off (round 3 kernels)" "17=0,19=0" ""
  run "attn1 SL=2" "18=4" ""
  run "attn1 no rotation" "18=1" ""
done | tee $OUT/r4_ab2.txt
rm -rf /tmp/la_prof; mkdir -p /tmp/la_prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/la_prof/stats -o run -- bash -c "cd $REPO && python bench.py --steps 10 --warmup 2 --no-cpu-baseline --secondary '' --profile-iters 1" > $REPO/$OUT/r4_prof_stats2.log 2>&1 )
python - <<'PY' | tee gpurun_out/r4_kernel_stats2.txt
import csv, glob
for f in glob.glob('/tmp/la_prof/stats/**/*kernel_stats*.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r['Name'].startswith(('k_', 'void k_'))]
    for r in rows[:12]:
        print(f"{r['Name'][:60]:62s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.2f} us  min {int(r['MinNs'])/1e3:7.2f}  max {int(r['MaxNs'])/1e3:7.2f} total {int(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
