#!/bin/bash
# round 4, call 10: attention microbench with the speculative touch, e2e A/B (spec touch, write-through slabs), per-predecessor durations
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
REPO=$PWD
timeout 600 python scripts/gpu_attn1.py > $OUT/r4_attn1_micro2.txt 2>&1; grep -v amdgpu.ids $OUT/r4_attn1_micro2.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider --timeout 900 -x > $OUT/r4_pytest_k.log 2>&1
echo "pytest exit $?" >> $OUT/r4_pytest_k.log; tail -3 $OUT/r4_pytest_k.log | cut -c1-300
LA_LAB_SET="23=1" timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider --timeout 900 -x -k "match_oracle or golden or 7b_shape" > $OUT/r4_pytest_wt.log 2>&1
echo "slab_wt tests: $(tail -1 $OUT/r4_pytest_wt.log)"
run() {
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
for rep in 1 2 3; do
  run "default (spec touch)" "" ""
  run "no spec touch" "18=8" ""
  run "slabs write-through" "23=1" ""
done | tee $OUT/r4_ab3.txt
rm -rf /tmp/la_prof; mkdir -p /tmp/la_prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/la_prof/stats -o run -- bash -c "cd $REPO && python bench.py --steps 10 --warmup 2 --no-cpu-baseline --secondary '' --profile-iters 1" > $REPO/$OUT/r4_prof_stats3.log 2>&1 )
python - <<'PY' | tee gpurun_out/r4_kernel_by_predecessor.txt
import csv, glob, collections
for f in glob.glob('/tmp/la_prof/stats/**/*kernel_trace*.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    prev, agg2 = None, collections.defaultdict(lambda: [0, 0, 0])
    pend = 0
    for r in rows:
        k = r['Kernel_Name'][:40]
        if k.startswith(('void k_row_norm<', 'void k_gemm64<2, 0', 'k_tree_attn1', 'void k_gemm64r')):
            key = (k, (prev['Kernel_Name'] if prev else '')[:34])
            a = agg2[key]
            a[0] += 1; a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
            if prev: a[2] += int(r['Start_Timestamp']) - int(prev['End_Timestamp'])
        prev = r
    for (k, pv), (n, t, g) in sorted(agg2.items(), key=lambda kv: -kv[1][1]):
        if n >= 50:
            print(f'{k:42s} after {pv:36s} n={n:6d} avg {t / n / 1e3:6.2f} us  gap before {g / n / 1e3:5.2f} us')
PY
