#!/bin/bash
# One GPU call: full -m gpu suite, prefetch A/B, default bench with the best setting, rocprofv3 stats + FETCH_SIZE pass.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
( timeout 460 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]}" >> $OUT/pytest_gpu.log
timeout 200 python scripts/gpu_pf_ab.py --out $OUT/pf_ab.json > $OUT/pf_ab.log 2>&1
echo "ab exit $?" >> $OUT/pf_ab.log
export LA_PF_KIB=$(cat $OUT/best_kib 2>/dev/null || echo 0)
export LA_PF_DELAY=$(cat $OUT/best_delay 2>/dev/null || echo 0)
export LA_PF_TAIL=$(cat $OUT/best_tail 2>/dev/null || echo 0)
echo "best: $LA_PF_KIB $LA_PF_DELAY $LA_PF_TAIL" >> $OUT/pf_ab.log
timeout 200 python bench.py --steps 64 --warmup 8 > $OUT/bench_best.json 2> $OUT/bench_best.err
echo "bench exit $?" >> $OUT/bench_best.err
RAW=/tmp/la_prof
rm -rf $RAW; mkdir -p $RAW
BENCH="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-iters 1"
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o run -- bash -c "cd $REPO && $BENCH" > $OUT/prof_stats.log 2>&1 )
echo "stats exit $?" >> $OUT/prof_stats.log
( cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/pmc_FETCH_SIZE -o run -- bash -c "cd $REPO && $BENCH" > $OUT/prof_pmc_FETCH_SIZE.log 2>&1 )
echo "pmc exit $?" >> $OUT/prof_pmc_FETCH_SIZE.log
python - <<'PY'
import csv, glob, os, collections
raw, out = '/tmp/la_prof', 'gpurun_out'
for f in glob.glob(os.path.join(raw, 'stats', '**', '*kernel_stats*.csv'), recursive=True):
    with open(os.path.join(out, 'kernel_stats_summary.txt'), 'w') as fo:
        for r in csv.DictReader(open(f)):
            if r['Name'].startswith(('k_', 'void k_')):
                fo.write(f"{r['Name'][:60]:62s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.2f} us min {int(r['MinNs'])/1e3:7.2f} max {int(r['MaxNs'])/1e3:7.2f} total {int(r['TotalDurationNs'])/1e6:8.2f} ms\n")
for f in glob.glob(os.path.join(raw, 'pmc_FETCH_SIZE', '**', '*counter_collection*.csv'), recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name') == 'FETCH_SIZE':
            k = r['Kernel_Name'][:70]
            agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
    with open(os.path.join(out, 'pmc_FETCH_SIZE_summary.txt'), 'w') as fo:
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            fo.write(f'{k}\t{n}\t{v / n:.1f}\n')
PY
tail -3 $OUT/pytest_gpu.log; tail -4 $OUT/pf_ab.log; head -c 600 $OUT/bench_best.json; echo; head -12 $OUT/kernel_stats_summary.txt
