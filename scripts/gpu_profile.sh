#!/bin/bash
# rocprofv3 kernel-trace stats of a short bench run + separate PMC passes (FETCH_SIZE / WRITE_SIZE).
# Outputs: gpurun_out/prof_* (scratch); the summaries are copied into profiles/ by hand afterwards.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
BENCH="python bench.py --steps ${STEPS:-12} --warmup 2 --no-cpu-baseline --profile-iters 1"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o run -- bash -c "cd $PWD && $BENCH" > $OUT/prof_stats.log 2>&1 )
echo "stats exit $?" >> $OUT/prof_stats.log
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace -d $OUT/prof_pmc_$C -o run -- bash -c "cd $PWD && $BENCH" > $OUT/prof_pmc_$C.log 2>&1 )
  echo "pmc $C exit $?" >> $OUT/prof_pmc_$C.log
done
find $OUT/prof_stats -name "*kernel_stats*" | head; find $OUT -name "*.csv" | head -20
# keep the merge-back small: drop per-dispatch traces larger than 20 MB
find $OUT -name "*kernel_trace*.csv" -size +20M -delete
python - <<'PY'
import csv, glob, os, collections
out = os.environ.get('OUT', 'gpurun_out')
for f in glob.glob(os.path.join('gpurun_out', 'prof_stats', '**', '*kernel_stats*.csv'), recursive=True):
    print('==', f)
    rows = list(csv.DictReader(open(f)))
    for r in rows[:25]:
        print({k: r[k] for k in list(r)[:7]})
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(os.path.join('gpurun_out', f'prof_pmc_{c}', '**', '*counter_collection*.csv'), recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') == c:
                k = r['Kernel_Name'][:60]
                agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
        print('==', c, f)
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
            print(f'{k:60s} n={n:6d} mean={v / n:14.1f}')
        with open(os.path.join('gpurun_out', f'pmc_{c}_summary.txt'), 'w') as fo:
            for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                fo.write(f'{k}\t{n}\t{v / n:.1f}\n')
PY
