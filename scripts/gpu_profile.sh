#!/bin/bash
# rocprofv3 kernel-trace stats of a short bench run + separate PMC passes (FETCH_SIZE / WRITE_SIZE).
# Raw traces stay in /tmp on the GPU box; only the small summaries land in gpurun_out/ (64 MiB merge limit),
# from where they are copied into profiles/ by hand.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
RAW=/tmp/la_prof
rm -rf $RAW; mkdir -p $OUT $RAW
BENCH="python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --profile-iters 1 --secondary \"\""
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o run -- bash -c "cd $REPO && $BENCH" > $OUT/prof_stats.log 2>&1 )
echo "stats exit $?" >> $OUT/prof_stats.log
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $RAW/pmc_$C -o run -- bash -c "cd $REPO && $BENCH" > $OUT/prof_pmc_$C.log 2>&1 )
  echo "pmc $C exit $?" >> $OUT/prof_pmc_$C.log
done
# MFMA / wave-state counters (SQ block, own pass): evidence that the GEMMs are HBM-bound, not matrix-core-bound
( cd /tmp && timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $RAW/pmc_SQ -o run -- bash -c "cd $REPO && $BENCH" > $OUT/prof_pmc_SQ.log 2>&1 )
echo "pmc SQ exit $?" >> $OUT/prof_pmc_SQ.log
find $RAW -type f | head -30
for f in $(find $RAW/stats -name "*kernel_stats*.csv" -o -name "*stats*.csv" | head -5); do cp $f $OUT/; done
python - <<'PY'
import csv, glob, os, collections
raw = '/tmp/la_prof'
out = 'gpurun_out'
for f in glob.glob(os.path.join(raw, 'stats', '**', '*kernel_stats*.csv'), recursive=True):
    print('==', f)
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if r['Name'].startswith(('k_', 'void k_')):
            print(f"{r['Name'][:52]:54s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.2f} us min {int(r['MinNs'])/1e3:7.2f} max {int(r['MaxNs'])/1e3:7.2f} total {int(r['TotalDurationNs'])/1e6:8.2f} ms")
# per-kernel mean duration from the trace (ns), steady-state only is not separable here: report all
for f in glob.glob(os.path.join(raw, 'stats', '**', '*kernel_trace*.csv'), recursive=True):
    agg = collections.defaultdict(lambda: [0, 0])
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        agg[k][0] += 1; agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    with open(os.path.join(out, 'kernel_trace_summary.txt'), 'w') as fo:
        for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            fo.write(f'{k}\t{n}\t{t / n / 1e3:.2f}us\t{t / 1e6:.3f}ms\n')
# the two norm launches and the two slab GEMMs of a layer carry the same kernel name: split them by their position in the step
for f in glob.glob(os.path.join(raw, 'stats', '**', '*kernel_trace*.csv'), recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    prev, agg2 = None, collections.defaultdict(lambda: [0, 0])
    for r in rows:
        k = r['Kernel_Name'][:40]
        if k.startswith(('void k_row_norm<', 'void k_gemm64<2, 0')):
            key = (k, (prev or '')[:34])
            agg2[key][0] += 1; agg2[key][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        prev = r['Kernel_Name']
    with open(os.path.join(out, 'kernel_by_predecessor.txt'), 'w') as fo:
        for (k, pv), (n, t) in sorted(agg2.items(), key=lambda kv: -kv[1][1]):
            if n >= 50:
                line = f'{k:42s} after {pv:36s} n={n:6d} avg {t / n / 1e3:6.2f} us'
                fo.write(line + '\n'); print(line)
for f in glob.glob(os.path.join(raw, 'pmc_SQ', '**', '*counter_collection*.csv'), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        a = agg[k][r.get('Counter_Name')]
        a[0] += 1; a[1] += float(r['Counter_Value'])
    print('== SQ', f)
    with open(os.path.join(out, 'pmc_SQ_summary.txt'), 'w') as fo:
        for k, cs in agg.items():
            if not k.startswith(('k_', 'void k_')):
                continue
            m = {c: v[1] / max(v[0], 1) for c, v in cs.items()}
            # raw per-dispatch means; utilisation is derived in profiles/pmc_latest.json from the kernel's duration
            # (SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs)); GRBM_GUI_ACTIVE is summed over the 8 XCDs
            line = f"{k[:52]:54s} " + ' '.join(f'{c}={v:.0f}' for c, v in sorted(m.items()))
            fo.write(line + '\n'); print(line)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(os.path.join(raw, f'pmc_{c}', '**', '*counter_collection*.csv'), recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') == c:
                k = r['Kernel_Name'][:70]
                agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
        print('==', c, f)
        with open(os.path.join(out, f'pmc_{c}_summary.txt'), 'w') as fo:
            for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                fo.write(f'{k}\t{n}\t{v / n:.1f}\n')
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            if k.startswith(('k_', 'void k_')):
                print(f'{k[:52]:54s} n={n:6d} mean={v / n:14.1f}')
PY
du -sh $OUT
