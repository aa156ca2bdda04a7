#!/bin/bash
# rocprofv3 kernel stats of a short default bench.py run -> gpurun_out/quick_kernel_stats.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD
RAW=/tmp/la_quick
rm -rf $RAW; mkdir -p $REPO/gpurun_out $RAW
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o run -- bash -c "cd $REPO && python bench.py --no-cpu-baseline --steps ${STEPS:-24} --warmup 4 ${BENCH_ARGS:-}" > $REPO/gpurun_out/quick.log 2>&1 )
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/la_quick/stats/**/*kernel_stats*.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open('gpurun_out/quick_kernel_stats.txt', 'w') as fo:
        for r in rows:
            if r['Name'].startswith(('k_', 'void k_')):
                line = f"{r['Name'][:70]:72s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us min {int(r['MinNs'])/1e3:8.2f} max {int(r['MaxNs'])/1e3:8.2f} total {int(r['TotalDurationNs'])/1e6:9.2f} ms"
                fo.write(line + '\n'); print(line)
PY
