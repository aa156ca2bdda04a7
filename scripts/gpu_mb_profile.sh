#!/bin/bash
# rocprofv3 kernel stats of the multi-block step benchmark (scripts/gpu_mb_bench.py); summary -> gpurun_out/
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
RAW=/tmp/la_mbprof
rm -rf $RAW; mkdir -p $OUT $RAW
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o run -- bash -c "cd $REPO && python scripts/gpu_mb_bench.py ${MB_ARGS:-}" > $OUT/mbprof.log 2>&1 )
python - <<'PY'
import csv, glob, os
for f in glob.glob('/tmp/la_mbprof/stats/**/*kernel_stats*.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open('gpurun_out/mb_kernel_stats.txt', 'w') as fo:
        for r in rows:
            if r['Name'].startswith(('k_', 'void k_')):
                line = f"{r['Name'][:70]:72s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us min {int(r['MinNs'])/1e3:8.2f} max {int(r['MaxNs'])/1e3:8.2f} total {int(r['TotalDurationNs'])/1e6:9.2f} ms"
                fo.write(line + '\n'); print(line)
PY
