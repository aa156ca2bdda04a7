#!/bin/bash
# rocprofv3 kernel stats of the batch configurations (secondary legs of bench.py)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
for leg in "mixtral 4" "mistral 8" "13b 4"; do
  set -- $leg
  RAW=/tmp/la_prof_$1; rm -rf $RAW; mkdir -p $RAW
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o run -- bash -c "cd $REPO && BENCH_IS_SECONDARY=1 python bench.py --model $1 --batch $2 --steps 12 --warmup 2 --no-cpu-baseline" > $OUT/prof_$1.log 2>&1 )
  python - "$RAW" "$OUT/kernel_stats_$1_b$2.txt" <<'PY'
import csv, glob, os, sys
raw, out = sys.argv[1], sys.argv[2]
for f in glob.glob(os.path.join(raw, '**', '*kernel_stats*.csv'), recursive=True):
    with open(out, 'w') as fo:
        for r in csv.DictReader(open(f)):
            if r['Name'].startswith(('k_', 'void k_')):
                fo.write(f"{r['Name'][:70]:72s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us min {int(r['MinNs'])/1e3:8.2f} max {int(r['MaxNs'])/1e3:8.2f} total {int(r['TotalDurationNs'])/1e6:9.2f} ms\n")
PY
done
head -22 $OUT/kernel_stats_mixtral_b4.txt
