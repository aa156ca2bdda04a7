#!/bin/bash
# round 3 baseline: gpu tests, device-trie kernel stats (rocprofv3), short default bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/r3_pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/r3_pytest_gpu.log
rm -rf /tmp/la_trie_prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/la_trie_prof -o run -- python $REPO/scripts/gpu_trie_time.py > $OUT/r3_trie_prof.log 2>&1 )
python - <<'PY' > gpurun_out/r3_trie_kernel_stats.txt 2>&1
import csv, glob
for f in glob.glob('/tmp/la_trie_prof/**/*kernel_stats*.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:70]:72s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us min {int(r['MinNs'])/1e3:9.2f} max {int(r['MaxNs'])/1e3:9.2f}")
PY
timeout 600 python bench.py --steps 20 --warmup 5 --secondary "" > $OUT/r3_bench_base.json 2> $OUT/r3_bench_base.err; echo "bench exit $?" >> $OUT/r3_bench_base.err
