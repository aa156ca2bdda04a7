#!/bin/bash
# Mistral-7B bs=8: host trie vs chained device trie with the per-step update on the host (patch) vs on the device
# (la_trie_stream_put_dev), twice each in one call; then rocprofv3 kernel stats of the device-update run (trie kernels only)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
run() {
  timeout 300 python bench.py --model mistral --batch 8 --steps 24 --warmup 4 --no-cpu-baseline --secondary "" "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$*', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'), d['config'].get('device_trie_stats'))"
}
for i in 1 2; do
  run
  run --device-trie --host-trie-update
  run --device-trie
done
RAW=/tmp/la_prof_put; rm -rf $RAW; mkdir -p $RAW
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o run -- bash -c "cd $REPO && python bench.py --model mistral --batch 8 --steps 24 --warmup 4 --no-cpu-baseline --secondary '' --device-trie" > $OUT/prof_put.log 2>&1 )
python - "$RAW" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], '**', '*kernel_stats*.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'trie' in r['Name'] or 'fill_from' in r['Name']:
            print(f"{r['Name'][:60]:62s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us min {int(r['MinNs'])/1e3:8.2f} max {int(r['MaxNs'])/1e3:8.2f}")
PY
