#!/bin/bash
# rocprofv3 kernel stats of the device trie (k_trie_hier_get / k_trie_patch) at B = 1, 8, 64, 256 + wall time per call
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
rm -rf /tmp/la_trie_prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/la_trie_prof -o run -- python $REPO/scripts/gpu_trie_time.py > $OUT/r3_trie_prof.log 2>&1 )
python - <<'PY' > gpurun_out/r3_trie_kernel_stats.txt 2>&1
import csv, glob, collections
for f in glob.glob('/tmp/la_trie_prof/**/*kernel_trace*.csv', recursive=True):
    # per (kernel, grid size) durations: the grid of k_trie_hier_get = 64 * B threads
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:48]
        if 'trie' not in k: continue
        g = r.get('Grid_Size', r.get('Grid_Size_X', '?'))
        agg[(k, g)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for (k, g), v in sorted(agg.items()):
        v.sort()
        print(f'{k:50s} grid {g:>8s} n {len(v):4d} median {v[len(v)//2]/1e3:9.2f} us min {v[0]/1e3:9.2f} max {v[-1]/1e3:9.2f}')
PY
grep -v "^W2\|^E2" $OUT/r3_trie_prof.log | tail -12
cat $OUT/r3_trie_kernel_stats.txt
