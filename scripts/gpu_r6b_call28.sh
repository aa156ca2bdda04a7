#!/bin/bash
# round 6, session 3, call 28: bench.py --batch above 8 with the on-GPU trie (two chained engine passes per step) vs the host trie
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
run() {  # tag, args
  timeout 900 python bench.py $2 --steps 16 --warmup 4 --secondary "" --no-cpu-baseline > $OUT/r6b28_$1.log 2>&1
  tail -1 $OUT/r6b28_$1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['config'].get('lookahead_equals_greedy'), d['config'].get('draft_retrieval'), d['config'].get('mean_accept_len'))" || tail -8 $OUT/r6b28_$1.log
}
run mistral16_host "--model mistral --batch 16"
run mistral16_dev "--model mistral --batch 16 --device-trie"
run mistral12_dev "--model mistral --batch 12 --device-trie"
run mistral8_dev "--model mistral --batch 8"
