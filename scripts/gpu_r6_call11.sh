#!/bin/bash
# round 6, GPU call 11: chain-form ordering (no level steps) + LDS-only barriers in the expansion; all four path mixes of the workgroup kernel
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_trie.py tests/test_gpu_e2e.py tests/test_gpu_batch.py -x -q > gpurun_out/r6c11_tests.log 2>&1; echo "tests exit $?"
tail -5 gpurun_out/r6c11_tests.log
timeout 600 python scripts/gpu_trie_wg_ab.py > gpurun_out/r6c11_trie_wg_ab.log 2>&1; echo "ab exit $?"
grep "^wg\|^wave\|^host\|^forest" gpurun_out/r6c11_trie_wg_ab.log
for rep in 1 2; do
for leg in "--host-trie" ""; do
  tag=$(echo "m8$leg" | tr -d ' -')
  timeout 900 python bench.py --model mistral --batch 8 --secondary "" $leg > gpurun_out/r6c11_bench_${tag}_$rep.log 2>&1
  echo "== $tag rep $rep: $(tail -1 gpurun_out/r6c11_bench_${tag}_$rep.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config'].get('draft_retrieval'), d['config'].get('lookahead_equals_greedy'), d['config'].get('mean_accept_len'))")"
done; done
