#!/bin/bash
# round 6, GPU call 8: workgroup kernel after the pass trimming (parity + A/B), then the Mistral-7B bs=8 step with host trie / device trie (both kernels)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_trie.py -x -q > gpurun_out/r6c8_trie_tests.log 2>&1; echo "trie tests exit $?"
tail -5 gpurun_out/r6c8_trie_tests.log
timeout 600 python scripts/gpu_trie_wg_ab.py > gpurun_out/r6c8_trie_wg_ab.log 2>&1; echo "ab exit $?"
grep -v "^RESULT\|amdgpu.ids" gpurun_out/r6c8_trie_wg_ab.log
for rep in 1 2; do
for leg in "" "--device-trie --trie-algo wave" "--device-trie --trie-algo wg"; do
  tag=$(echo "host$leg" | tr -d ' -')
  timeout 900 python bench.py --model mistral --batch 8 --secondary "" $leg > gpurun_out/r6c8_bench_${tag}_$rep.log 2>&1
  echo "== $tag rep $rep: $(tail -1 gpurun_out/r6c8_bench_${tag}_$rep.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config'].get('draft_retrieval'), d['config'].get('lookahead_equals_greedy'), d['config'].get('mean_accept_len'))")"
done; done
