#!/bin/bash
# round 6, GPU call 7: the workgroup-per-query retrieval — parity tests (traces incl. wide trees, large forests) and the A/B against the
# one-wavefront kernel
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_trie.py -x -q > gpurun_out/r6c7_trie_tests.log 2>&1; echo "trie tests exit $?"
tail -15 gpurun_out/r6c7_trie_tests.log
timeout 600 python scripts/gpu_trie_wg_ab.py > gpurun_out/r6c7_trie_wg_ab.log 2>&1; echo "ab exit $?"
tail -40 gpurun_out/r6c7_trie_wg_ab.log
