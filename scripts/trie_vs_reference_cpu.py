# -*- coding: utf-8 -*-
"""Build-container only (needs /root/reference): the reference's perf_check_trie loop (benchmarks/benchmark.py:353-395)
on the reference's Python LookaheadCache and on the native trie, same synthetic corpus (SURVEY §6 probe shape)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from bench import phrase_prompt  # noqa: E402
from painlessinferenceacceleration_amd.benchmark import Benchmark  # noqa: E402
from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache  # noqa: E402

sys.path.insert(0, '/root/reference/lookahead')
from lookahead.common.lookahead_cache import LookaheadCache as RefCache  # noqa: E402

V = 32000
warm = [phrase_prompt(10 + i, 256, V) for i in range(100)]
ins = [phrase_prompt(500 + i, 512, V) for i in range(40)]
outs = warm[:40]            # the benchmark warms the trie with the answers of the same queries: lookups hit
res = {}
for name, cache in (('reference_python', RefCache(eos_ids=[2])), ('native', LookaheadCache(eos_ids=[2]))):
    res[name] = Benchmark.perf_check_trie(cache, warm, ins, outs, max_node_rate=16, decoding_length=64, branch_length=12, edl=8)
res['speedup_put'] = res['reference_python']['put_us_per_token'] / res['native']['put_us_per_token']
res['speedup_get'] = res['reference_python']['get_ms_per_query'] / res['native']['get_ms_per_query']
print(json.dumps(res, indent=1))
