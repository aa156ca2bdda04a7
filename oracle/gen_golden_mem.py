# -*- coding: utf-8 -*-
"""ORACLE tooling (build container only): a trie snapshot written by the REFERENCE's own save_mem
(lookahead_cache.py:578-582) plus retrievals recorded from the reference after load_mem, for the importer test
(LookaheadCache.load_reference_mem).  Writes tests/golden/ref_mem_sample.json and ref_mem_queries.json."""
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference/lookahead')
from lookahead.common.lookahead_cache import LookaheadCache  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
rs = np.random.RandomState(21)
cache = LookaheadCache(eos_ids=[2])
phrases = [rs.randint(3, 60, size=rs.randint(3, 9)).tolist() for _ in range(25)]


def text(n):
    out = []
    while len(out) < n:
        out.extend(phrases[rs.randint(0, len(phrases))])
    return out[:n]


for i in range(4):
    cache.put(text(36), branch_length=7, mode='output', idx=-1)
for i in range(2):
    cache.put(text(24), branch_length=7, mode='input', idx=i)
cache.stream_put(text(20), branch_length=7, mode='output', idx=1, final=False)
path = os.path.join(OUT, 'ref_mem_sample.json')
cache.save_mem(path)
fresh = LookaheadCache(eos_ids=[2])
fresh.load_mem(path)
queries = []
for _ in range(40):
    q = text(80)[-2:]
    idx = int(rs.randint(0, 3))
    mode = ['input', 'output', 'mix'][rs.randint(0, 3)]
    dl = int(rs.choice([8, 16, 64]))
    ids, mask, sizes = fresh.hier_get(q, decoding_length=dl, branch_length=8, min_input_size=0, min_output_size=max(dl // 2, 1),
                                      mode=mode, idx=idx)
    queries.append({'q': [int(x) for x in q], 'idx': idx, 'mode': mode, 'dl': dl, 'ids': [int(x) for x in ids],
                    'rows': [int(sum(int(b) << j for j, b in enumerate(r))) for r in np.asarray(mask)], 'sizes': [int(x) for x in sizes]})
json.dump({'n_trees': len(fresh.mem), 'queries': queries}, open(os.path.join(OUT, 'ref_mem_queries.json'), 'w'), separators=(',', ':'))
print('trees', len(fresh.mem), 'file bytes', os.path.getsize(path), 'non-trivial drafts', sum(len(x['ids']) > 1 for x in queries))
