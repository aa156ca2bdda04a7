# -*- coding: utf-8 -*-
"""Pin the ORACLE (oracle/trie_oracle.py) against golden vectors recorded from the reference itself."""
import json
import os

import pytest

from oracle.trie_oracle import TrieOracle
from tests import trie_replay as tr


@pytest.mark.parametrize('path', tr.trace_files(), ids=os.path.basename)
def test_oracle_replays_reference_trace(path):
    trace = tr.load(path)
    init = trace['init']
    cache = TrieOracle(eos_ids=init['eos_ids'], stop_words={w: 1 for w in init['stop_words']},
                       max_node=init['max_node'], max_output_node=init['max_output_node'])
    n = tr.replay(cache, trace, has_batch=True, has_par=True, has_one=True)
    assert n > 50
    assert cache.n_trees() == trace['final']['n_trees']
    assert cache.n_nodes() == trace['final']['n_nodes']


def test_oracle_reference_unit_tests():
    """lookahead/tests/test_lookahead_cache.py:16-45 (Tree(1).put(...); get([1]))."""
    kats = json.load(open(os.path.join(tr.GOLDEN, 'trie_kats.json')))
    for kat in kats['reference_tests']:
        cache = TrieOracle(eos_ids=None)
        for p in kat['puts']:
            cache.put([9] + list(p), branch_length=8, mode='output', idx=-1)   # tree 9 <- p, like Tree(1).put(p)
        tr.check_get(cache.hier_get([9, 1], decoding_length=63, branch_length=3), kat['out'], kat['name'])


def test_oracle_t64b8():
    kat = json.load(open(os.path.join(tr.GOLDEN, 'trie_kats.json')))['t64b8']
    cache = TrieOracle(eos_ids=None)
    for p in kat['puts']:
        cache.put(list(p), branch_length=13, mode='output', idx=-1)
    res = cache.hier_get(kat['query'], decoding_length=64, branch_length=12, min_input_size=0, min_output_size=32)
    tr.check_get(res, kat['out'], 't64b8')
    assert len(res[0]) == 64 and res[2] == [0, 63]
