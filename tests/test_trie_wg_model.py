# -*- coding: utf-8 -*-
"""CPU: the level-synchronous retrieval algorithm of the workgroup-per-query device kernel (tests/trie_wg_model.py = the model
csrc/la_trie_wg.hip::k_trie_hier_get_wg implements, level form and chain form) replayed over every golden reference trace — every recorded hier_get, trees wider
than 64 rows included — and differentially against the native host trie on a large forest with dead (reset) input nodes."""
import os
import random

import numpy as np
import pytest

from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
from tests import trie_replay as tr
from tests import trie_wg_model as wg


@pytest.mark.parametrize('form', ['levels', 'chains'])
@pytest.mark.parametrize('path', tr.trace_files(), ids=os.path.basename)
def test_level_synchronous_model_replays_reference_trace(path, form):
    trace = tr.load(path)
    init = trace['init']
    cache = LookaheadCache(eos_ids=init['eos_ids'], stop_words={w: 1 for w in init['stop_words']},
                           max_node=init['max_node'], max_output_node=init['max_output_node'])
    checked = wide = 0
    for i, op in enumerate(trace['ops']):
        name = op['op']
        if name == 'put':
            cache.put(list(op['tokens']), branch_length=op['branch_length'], final=op['final'], mode=op['mode'], idx=op['idx'])
        elif name == 'stream_put':
            cache.stream_put(list(op['tokens']), branch_length=op['branch_length'], final=op['final'], idx=op['idx'])
        elif name == 'hier_get':
            if checked >= 40 and op['decoding_length'] <= 64:
                continue
            img = wg.image_of(cache, op['idx'])
            got = wg.hier_get(img, list(op['tokens']), op['decoding_length'], op['branch_length'], op['min_input_size'], op['min_output_size'],
                              op['mode'], stop_words=init['stop_words'], form=form)
            exp = op['out']
            ctx = f"op {i}: { {k: v for k, v in op.items() if k != 'out'} }"
            assert got[0] == exp['ids'], ctx
            assert got[1] == exp['rows'][:len(exp['ids'])], ctx
            assert [int(x) for x in got[2]] == exp['sizes'], ctx
            checked += 1
            wide += op['decoding_length'] > 64
        elif name == 'reset_input_freqs':
            cache.reset_input_freqs(op['idx'])
        elif name == 'squeeze_branch_counts':
            cache.squeeze_branch_counts()
        elif name == 'fresh':
            cache.fresh()
        elif name == 'limits':
            cache.max_node, cache.max_output_node = op['max_node'], op['max_output_node']
    assert checked >= 10


@pytest.mark.parametrize('form', ['levels', 'chains'])
def test_level_synchronous_model_equals_host_trie_with_dead_nodes_cutoffs_and_wide_trees(form):
    """large forest (100 x 256-token warm-up), an input-mode prompt in plane 0 and a SECOND prompt whose input frequencies were reset
    (dead nodes that the cut-off rule still emits when the subtree fits the budget): 120 queries x {mix, output, input} x budgets
    {16, 64, 128, 256}."""
    rng = random.Random(1)
    nr = np.random.RandomState(1)
    cache = LookaheadCache(eos_ids=[None])
    phrases = [nr.randint(3, 2000, size=nr.randint(3, 10)).tolist() for _ in range(200)]
    for _ in range(100):
        seq = []
        while len(seq) < 256:
            seq.extend(phrases[min(int(nr.zipf(1.3)) - 1, 199)])
        cache.put(seq[:256], branch_length=13, mode='output', idx=-1)
    dead = sum((phrases[rng.randrange(200)] + [rng.randrange(3, 2000)] for _ in range(40)), [])
    cache.put(dead, branch_length=13, mode='input', idx=0)
    cache.reset_input_freqs(0)                                   # leaves dead input-only nodes behind (no squeeze yet)
    prompt = sum((phrases[rng.randrange(200)] for _ in range(60)), [])
    cache.put(prompt, branch_length=13, mode='input', idx=0)
    img = wg.image_of(cache, 0)
    queries = []
    for _ in range(120):
        ph = phrases[rng.randrange(200)]
        k = rng.randrange(1, len(ph))
        queries.append(ph[max(0, k - 2):k] if rng.random() < 0.8 else [rng.randrange(3, 2000), rng.randrange(3, 2000)])
    n_wide = 0
    for mode, mi, mo in [('mix', 0, 32), ('mix', 2, 8), ('output', 0, 16), ('input', 1, 0)]:
        for dl, bl in [(16, 6), (64, 12), (128, 32), (256, 20)]:
            for qy in queries[:60 if dl > 64 else 120]:
                ids, mask, sizes = cache.hier_get(qy, decoding_length=dl, branch_length=bl, min_input_size=mi, min_output_size=mo if dl <= 64 else dl // 2, mode=mode, idx=0)
                got = wg.hier_get(img, qy, dl, bl, mi, mo if dl <= 64 else dl // 2, mode, form=form)
                assert got[0] == [int(x) for x in ids], (mode, dl, qy)
                assert got[1] == tr.rows_of(mask)[:len(got[0])], (mode, dl, qy)
                assert [int(x) for x in got[2]] == [int(x) for x in sizes], (mode, dl, qy)
                n_wide += len(got[0]) > 64
    assert n_wide > 20
