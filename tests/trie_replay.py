# -*- coding: utf-8 -*-
"""Replay a golden trie trace (tests/golden/trie_trace_*.json, produced by oracle/gen_golden.py from the
reference itself) through any object with the LookaheadCache surface and compare every recorded output."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def trace_files():
    return sorted(glob.glob(os.path.join(GOLDEN, 'trie_trace_*.json')))


def rows_of(mask):
    mask = np.asarray(mask).astype(np.int64)
    return [int(sum(int(v) << j for j, v in enumerate(row))) for row in mask]


def check_get(res, exp, ctx):
    ids, mask, sizes = res
    assert [int(x) for x in ids] == exp['ids'], ctx
    assert list(np.asarray(mask).shape) == exp['shape'], ctx
    assert rows_of(mask) == exp['rows'], ctx
    assert [int(x) for x in sizes] == exp['sizes'], ctx


def replay(cache, trace, has_batch=True, has_par=True, has_one=True):
    n_checked = 0
    for i, op in enumerate(trace['ops']):
        name = op['op']
        ctx = f"op {i}: { {k: v for k, v in op.items() if k != 'out'} }"
        if name == 'put':
            cache.put(list(op['tokens']), branch_length=op['branch_length'], final=op['final'], mode=op['mode'],
                      idx=op['idx'])
        elif name == 'stream_put':
            cache.stream_put(list(op['tokens']), branch_length=op['branch_length'], final=op['final'],
                             mode='output', idx=op['idx'])
        elif name in ('hier_get', 'one_get', 'par_get'):
            if (name == 'one_get' and not has_one) or (name == 'par_get' and not has_par):
                continue
            kw = {k: op[k] for k in ('decoding_length', 'branch_length', 'min_input_size', 'min_output_size',
                                     'mode', 'idx')}
            check_get(getattr(cache, name)(list(op['tokens']), **kw), op['out'], ctx)
            n_checked += 1
        elif name == 'bat_get':
            if not has_batch or (op['decoding_mode'] != 'hier' and not has_one):
                continue
            ids, masks, sizes = cache.bat_get([list(x) for x in op['tokens']], decoding_length=op['decoding_length'],
                                              branch_length=op['branch_length'],
                                              decoding_cursors=list(op['cursors']), mode=op['mode'],
                                              indices=list(range(len(op['tokens']))),
                                              decoding_mode=op['decoding_mode'])
            exp = op['out']
            assert [[int(v) for v in x] for x in ids] == exp['ids'], ctx
            assert list(masks.shape) == exp['shape'], ctx
            assert [rows_of(m) for m in masks] == exp['rows'], ctx
            assert [[int(v) for v in s] for s in sizes] == exp['sizes'], ctx
            n_checked += 1
        elif name == 'reset_input_freqs':
            cache.reset_input_freqs(op['idx'])
        elif name == 'squeeze_branch_counts':
            cache.squeeze_branch_counts()
        elif name == 'fresh':
            cache.fresh()
        elif name == 'limits':
            cache.max_node, cache.max_output_node = op['max_node'], op['max_output_node']
        else:
            raise ValueError(name)
    return n_checked


def load(path):
    with open(path) as f:
        return json.load(f)
