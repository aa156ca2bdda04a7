# -*- coding: utf-8 -*-
"""ORACLE tooling: generate golden vectors by running the REFERENCE itself (build container only).

Imports /root/reference/lookahead/lookahead/common/lookahead_cache.py in place (numpy only, nothing
is copied) and records seeded operation traces with the reference's outputs into
tests/golden/trie_*.json, plus the reference's own known-answer tests
(lookahead/tests/test_lookahead_cache.py:16-45).  The accept-scan / llama vectors are produced by
oracle/gen_golden_model.py.  /root/reference does not exist on the GPU box: tests only read the
committed JSON.

    python oracle/gen_golden.py            # rewrites tests/golden/trie_*.json
"""
import importlib.util
import json
import os
import random
import sys

import numpy as np

REF = '/root/reference/lookahead/lookahead/common/lookahead_cache.py'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def load_reference():
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location('ref_lookahead_cache', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def rows_of(mask):
    mask = np.asarray(mask)
    return [int(sum(int(v) << j for j, v in enumerate(row))) for row in mask.astype(np.int64)]


def enc(res):
    ids, mask, sizes = res
    return {'ids': [int(x) for x in ids], 'rows': rows_of(mask), 'shape': list(np.asarray(mask).shape),
            'sizes': [int(x) for x in sizes]}


def make_trace(ref, seed, vocab, n_ops, max_node=65536, max_output_node=512, eos=(2,), stop_words=(),
               zipf=1.2, branch_choices=(4, 8, 12), dl_choices=(2, 8, 16, 64), with_batch=True):
    rng = random.Random(seed)
    nrng = np.random.RandomState(seed)
    cache = ref.LookaheadCache(eos_ids=list(eos) if eos is not None else None,
                               stop_words={w: 1 for w in stop_words}, max_node=max_node,
                               max_output_node=max_output_node)
    # phrase bank => repeated n-grams (so drafts are non-trivial), Zipf-picked
    phrases = [[rng.randrange(0, vocab) for _ in range(rng.randint(2, 9))] for _ in range(max(8, vocab // 4))]
    weights = np.array([1.0 / (i + 1) ** zipf for i in range(len(phrases))])
    weights /= weights.sum()

    def seq(n):
        out = []
        while len(out) < n:
            out.extend(phrases[nrng.choice(len(phrases), p=weights)])
        return out[:n]

    ops = []
    history = seq(40)
    for _ in range(n_ops):
        r = rng.random()
        if r < 0.18:
            toks = seq(rng.randint(0, 60))
            kw = dict(branch_length=rng.choice(branch_choices), final=rng.random() < 0.15,
                      mode=rng.choice(['input', 'output']), idx=rng.choice([0, 0, 1, 3, -1]))
            if kw['mode'] == 'output' and rng.random() < 0.5:
                kw['idx'] = -1
            cache.put(list(toks), **kw)
            ops.append({'op': 'put', 'tokens': toks, **kw})
            history = (history + toks)[-80:]
        elif r < 0.42:
            toks = seq(rng.randint(0, 14))
            kw = dict(branch_length=rng.choice(branch_choices), final=rng.random() < 0.1, idx=rng.choice([0, 0, 1, 2]))
            cache.stream_put(list(toks), mode='output', **kw)
            ops.append({'op': 'stream_put', 'tokens': toks, **kw})
            history = (history + toks)[-80:]
        elif r < 0.90:
            # query with a suffix of something seen (hit) or random tokens (miss)
            if rng.random() < 0.8 and len(history) >= 3:
                p = rng.randrange(1, len(history))
                q = history[max(0, p - rng.randint(1, 3)):p]
            else:
                q = [rng.randrange(0, vocab) for _ in range(rng.randint(0, 3))]
            dl = rng.choice(dl_choices)
            kw = dict(decoding_length=dl, branch_length=rng.choice(branch_choices + (0,)),
                      min_input_size=rng.choice([0, 0, 0, 1, 2]),
                      min_output_size=rng.choice([0, max(dl // 2, 1), max(dl // 2, 1), 1]),
                      mode=rng.choice(['mix', 'mix', 'mix', 'input', 'output']), idx=rng.choice([0, 0, 1, 3]))
            kind = rng.random()
            name = 'hier_get' if kind < 0.78 else 'one_get' if kind < 0.90 else 'par_get'
            try:
                res = getattr(cache, name)(list(q), **kw)
            except (IndexError, ValueError):
                continue      # the reference itself raises (e.g. par_get on an empty query): not a parity case
            ops.append({'op': name, 'tokens': q, **kw, 'out': enc(res)})
        elif r < 0.95 and with_batch:
            bs = rng.choice([1, 2, 4])
            qs, cursors = [], []
            for _b in range(bs):
                p = rng.randrange(1, len(history)) if len(history) > 2 else 1
                qs.append(history[max(0, p - 2):p] or [rng.randrange(0, vocab)])
                cursors.append(rng.randint(5, 40))
            kw = dict(decoding_length=rng.choice([16, 32, 64]), branch_length=rng.choice(branch_choices),
                      mode=rng.choice(['mix', 'output']), decoding_mode=rng.choice(['hier', 'hier', 'one']))
            ids, masks, sizes = cache.bat_get([list(x) for x in qs], decoding_cursors=list(cursors),
                                              indices=list(range(bs)), **kw)
            ops.append({'op': 'bat_get', 'tokens': qs, 'cursors': cursors, **kw,
                        'out': {'ids': [[int(v) for v in x] for x in ids], 'shape': list(masks.shape),
                                'rows': [[int(sum(int(v) << j for j, v in enumerate(row))) for row in m]
                                         for m in masks.astype(np.int64)],
                                'sizes': [[int(v) for v in s] for s in sizes]}})
        elif r < 0.97:
            cache.reset_input_freqs(0)
            ops.append({'op': 'reset_input_freqs', 'idx': 0})
        elif r < 0.985:
            cache.squeeze_branch_counts()
            ops.append({'op': 'squeeze_branch_counts'})
        elif r < 0.99:
            cache.fresh()
            ops.append({'op': 'fresh'})
        else:
            mn, mo = rng.choice([(65536, 512), (200, 20), (64, 8)])
            cache.max_node, cache.max_output_node = mn, mo
            ops.append({'op': 'limits', 'max_node': mn, 'max_output_node': mo})
    n_nodes = 0
    for t in cache.mem.values():
        sizes = [0]
        t._count_node(t.nodes, sizes)
        n_nodes += sizes[0]
    return {'seed': seed, 'vocab': vocab, 'init': {'eos_ids': list(eos) if eos is not None else None,
                                                    'stop_words': list(stop_words), 'max_node': max_node,
                                                    'max_output_node': max_output_node},
            'ops': ops, 'final': {'n_trees': len(cache.mem), 'n_nodes': n_nodes}}


def reference_kats(ref):
    """The reference's own unit tests, lookahead/tests/test_lookahead_cache.py:16-45, as data."""
    out = []
    t = ref.Tree(1)
    t.put([1, 2, 3, 4], mode='output', idx=-1)
    out.append({'name': 'single_chain', 'puts': [[1, 2, 3, 4]], 'query': [1],
                'out': enc(t.get([1], max_size=63, max_length=8))})
    t = ref.Tree(1)
    t.put([1, 2, 3], mode='output', idx=-1)
    t.put([1, 2, 4], mode='output', idx=-1)
    out.append({'name': 'two_branches', 'puts': [[1, 2, 3], [1, 2, 4]], 'query': [1],
                'out': enc(t.get([1], max_size=63, max_length=8))})
    return out


def t64b8(ref):
    """SURVEY §8d fixed 'T64/B8' draft tree: main 13-gram put twice, 7 side branches once."""
    cache = ref.LookaheadCache(eos_ids=[None])
    q = [100, 101]
    main = list(range(1000, 1012))
    puts = [q + main, q + main]
    forks = [(1, 11), (2, 10), (3, 9), (4, 8), (6, 6), (8, 4), (9, 3)]
    base = 2000
    for depth, length in forks:
        side = main[:depth] + list(range(base, base + length))
        base += 100
        puts.append(q + side)
    for p in puts:
        cache.put(list(p), branch_length=13, mode='output', idx=-1)
    res = cache.hier_get(list(q), decoding_length=64, branch_length=12, min_input_size=0, min_output_size=32,
                         mode='mix', idx=0)
    return {'puts': puts, 'query': q, 'out': enc(res)}


def main():
    ref = load_reference()
    os.makedirs(OUT, exist_ok=True)
    traces = [
        make_trace(ref, seed=1, vocab=12, n_ops=260, max_output_node=512, dl_choices=(2, 4, 8, 16)),
        make_trace(ref, seed=2, vocab=40, n_ops=320, eos=(2,), stop_words=(5, 7)),
        make_trace(ref, seed=3, vocab=300, n_ops=320, eos=(2, 9)),
        make_trace(ref, seed=4, vocab=40, n_ops=300, eos=None, max_node=120, max_output_node=12, zipf=0.6),
        make_trace(ref, seed=5, vocab=6, n_ops=220, dl_choices=(3, 8, 64, 100), branch_choices=(2, 5, 20)),
    ]
    # squeeze trace: >= 1024 dirty trees with tiny limits so that final=True prunes (H1d)
    rng = random.Random(77)
    cache = ref.LookaheadCache(eos_ids=[None], max_node=40, max_output_node=10)
    ops = []
    for r in range(60):
        toks = [rng.randrange(0, 1500) if rng.random() < 0.7 else rng.randrange(0, 30) for _ in range(64)]
        fin = (r % 7 == 6)
        cache.stream_put(list(toks), branch_length=6, final=fin, mode='output', idx=0)
        ops.append({'op': 'stream_put', 'tokens': toks, 'branch_length': 6, 'final': fin, 'idx': 0})
        for _ in range(6):
            q = [rng.randrange(0, 30) for _ in range(2)]
            kw = dict(decoding_length=16, branch_length=6, min_input_size=0, min_output_size=8, mode='mix', idx=0)
            ops.append({'op': 'hier_get', 'tokens': q, **kw, 'out': enc(cache.hier_get(list(q), **kw))})
    n_nodes = 0
    for t in cache.mem.values():
        sizes = [0]
        t._count_node(t.nodes, sizes)
        n_nodes += sizes[0]
    traces.append({'seed': 77, 'vocab': 1500, 'init': {'eos_ids': None, 'stop_words': [], 'max_node': 40,
                                                       'max_output_node': 10},
                   'ops': ops, 'final': {'n_trees': len(cache.mem), 'n_nodes': n_nodes}})
    for i, tr in enumerate(traces):
        with open(os.path.join(OUT, f'trie_trace_{i}.json'), 'w') as f:
            json.dump(tr, f, separators=(',', ':'))
    with open(os.path.join(OUT, 'trie_kats.json'), 'w') as f:
        json.dump({'reference_tests': reference_kats(ref), 't64b8': t64b8(ref)}, f, separators=(',', ':'))
    print('wrote', len(traces), 'traces +', 'kats to', OUT)


if __name__ == '__main__':
    main()
