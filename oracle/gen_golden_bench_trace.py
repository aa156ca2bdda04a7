# -*- coding: utf-8 -*-
"""ORACLE tooling: ONE reference trace at the BENCHMARK's forest size (build container only).

Runs the reference LookaheadCache (imported in place from /root/reference, nothing copied) through the life of a
benchmark process — `Benchmark.warm_up` (benchmarks/benchmark.py:159-169: 100 answers x 256 tokens, branch_length + 1 = 13,
mode='output', idx=-1), then two requests shaped like `lookahead_generation` (pretrained_model.py:1117-1260: prompt put in
input mode, per-step hier_get + stream_put, final flush) — and records every hier_get's output.  The point of the trace
(SURVEY H1d): after the warm-up >= 1024 trees are dirty and more than 100 k nodes exist, so the first `final=True`
flush runs `squeeze_branch_counts` (lookahead_cache.py:572-576) over the whole forest and `Tree.squeeze/_squeeze/_count_node`
(:295-318) halve / prune every tree above `max_output_node`; the queries after it see the pruned forest.

    python oracle/gen_golden_bench_trace.py        # rewrites tests/golden/trie_trace_6.json
"""
import json
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import OUT, enc, load_reference  # noqa: E402


def main():
    ref = load_reference()
    rng = random.Random(2024)
    nrng = np.random.RandomState(2024)
    vocab = 32000
    # phrase bank with a Zipf pick (repeated n-grams => popular tokens own trees far above max_output_node = 512) mixed with
    # uniformly random tokens (=> thousands of distinct tree roots)
    phrases = [[rng.randrange(3, vocab) for _ in range(rng.randint(3, 12))] for _ in range(600)]
    hot = [rng.randrange(3, vocab) for _ in range(12)]          # a few very frequent tokens (punctuation-like)

    def text(n):
        out = []
        while len(out) < n:
            r = rng.random()
            if r < 0.55:
                out.extend(phrases[min(int(nrng.zipf(1.25)) - 1, len(phrases) - 1)])
            elif r < 0.75:
                out.append(hot[rng.randrange(len(hot))])
            else:
                out.append(rng.randrange(3, vocab))
        return out[:n]

    cache = ref.LookaheadCache(eos_ids=[2])                      # reference defaults: max_node 65536, max_output_node 512
    ops = []

    def n_nodes():
        tot = 0
        for t in cache.mem.values():
            sizes = [0]
            t._count_node(t.nodes, sizes)
            tot += sizes[0]
        return tot

    def get(q, **over):
        kw = dict(decoding_length=64, branch_length=12, min_input_size=0, min_output_size=32, mode='mix', idx=0)
        kw.update(over)
        ops.append({'op': 'hier_get', 'tokens': list(q), **kw, 'out': enc(cache.hier_get(list(q), **kw))})

    answers = []
    for _ in range(100):
        a = text(256)
        answers.append(a)
        cache.put(list(a), branch_length=13, mode='output', idx=-1)
        ops.append({'op': 'put', 'tokens': a, 'branch_length': 13, 'final': False, 'mode': 'output', 'idx': -1})
    warm_nodes, warm_dirty = n_nodes(), len(cache._update_trees)
    assert warm_dirty >= 1024 and warm_nodes > 100000, (warm_dirty, warm_nodes)

    def request(n_steps, n_get_cap):
        """one lookahead_generation call: prompt -> input-mode put, then (query, accepted tokens) per step, final flush"""
        src = answers[rng.randrange(len(answers))]
        prompt = text(96) + src[:32]
        cache.put(list(prompt), branch_length=13, mode='input', idx=0)
        ops.append({'op': 'put', 'tokens': prompt, 'branch_length': 13, 'final': False, 'mode': 'input', 'idx': 0})
        pos, seq, gets = 32, list(prompt), 0
        for step in range(n_steps):
            if gets < n_get_cap:
                get(seq[-2:])
                if step % 5 == 0:
                    get(seq[-1:], min_output_size=8, mode='output')
                    gets += 1
                if step % 7 == 0:
                    get([hot[step % len(hot)]], decoding_length=32, min_input_size=1, min_output_size=16)
                    gets += 1
                gets += 1
            n_acc = rng.randint(1, 9)
            toks = src[pos:pos + n_acc] if rng.random() < 0.8 else text(n_acc)
            pos += n_acc
            seq.extend(toks)
            cache.stream_put(list(toks), branch_length=12, final=False, mode='output', idx=0)
            ops.append({'op': 'stream_put', 'tokens': toks, 'branch_length': 12, 'final': False, 'idx': 0})
        cache.stream_put([], branch_length=12, final=True, mode='output', idx=0)
        ops.append({'op': 'stream_put', 'tokens': [], 'branch_length': 12, 'final': True, 'idx': 0})
        return seq

    request(20, 24)                       # the first flush squeezes the warmed forest
    after_nodes = n_nodes()
    assert after_nodes < warm_nodes and len(cache._update_trees) == 0, (warm_nodes, after_nodes)
    for _ in range(6):                    # popular roots right after the squeeze (halved frequencies, pruned singles)
        ph = phrases[min(int(nrng.zipf(1.25)) - 1, len(phrases) - 1)]
        k = rng.randrange(1, len(ph))
        get(ph[max(0, k - 2):k])
    for h in hot[:4]:
        get([h], min_output_size=16)
    request(16, 30)                       # a second request on the pruned forest (< 1024 dirty trees: no squeeze)
    trace = {'seed': 2024, 'vocab': vocab,
             'init': {'eos_ids': [2], 'stop_words': [], 'max_node': 65536, 'max_output_node': 512},
             'ops': ops,
             'final': {'n_trees': len(cache.mem), 'n_nodes': n_nodes()},
             'note': {'nodes_after_warmup': warm_nodes, 'dirty_trees_after_warmup': warm_dirty, 'nodes_after_first_flush': after_nodes}}
    path = os.path.join(OUT, 'trie_trace_6.json')
    with open(path, 'w') as f:
        json.dump(trace, f, separators=(',', ':'))
    n_get = sum(1 for o in ops if o['op'] == 'hier_get')
    print(f'wrote {path}: {len(ops)} ops, {n_get} hier_get, nodes {warm_nodes} -> {after_nodes} at the first flush, '
          f'{warm_dirty} dirty trees, final {trace["final"]}')


if __name__ == '__main__':
    main()
