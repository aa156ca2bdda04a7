# -*- coding: utf-8 -*-
"""-m gpu: device-side hier_get (csrc/la_trie_wg.hip: one workgroup per query, trees up to 256 rows; csrc/la_trie_dev.hip: one
wavefront per query, <= 64 rows) must be bit-identical to the reference's golden traces (recorded from lookahead_cache.py by
oracle/gen_golden.py) and to the host trie on larger forests."""
import os
import random

import numpy as np
import pytest
import torch

from painlessinferenceacceleration_amd.device_trie import DeviceTrie
from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
from tests import trie_replay as tr

pytestmark = pytest.mark.gpu


def _rows(mask_rows):
    m = np.asarray(mask_rows, dtype=np.uint64)
    if m.ndim == 1:
        return [int(x) for x in m]
    return [sum(int(x) << (64 * w) for w, x in enumerate(row)) for row in m]


@pytest.mark.parametrize('algo', ['wg', 'wave'])
@pytest.mark.parametrize('path', tr.trace_files(), ids=os.path.basename)
def test_device_hier_get_replays_reference_trace(path, algo):
    """every recorded hier_get of the reference traces (the first 60 per trace, and with the workgroup kernel EVERY query wider than 64
    rows on top: decoding_length 128 / 256 of the benchmark-size traces) answered by the device kernel"""
    trace = tr.load(path)
    init = trace['init']
    cache = LookaheadCache(eos_ids=init['eos_ids'], stop_words={w: 1 for w in init['stop_words']},
                           max_node=init['max_node'], max_output_node=init['max_output_node'])
    checked = 0
    for i, op in enumerate(trace['ops']):
        name = op['op']
        if name == 'put':
            cache.put(list(op['tokens']), branch_length=op['branch_length'], final=op['final'], mode=op['mode'], idx=op['idx'])
        elif name == 'stream_put':
            cache.stream_put(list(op['tokens']), branch_length=op['branch_length'], final=op['final'], idx=op['idx'])
        elif name == 'hier_get':
            wide = op['decoding_length'] > 64
            if (wide and (algo == 'wave' or len(op['tokens']) > 8)) or (checked >= 60 and not wide):
                continue
            dev = DeviceTrie(cache, idx=op['idx'], algo=algo, max_rows=op['decoding_length'])
            got = dev.hier_get([list(op['tokens'])], decoding_length=op['decoding_length'], branch_length=op['branch_length'],
                               min_input_size=op['min_input_size'], min_output_size=op['min_output_size'], mode=op['mode'])[0]
            exp = op['out']
            ctx = f"op {i}: { {k: v for k, v in op.items() if k != 'out'} }"
            assert got[0] == exp['ids'], ctx
            assert _rows(got[1]) == exp['rows'][:len(exp['ids'])], ctx     # (empty ids come with the default [[1]] mask)
            assert got[2] == exp['sizes'], ctx
            checked += 1
        elif name == 'reset_input_freqs':
            cache.reset_input_freqs(op['idx'])
        elif name == 'squeeze_branch_counts':
            cache.squeeze_branch_counts()
        elif name == 'fresh':
            cache.fresh()
        elif name == 'limits':
            cache.max_node, cache.max_output_node = op['max_node'], op['max_output_node']
    assert checked >= 20


@pytest.mark.parametrize('limits', [(0, 0, 0), (0, 0, -1), (16, 8, -1), (64, 100, 20)], ids=['default', 'chains-256-threads', 'scratch-levels-and-candidates', 'mixed'])
def test_workgroup_hier_get_wide_trees_match_host_on_large_forest_with_dead_nodes(limits):
    """limits = la_trie_query.lds_level_cap / lds_cand_cap / one_wave_cap: the library's (LDS level buffers, candidates ordered by the chain
    form, by one wave up to 256 of them); the chain form by all 256 threads; levels above 16 entries and candidate sets above 8 through the
    global scratch (the LEVEL form of the ordering: tests/trie_wg_model.py); a mix.
    The forest of tests/test_trie_wg_model.py (100 x 256-token warm-up, a prompt whose input frequencies were reset = dead nodes, a
    live prompt): 120 queries x 4 mode settings x budgets {16, 64, 128, 256} in batched launches against the host trie, multi-word
    row masks included; the 1-token queries walk subtrees of thousands of entries (global-scratch levels / candidate sets)."""
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
    cache.reset_input_freqs(0)
    prompt = sum((phrases[rng.randrange(200)] for _ in range(60)), [])
    cache.put(prompt, branch_length=13, mode='input', idx=0)
    queries = []
    for _ in range(120):
        ph = phrases[rng.randrange(200)]
        k = rng.randrange(1, len(ph))
        queries.append(ph[max(0, k - 2):k] if rng.random() < 0.6 else ph[k - 1:k] if rng.random() < 0.5 else [rng.randrange(3, 2000), rng.randrange(3, 2000)])
    dev = DeviceTrie(cache, idx=0, max_rows=256)
    dev.wg_limits = limits
    n_wide = 0
    for mode, mi, mo in [('mix', 0, 32), ('mix', 2, 8), ('output', 0, 16), ('input', 1, 0)]:
        for dl, bl in [(16, 6), (64, 12), (128, 32), (256, 20)]:
            mo_ = mo if dl <= 64 else dl // 2
            got = dev.hier_get(queries, decoding_length=dl, branch_length=bl, min_input_size=mi, min_output_size=mo_, mode=mode)
            for qy, g in zip(queries, got):
                ids, rowmask, parent, sizes = cache.hier_get_packed(qy, decoding_length=dl, branch_length=bl, min_input_size=mi,
                                                                   min_output_size=mo_, mode=mode, idx=0)
                assert g[0] == ids.tolist(), (mode, dl, qy)
                assert _rows(g[1])[:len(g[0])] == _rows(rowmask), (mode, dl, qy)
                assert g[2] == sizes, (mode, dl, qy)
                n_wide += len(g[0]) > 64
    assert n_wide > 20


@pytest.mark.parametrize('seed', range(4))
def test_workgroup_hier_get_matches_the_oracle_fuzz(seed):
    """Differential fuzz against the ORACLE (oracle/trie_oracle.py, the pinned restatement of lookahead_cache.py), not against the host trie:
    the op stream of tests/test_native_trie.py::test_native_matches_oracle_fuzz (puts in both modes, stream_puts with final flushes = squeezes,
    fresh, tiny node limits, eos / stop words, vocabularies of 5 .. 2000 tokens) goes through the native trie + its incremental device mirror
    and through the oracle; every hier_get — all three modes, budgets 2 .. 200, branch lengths 0 / 3 / 12 / 30, 0 .. 3-token queries, both input
    slots — is answered by the workgroup kernel and must equal the oracle's ids, mask and sizes."""
    from oracle.trie_oracle import TrieOracle
    rng = random.Random(2000 + seed)
    vocab = rng.choice([5, 20, 200, 2000])
    kw = dict(eos_ids=rng.choice([None, (2,), (2, 3)]), stop_words={4: 1} if seed % 2 else {},
              max_node=rng.choice([65536, 60]), max_output_node=rng.choice([512, 8]))
    a, b = LookaheadCache(**kw), TrieOracle(**kw)
    dev = DeviceTrie(a, idxs=[0, 1], max_rows=256)
    hist = [rng.randrange(vocab) for _ in range(30)]
    n_q = n_wide = 0
    for step in range(1200):
        r = rng.random()
        if r < 0.25:
            toks = [rng.choice(hist) if rng.random() < 0.7 else rng.randrange(vocab) for _ in range(rng.randint(0, 40))]
            args = dict(branch_length=rng.choice([3, 8, 13, 31]), final=rng.random() < 0.1, mode=rng.choice(['input', 'output']),
                        idx=rng.choice([0, 1, 2]))
            a.put(toks, **args); b.put(toks, **args)
            hist = (hist + toks)[-60:]
        elif r < 0.5:
            toks = [rng.choice(hist) if rng.random() < 0.7 else rng.randrange(vocab) for _ in range(rng.randint(0, 13))]
            args = dict(branch_length=rng.choice([3, 8, 13]), final=rng.random() < 0.08, idx=rng.choice([0, 1]))
            a.stream_put(toks, **args); b.stream_put(toks, **args)
            hist = (hist + toks)[-60:]
        elif r < 0.98:
            p = rng.randrange(1, len(hist))
            q = hist[max(0, p - rng.randint(0, 3)):p]
            dl = rng.choice([2, 7, 16, 64, 100, 200])
            args = dict(decoding_length=dl, branch_length=rng.choice([0, 3, 12, 30]), min_input_size=rng.choice([0, 0, 1]),
                        min_output_size=rng.choice([0, dl // 2, 1]), mode=rng.choice(['mix', 'mix', 'input', 'output']))
            idx = rng.choice([0, 1])
            ref = b.hier_get(q, idx=idx, **args)
            got = dev.hier_get([q], idxs=[idx], **args)[0]
            n = len(ref[0])
            assert got[0] == [int(x) for x in ref[0]], (step, q, args)
            assert _rows(got[1])[:n] == (tr.rows_of(ref[1])[:n] if n else []), (step, q, args)
            assert got[2] == [int(x) for x in ref[2]], (step, q, args)
            n_q += 1
            n_wide += n > 64
        else:
            a.fresh(); b.fresh()
    assert n_q > 400 and dev.stats['patches'] > 100, (n_q, dev.stats)
    print('fuzz', seed, 'queries', n_q, 'wider than 64 rows', n_wide, dev.stats)


@pytest.mark.parametrize('algo', ['wg', 'wave'])
def test_device_hier_get_batched_matches_host_on_large_forest(algo):
    """100 x 256-token warm-up (the reference benchmark's recipe) + an input-mode prompt; 256 queries in one launch,
    all three modes, oversize subtrees (thresholds) included."""
    rng = random.Random(0)
    nr = np.random.RandomState(0)
    cache = LookaheadCache(eos_ids=[None])
    phrases = [nr.randint(3, 3000, size=nr.randint(3, 10)).tolist() for _ in range(300)]
    for _ in range(100):
        seq = []
        while len(seq) < 256:
            seq.extend(phrases[min(int(nr.zipf(1.3)) - 1, 299)])
        cache.put(seq[:256], branch_length=13, mode='output', idx=-1)
    prompt = sum((phrases[rng.randrange(300)] for _ in range(60)), [])
    cache.put(prompt, branch_length=13, mode='input', idx=0)
    queries = []
    for _ in range(256):
        ph = phrases[rng.randrange(300)]
        k = rng.randrange(1, len(ph))
        queries.append(ph[max(0, k - 2):k] if rng.random() < 0.8 else [rng.randrange(3, 3000), rng.randrange(3, 3000)])
    dev = DeviceTrie(cache, idx=0, algo=algo)
    big = 0
    for mode, mi, mo in [('mix', 0, 32), ('mix', 2, 8), ('output', 0, 16), ('input', 1, 0)]:
        got = dev.hier_get(queries, decoding_length=64, branch_length=12, min_input_size=mi, min_output_size=mo, mode=mode)
        for qy, g in zip(queries, got):
            ids, rowmask, parent, sizes = cache.hier_get_packed(qy, decoding_length=64, branch_length=12, min_input_size=mi,
                                                               min_output_size=mo, mode=mode, idx=0)
            assert g[0] == ids.tolist(), (mode, qy)
            assert _rows(g[1]) == _rows(rowmask), (mode, qy)
            assert g[2] == sizes, (mode, qy)
            big += len(g[0]) == 64
    assert big > 20        # the budget cap (and with it the cut-off rule) was exercised


@pytest.mark.parametrize('path', tr.trace_files(), ids=os.path.basename)
def test_incremental_device_mirror_replays_reference_trace_with_interleaved_updates(path):
    """ONE DeviceTrie for the whole trace: every put / stream_put / reset between the recorded hier_get calls reaches the device
    as a patch (la_cache_mirror_patch + k_trie_patch), squeeze / fresh as a new image; every recorded hier_get (all idx slots of
    the trace mirrored as fi planes) is answered by the device kernel and must equal the reference's output."""
    trace = tr.load(path)
    init = trace['init']
    cache = LookaheadCache(eos_ids=init['eos_ids'], stop_words={w: 1 for w in init['stop_words']},
                           max_node=init['max_node'], max_output_node=init['max_output_node'])
    idxs = sorted({op['idx'] for op in trace['ops'] if op['op'] == 'hier_get' and op['idx'] >= 0}) or [0]
    dev = DeviceTrie(cache, idxs=idxs)
    checked = 0
    for i, op in enumerate(trace['ops']):
        name = op['op']
        if name == 'put':
            cache.put(list(op['tokens']), branch_length=op['branch_length'], final=op['final'], mode=op['mode'], idx=op['idx'])
        elif name == 'stream_put':
            cache.stream_put(list(op['tokens']), branch_length=op['branch_length'], final=op['final'], idx=op['idx'])
        elif name == 'hier_get':
            if op['decoding_length'] > 64 or op['idx'] < 0 or len(op['tokens']) > 8:
                continue
            got = dev.hier_get([list(op['tokens'])], idxs=[op['idx']], decoding_length=op['decoding_length'],
                               branch_length=op['branch_length'], min_input_size=op['min_input_size'],
                               min_output_size=op['min_output_size'], mode=op['mode'])[0]
            exp = op['out']
            ctx = f"op {i}: { {k: v for k, v in op.items() if k != 'out'} }"
            assert got[0] == exp['ids'], ctx
            assert _rows(got[1]) == exp['rows'][:len(exp['ids'])], ctx
            assert got[2] == exp['sizes'], ctx
            checked += 1
        elif name == 'reset_input_freqs':
            cache.reset_input_freqs(op['idx'])
        elif name == 'squeeze_branch_counts':
            cache.squeeze_branch_counts()
        elif name == 'fresh':
            cache.fresh()
        elif name == 'limits':
            cache.max_node, cache.max_output_node = op['max_node'], op['max_output_node']
    assert checked >= 20 and dev.stats['patches'] >= 5, dev.stats
    print(os.path.basename(path), 'queries', checked, dev.stats)


@pytest.mark.parametrize('dmode', ['hier', 'one'])
def test_batch_loop_with_device_trie_equals_host_trie_loop(dmode):
    """pretrained_model_batch.lookahead_generation with decoding_kwargs['device_trie']: the drafts of all samples come from one
    device launch per step over the incremental mirror; sequences, dls and edls must equal the host-trie loop's, request after
    request (the second request runs on the trie the first one grew) — with the per-step trie update shipped from the host as a
    patch, and with the update done by the device itself (device_trie_update: la_trie_stream_put_dev + host replay)."""
    import torch
    from painlessinferenceacceleration_amd.modeling_llama_batch import LlamaForCausalLM as BatchLlama
    from tests.tiny_model import noisy_copies, tiny_decisive_weights, tiny_shape
    shape = tiny_shape()
    sd = tiny_decisive_weights(0, torch.bfloat16)
    rs = np.random.RandomState(2)
    B, P = 4, 24
    prompts = rs.randint(3, shape.vocab, size=(B, P))
    outs = []
    # (the last configuration runs the 4 samples as two passes of 2 blocks: two device updates per step, replayed in batch order)
    for use_dev, dev_update, mblocks in ((False, False, None), (True, False, None), (True, True, None), (True, True, 2)):
        model = BatchLlama(shape, dict(sd), max_length=256, max_batch=B, eos_token_id=2, max_blocks=mblocks)
        truth = model.greedy_search(torch.from_numpy(prompts), P + 100, eos_token_id=None)[:, P:].tolist()
        model.lookahead_cache = LookaheadCache(eos_ids=[2])
        for b in range(B):
            for c in noisy_copies(prompts[b, -2:].tolist() + truth[b], 6, 0.3, shape.vocab, seed=50 + b):
                model.lookahead_cache.put(c, branch_length=13, mode='output', idx=-1)
        runs = []
        for req in range(2):
            dk = {'use_lookahead': True, 'decoding_mode': dmode, 'decoding_length': 64, 'branch_length': 12,
                  'per_sample_budget': True, 'device_trie': use_dev, 'device_trie_update': dev_update}
            out = model.lookahead_generation(torch.from_numpy(prompts), stopping_criteria=P + 90, eos_token_id=2, pad_token_id=0,
                                             return_dict_in_generate=True, decoding_kwargs=dk)
            runs.append((out.sequences.tolist(), out.kwargs['dls'], out.kwargs['edls']))
            assert [s[P:P + 60] for s in out.sequences.tolist()] == [t[:60] for t in truth]       # lossless
        if use_dev and not dev_update:
            assert model._dev_trie.stats['patches'] > 10
        if dev_update:
            # the per-step updates ran on the device (la_trie_stream_put_dev behind every verify pass, replayed on the host): what is
            # left as patches are the host-side updates (prompt puts, first tokens, final flushes)
            dt = model._dev_trie
            assert dt.stats_put['calls'] > 10 and dt.stats_put['replays'] > 10 and dt.stats['patches'] < 10, (dt.stats, dt.stats_put)
            torch.cuda.synchronize()
            assert int(dt.meta.cpu()[1]) == 0 and int(dt.meta.cpu()[3]) > 100
        outs.append(runs)
    assert outs[0] == outs[1] == outs[2] == outs[3]
    if dmode == 'hier':
        assert max(outs[0][0][1]) > 16      # per-sample budget: trees larger than the reference's (64 // 4) // 4 rows
    else:
        assert 4 < max(outs[0][0][1]) <= 13  # 'one': a single chain of at most branch_length + 1 rows per sample (la_trie_one_get_dev2)


@pytest.mark.parametrize('dmode', ['hier', 'one', 'par'])
def test_single_sequence_loop_with_device_trie_equals_host_trie_loop(dmode):
    """pretrained_model.lookahead_generation (bs = 1) with decoding_kwargs['device_trie']: every draft comes from the wavefront
    trie walk over the incremental device mirror; tokens, dls and edls must equal the host-trie loop's (interpreter and native
    loop), request after request."""
    import torch
    from painlessinferenceacceleration_amd.modeling_llama import LlamaForCausalLM
    from tests.tiny_model import noisy_copies, tiny_decisive_weights, tiny_shape
    shape = tiny_shape()
    sd = tiny_decisive_weights(0, torch.bfloat16)
    rs = np.random.RandomState(4)
    P = 40
    prompt = rs.randint(3, shape.vocab, size=(1, P))
    outs = []
    for use_dev, native in ((False, True), (False, False), (True, False)):
        model = LlamaForCausalLM(shape, dict(sd), max_length=512, eos_token_id=2)
        truth = model.greedy_search(torch.from_numpy(prompt), P + 120, eos_token_id=None)[0, P:].tolist()
        model.lookahead_cache = LookaheadCache(eos_ids=[2])
        for c in noisy_copies(prompt[0, -2:].tolist() + truth, 8, 0.3, shape.vocab, seed=60):
            model.lookahead_cache.put(c, branch_length=13, mode='output', idx=-1)
        runs = []
        for req in range(2):
            dk = {'use_lookahead': True, 'decoding_mode': dmode, 'decoding_length': 64, 'branch_length': 12,
                  'device_trie': use_dev, 'native_loop': native}
            out = model.lookahead_generation(torch.from_numpy(prompt), stopping_criteria=P + 100, eos_token_id=2,
                                             return_dict_in_generate=True, decoding_kwargs=dk)
            runs.append((out.sequences.tolist(), out.kwargs['dls'], out.kwargs['edls']))
            assert out.sequences[0, P:P + 80].tolist() == truth[:80]                                # lossless
        if use_dev:
            assert model._dev_trie.stats['patches'] > 5
        outs.append(runs)
    assert outs[0] == outs[1] == outs[2]
    assert np.mean(outs[2][1][2][1:]) > 2.0


def test_second_mirror_revokes_the_first_and_staging_is_reusable_back_to_back():
    """la_cache_mirror_enable replaces a cache's mirror: an older DeviceTrie would apply its next patch to the wrong image, so it
    must refuse to be used; and hier_get_dev(sync=True) called back to back (no stream sync in between) must not overwrite a
    pinned staging buffer an earlier H2D copy is still reading (an event behind the copies is waited for)."""
    cache = LookaheadCache(eos_ids=[None])
    rs = np.random.RandomState(3)
    seqs = [rs.randint(3, 200, size=60).tolist() for _ in range(20)]
    for s_ in seqs:
        cache.put(s_, branch_length=9, mode='output', idx=-1)
    a = DeviceTrie(cache, idx=0)
    q1, q2 = [seqs[0][5:7], seqs[1][9:11]], [seqs[2][3:5], seqs[3][1:3]]
    want1 = []
    for q in q1:                                      # (the packed getters return views of buffers the next call reuses)
        w = cache.hier_get_packed(q, 64, 8, 0, 32, 'mix', 0)
        want1.append((w[0].tolist(), w[1].copy()))
    # back to back: second call rewrites the staging buffers right after the first one queued its copies
    a.hier_get_dev(q1, decoding_length=64, branch_length=8, min_input_size=0, min_output_size=32, mode='mix')
    ids1 = a.out_ids[:2 * 64].clone(); n1 = a.out_n[:2].clone(); rm1 = a.out_rm[:2 * 64].clone()
    a.hier_get_dev(q2, decoding_length=64, branch_length=8, min_input_size=0, min_output_size=32, mode='mix')
    torch.cuda.synchronize()
    for b, (ids, rm) in enumerate(want1):
        n = int(n1[b])
        assert ids1[b * 64:b * 64 + n].cpu().tolist() == ids and rm1[b * 64:b * 64 + n].cpu().numpy().view(np.uint64).tolist() == rm.tolist()
    b_ = DeviceTrie(cache, idx=0)
    with pytest.raises(RuntimeError):
        a.hier_get([seqs[0][5:7]], decoding_length=64, branch_length=8)
    assert b_.hier_get([seqs[0][5:7]], decoding_length=64, branch_length=8, min_output_size=32)[0][0] == want1[0][0]


def _host_image(cache, n_planes):
    """the host mirror's arrays (la_cache_mirror_image + la_cache_mirror_ccap) as numpy"""
    import ctypes as C
    from painlessinferenceacceleration_amd import _lib
    lib = _lib.lib
    n, full, ni, nd = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.la_cache_mirror_state(cache._h, C.byref(n), C.byref(full), C.byref(ni), C.byref(nd)) == 0
    k = n.value
    tok, cs, cc, cap = (np.zeros(k, dtype=np.int32) for _ in range(4))
    fo, fi = np.zeros(k, dtype=np.float64), np.zeros(max(n_planes, 1) * k, dtype=np.float64)
    pd = C.POINTER(C.c_double)
    assert lib.la_cache_mirror_image(cache._h, k, tok.ctypes.data_as(_lib.pi32), fo.ctypes.data_as(pd), fi.ctypes.data_as(pd),
                                     cs.ctypes.data_as(_lib.pi32), cc.ctypes.data_as(_lib.pi32)) == 0
    assert lib.la_cache_mirror_ccap(cache._h, k, cap.ctypes.data_as(_lib.pi32)) == 0
    return k, tok, cs, cc, cap, fo, fi.reshape(max(n_planes, 1), k)


def _assert_images_equal(dt, cache):
    torch.cuda.synchronize()
    k, tok, cs, cc, cap, fo, fi = _host_image(cache, len(dt.idxs))
    meta = dt.meta.cpu().numpy()
    assert meta[1] == 0, 'device arena overflowed'
    assert meta[0] == k, (meta, k)
    # compared: every LIVE record (reachable from the super-root) — ids, tokens, child blocks, block capacities, frequencies.
    # Dead words are never read again and may differ: the unused tail of a block the HOST grew (its patch carries only the records
    # it wrote), and the frequency of a copy left behind by a moved block (the device's parallel walk may have counted a branch on
    # the old copy just before the move carried the count to the new one; the strictly serial host counts on the new copy only).
    live = np.zeros(k, dtype=bool)
    stack = [0]
    live[0] = True
    while stack:
        u = stack.pop()
        for c_ in range(int(cs[u]), int(cs[u]) + int(cc[u])):
            live[c_] = True
            stack.append(c_)
    for name, want in (('tok', tok), ('cstart', cs), ('ccount', cc), ('ccap', cap)):
        got = getattr(dt, name)[:k].cpu().numpy()
        assert got[live].tolist() == want[live].tolist(), name
    assert live.sum() > 10
    assert dt.fo[:k].cpu().numpy()[live].tolist() == fo[live].tolist()
    for p in range(len(dt.idxs)):
        assert dt.fi[p * dt.cap:p * dt.cap + k].cpu().numpy()[live].tolist() == fi[p][live].tolist()
    # the token -> root table covers every tree
    r0, rc = int(cs[0]), int(cc[0])
    ro = dt.root_of.cpu().numpy()
    for j in range(rc):
        if 0 <= tok[r0 + j] < dt.put_vocab:
            assert ro[tok[r0 + j]] == r0 + j
    assert (ro >= 0).sum() == sum(1 for j in range(rc) if 0 <= tok[r0 + j] < dt.put_vocab)


@pytest.mark.parametrize('B,bl,eos,stop', [(1, 13, [None], []), (4, 9, [2], [7]), (8, 13, [2, 5], [])])
def test_device_stream_put_keeps_the_device_image_identical_to_the_host_mirror(B, bl, eos, stop):
    """la_trie_stream_put_dev (LookaheadCache.stream_put on the device, lookahead_cache.py:369-406): B sequences emit 1..40 tokens
    per step (repeating motifs, so branches exist partly / fully / not at all; -1 fillers, eos ids and stop-word roots included);
    the device inserts them from HBM, the host replays them, and after every few steps the device arrays must equal the host
    mirror's word for word (records, block capacities, frequencies, record count) — with host-side updates (an input-mode put that
    arrives as a patch, a reset of a slot's input frequencies) interleaved — and device queries must equal host queries."""
    rs = np.random.RandomState(100 + B)
    V = 300
    cache = LookaheadCache(eos_ids=eos, stop_words=set(stop))
    motifs = [rs.randint(8, V, size=rs.randint(5, 30)).tolist() for _ in range(12)]
    for m in motifs[:6]:
        cache.put(m, branch_length=bl, mode='output', idx=-1)
    for b in range(B):
        cache.put(rs.randint(8, V, size=40).tolist(), branch_length=bl, mode='input', idx=b)
        cache.stream_put(rs.randint(8, V, size=rs.randint(1, 4)).tolist(), branch_length=bl, final=False, mode='output', idx=b)
    dt = DeviceTrie(cache, idxs=list(range(B)), put_vocab=V, cap_slack=200000)
    dt.load_stream_buffers()
    src = torch.zeros(B * 40, dtype=torch.int32, device='cuda')
    cnt = torch.zeros(B, dtype=torch.int32, device='cuda')
    for step in range(40):
        active = [b for b in range(B) if rs.rand() < 0.85] or [0]
        puts = []
        hs, hc = np.zeros(B * 40, dtype=np.int32), np.zeros(B, dtype=np.int32)
        for k, b in enumerate(active):
            n = int(rs.randint(1, 41)) if rs.rand() < 0.3 else int(rs.randint(1, 9))
            toks = []
            while len(toks) < n:
                toks.extend(motifs[rs.randint(len(motifs))][rs.randint(0, 4):] if rs.rand() < 0.7 else rs.randint(8, V, size=3).tolist())
            toks = toks[:n]
            if rs.rand() < 0.15:
                toks[rs.randint(n)] = -1
            if eos[0] is not None and rs.rand() < 0.1:
                toks[rs.randint(n)] = eos[rs.randint(len(eos))]
            if stop and rs.rand() < 0.3:
                toks[rs.randint(n)] = stop[0]
            hs[k * 40:k * 40 + n] = toks
            hc[k] = n
            puts.append((b, toks))
        src.copy_(torch.from_numpy(hs)); cnt.copy_(torch.from_numpy(hc))
        dt.stream_put_dev(src.data_ptr(), 40, cnt.data_ptr(), active, bl)
        assert dt.replay(puts, bl)
        if step % 7 == 3:                               # a host-side update between device updates: reaches the device as a patch
            cache.put(rs.randint(8, V, size=25).tolist(), branch_length=bl, mode='input', idx=int(rs.randint(B)))
            assert dt.sync() == 'patch'
        if step % 11 == 5:
            cache.reset_input_freqs(int(rs.randint(B)))
            dt.sync()
        if step % 5 == 4 or step == 39:
            _assert_images_equal(dt, cache)
            # hold-back buffers
            ol = dt.olen.cpu().numpy(); ob = dt.obuf.cpu().numpy().reshape(B, -1)
            import ctypes as C
            from painlessinferenceacceleration_amd import _lib
            for b in range(B):
                buf = np.zeros(128, dtype=np.int32); n = C.c_int32()
                assert _lib.lib.la_cache_stream_buffer(cache._h, b, 128, buf.ctypes.data_as(_lib.pi32), C.byref(n)) == 0
                assert ol[b] == n.value and ob[b, :n.value].tolist() == buf[:n.value].tolist()
            qs = [motifs[rs.randint(len(motifs))][1:3] for _ in range(B)]
            got = dt.hier_get(qs, decoding_length=64, branch_length=bl - 1, min_output_size=32, mode='mix', idxs=list(range(B)))
            for b, q in enumerate(qs):
                w = cache.hier_get_packed(q, 64, bl - 1, 0, 32, 'mix', b)
                assert got[b][0] == w[0].tolist() and got[b][1].tolist() == w[1].tolist()
    assert dt.stats['full_uploads'] == 1, dt.stats        # everything after the first image went as device updates / patches
    st = dt.meta.cpu().numpy()
    assert st[2] > 100 and st[3] > 500, st                # branches inserted / records appended by the device


def test_device_stream_put_overflow_falls_back_to_a_full_image():
    """An insert that would pass the image capacity stops the device inserts (sticky flag); the host's replay passes the same
    capacity, reports it, and the next sync uploads a larger image after which device updates continue."""
    rs = np.random.RandomState(7)
    V, bl = 500, 13
    cache = LookaheadCache(eos_ids=[None])
    cache.put(rs.randint(8, V, size=50).tolist(), branch_length=bl, mode='output', idx=-1)
    dt = DeviceTrie(cache, idxs=[0, 1], put_vocab=V, cap_slack=300)
    dt.load_stream_buffers()
    cap0 = dt.cap
    src = torch.zeros(2 * 40, dtype=torch.int32, device='cuda')
    cnt = torch.zeros(2, dtype=torch.int32, device='cuda')
    overflowed = False
    for step in range(30):
        toks = [rs.randint(8, V, size=20).tolist() for _ in range(2)]
        hs = np.zeros(80, dtype=np.int32)
        hs[:20], hs[40:60] = toks[0], toks[1]
        src.copy_(torch.from_numpy(hs)); cnt.fill_(20)
        dt.stream_put_dev(src.data_ptr(), 40, cnt.data_ptr(), [0, 1], bl)
        ok = dt.replay([(0, toks[0]), (1, toks[1])], bl)
        if not ok:
            overflowed = True
            torch.cuda.synchronize()
            assert int(dt.meta.cpu()[1]) == 1
            assert dt.sync() == 'full' and dt.cap > cap0
        if step % 6 == 5:
            dt.sync()
            _assert_images_equal(dt, cache)
    assert overflowed and dt.stats['full_uploads'] >= 2
    _assert_images_equal(dt, cache)


@pytest.mark.parametrize('mode', ['mix', 'output', 'input'])
def test_device_one_get_equals_host_one_get(mode):
    """la_trie_one_get_dev2 (LookaheadCache.one_get on the device mirror: greedy most-frequent chain, lower-triangular masks) vs the
    host trie (bit-exact on the reference's traces, tests/test_native_trie.py) on a forest with ties, dead branches, stop words,
    prompts in two input slots, queries that match fully / partly / not at all, and every branch length."""
    rs = np.random.RandomState(31)
    V = 60
    cache = LookaheadCache(eos_ids=[None], stop_words={11: 1})
    seqs = [rs.randint(3, V, size=rs.randint(6, 40)).tolist() for _ in range(120)]
    for s_ in seqs[:80]:
        cache.put(s_, branch_length=9, mode='output', idx=-1)
    for s_ in seqs[80:100]:
        cache.put(s_, branch_length=9, mode='input', idx=0)
    for s_ in seqs[100:]:
        cache.put(s_, branch_length=9, mode='input', idx=1)
    dt = DeviceTrie(cache, idxs=[0, 1])
    checked = chains = 0
    for bl in (1, 2, 5, 8, 13):
        queries, idxs = [], []
        for _ in range(48):
            src = seqs[rs.randint(len(seqs))]
            p = rs.randint(1, len(src))
            q = src[max(0, p - rs.randint(1, 4)):p]
            if rs.rand() < 0.2:
                q = q[:-1] + [int(rs.randint(3, V))]
            if rs.rand() < 0.1:
                q = q + [11]
            queries.append(q)
            idxs.append(int(rs.randint(2)))
        got = dt.one_get(queries, decoding_length=64, branch_length=bl, mode=mode, idxs=idxs)
        for q, ix, g in zip(queries, idxs, got):
            ids, mask, sizes = cache.one_get(q, decoding_length=64, branch_length=bl, mode=mode, idx=ix)
            assert g[0] == [int(x) for x in ids], (q, ix, bl)
            assert g[2] == [int(x) for x in sizes], (q, ix, bl)
            n = len(ids)
            assert g[1].tolist() == [(2 << i) - 1 for i in range(n)]
            checked += 1
            chains += n > 2
    assert checked == 240 and chains > 40
    # the degenerate settings (:491-492)
    got = dt.one_get([seqs[0][:2]], decoding_length=1, branch_length=8, mode=mode, idxs=[0])
    assert got[0][0] == [seqs[0][1]] and got[0][2] == []
