# -*- coding: utf-8 -*-
"""Native trie (csrc/la_trie.cpp through the C ABI) vs the reference's golden traces and vs the oracle."""
import json
import os
import random

import numpy as np
import pytest

from oracle.trie_oracle import TrieOracle
from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
from tests import trie_replay as tr


@pytest.mark.parametrize('path', tr.trace_files(), ids=os.path.basename)
def test_native_replays_reference_trace(path):
    trace = tr.load(path)
    init = trace['init']
    cache = LookaheadCache(eos_ids=init['eos_ids'], stop_words={w: 1 for w in init['stop_words']},
                           max_node=init['max_node'], max_output_node=init['max_output_node'])
    n = tr.replay(cache, trace)
    assert n > 50
    st = cache.stats()
    assert st['n_trees'] == trace['final']['n_trees']
    assert st['n_nodes'] == trace['final']['n_nodes']


def test_native_reference_unit_tests():
    kats = json.load(open(os.path.join(tr.GOLDEN, 'trie_kats.json')))
    for kat in kats['reference_tests']:
        cache = LookaheadCache(eos_ids=None)
        for p in kat['puts']:
            cache.put([9] + list(p), branch_length=8, mode='output', idx=-1)   # tree 9 <- p, like Tree(1).put(p)
        tr.check_get(cache.hier_get([9, 1], decoding_length=63, branch_length=3), kat['out'], kat['name'])


def test_native_t64b8_packed():
    kat = json.load(open(os.path.join(tr.GOLDEN, 'trie_kats.json')))['t64b8']
    cache = LookaheadCache(eos_ids=None)
    for p in kat['puts']:
        cache.put(list(p), branch_length=13, mode='output', idx=-1)
    ids, rowmask, parent, sizes = cache.hier_get_packed(kat['query'], decoding_length=64, branch_length=12,
                                                       min_output_size=32)
    assert ids.tolist() == kat['out']['ids'] and [int(r) for r in rowmask] == kat['out']['rows']
    assert sizes == [0, 63]
    # parent pointers agree with the mask: row = parent row | own bit
    for i in range(1, 64):
        assert int(rowmask[i]) == int(rowmask[parent[i]]) | (1 << i)
    leaves = [i for i in range(64) if i == 63 or bin(int(rowmask[i + 1])).count('1') <= bin(int(rowmask[i])).count('1')]
    assert len(leaves) == 8 and all(bin(int(rowmask[i])).count('1') == 13 for i in leaves)


@pytest.mark.parametrize('seed', range(6))
def test_native_matches_oracle_fuzz(seed):
    """Differential fuzz (host logic only): identical op streams through the native trie and the oracle."""
    rng = random.Random(1000 + seed)
    vocab = rng.choice([5, 20, 200, 2000])
    kw = dict(eos_ids=rng.choice([None, (2,), (2, 3)]), stop_words={4: 1} if seed % 2 else {},
              max_node=rng.choice([65536, 60]), max_output_node=rng.choice([512, 8]))
    a, b = LookaheadCache(**kw), TrieOracle(**kw)
    hist = [rng.randrange(vocab) for _ in range(30)]
    for step in range(1500):
        r = rng.random()
        if r < 0.25:
            toks = [rng.choice(hist) if rng.random() < 0.7 else rng.randrange(vocab) for _ in range(rng.randint(0, 40))]
            args = dict(branch_length=rng.choice([3, 8, 13]), final=rng.random() < 0.1, mode=rng.choice(['input', 'output']),
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
            dl = rng.choice([2, 7, 16, 64])
            args = dict(decoding_length=dl, branch_length=rng.choice([0, 3, 12]), min_input_size=rng.choice([0, 0, 1]),
                        min_output_size=rng.choice([0, dl // 2, 1]), mode=rng.choice(['mix', 'mix', 'input', 'output']),
                        idx=rng.choice([0, 1]))
            ra, rb = a.hier_get(q, **args), b.hier_get(q, **args)
            assert ra[0] == rb[0] and ra[2] == rb[2], (step, q, args)
            assert np.array_equal(ra[1], rb[1]), (step, q, args)
        else:
            a.fresh(); b.fresh()
    assert a.stats()['n_trees'] == b.n_trees() and a.stats()['n_nodes'] == b.n_nodes()


def test_squeeze_after_warmup_like_benchmark():
    """after a 100x256-token warm-up >= 1024 trees are dirty and final=True prunes the forest."""
    rng = np.random.RandomState(0)
    a, b = LookaheadCache(max_output_node=100), TrieOracle(max_output_node=100)
    for _ in range(100):
        toks = rng.randint(3, 1800, size=256).tolist()
        a.put(toks, branch_length=13, mode='output', idx=-1); b.put(toks, branch_length=13, mode='output', idx=-1)
    before = a.stats()
    assert before['n_dirty_trees'] >= 1024
    a.stream_put([], branch_length=13, final=True, idx=0); b.stream_put([], branch_length=13, final=True, idx=0)
    assert a.stats()['n_nodes'] == b.n_nodes()
    assert a.stats()['n_nodes'] < before['n_nodes'] and a.stats()['n_dirty_trees'] == 0


def test_save_load_roundtrip(tmp_path):
    rng = random.Random(5)
    a = LookaheadCache(eos_ids=None)
    for _ in range(30):
        a.put([rng.randrange(50) for _ in range(40)], branch_length=9, mode=rng.choice(['input', 'output']), idx=0)
    path = str(tmp_path / 'trie.bin')
    a.save_mem(path)
    b = LookaheadCache(eos_ids=None)
    b.load_mem(path)
    assert a.stats()['n_nodes'] == b.stats()['n_nodes'] and a.stats()['n_trees'] == b.stats()['n_trees']
    for _ in range(200):
        q = [rng.randrange(50) for _ in range(2)]
        ra, rb = a.hier_get(q, 64, 12, 0, 32), b.hier_get(q, 64, 12, 0, 32)
        assert ra[0] == rb[0] and np.array_equal(ra[1], rb[1]) and ra[2] == rb[2]


def test_argument_errors_match_reference_asserts():
    c = LookaheadCache()
    with pytest.raises(AssertionError):
        c.put([1, 2, 3], mode='mix')
    with pytest.raises(AssertionError):
        c.stream_put([1, 2, 3], idx=-1)
    with pytest.raises(AssertionError):
        c.hier_get([1], mode='bogus')
    with pytest.raises(AssertionError):
        c.bat_get([[1]], decoding_cursors=[1, 2], indices=[0])


def test_import_of_a_trie_saved_by_the_reference():
    """tests/golden/ref_mem_sample.json was written by the reference's save_mem (oracle/gen_golden_mem.py); after
    load_reference_mem the native trie answers the recorded queries exactly as the reference did after load_mem, and a
    native save/load round trip keeps them."""
    import json
    import tempfile
    g = json.load(open(os.path.join(tr.GOLDEN, 'ref_mem_queries.json')))
    cache = LookaheadCache(eos_ids=[2])
    cache.load_reference_mem(os.path.join(tr.GOLDEN, 'ref_mem_sample.json'))
    assert cache.stats()['n_trees'] == g['n_trees']

    def check(c):
        hits = 0
        for q in g['queries']:
            ids, mask, sizes = c.hier_get(q['q'], decoding_length=q['dl'], branch_length=8, min_input_size=0,
                                          min_output_size=max(q['dl'] // 2, 1), mode=q['mode'], idx=q['idx'])
            assert [int(x) for x in ids] == q['ids'] and tr.rows_of(mask) == q['rows'] and list(sizes) == q['sizes'], q
            hits += len(ids) > 1
        return hits
    assert check(cache) >= 10
    with tempfile.TemporaryDirectory() as d:
        cache.save_mem(os.path.join(d, 'snap.latrie'))
        again = LookaheadCache(eos_ids=[2])
        again.load_mem(os.path.join(d, 'snap.latrie'))
        check(again)
    with pytest.raises(Exception):                     # anything but the two record classes is refused
        import pickle
        bad = json.dumps(pickle.dumps(os.system).decode('latin-1'))
        with tempfile.NamedTemporaryFile('w', suffix='.json', delete=False) as f:
            json.dump(bad, f)
        LookaheadCache().load_reference_mem(f.name)


def test_bat_get_packed_equals_per_sample_retrieval_and_bat_get():
    """la_cache_bat_get_packed (one native call per batch) against hier_get_packed / one_get per sample with bat_get's budget
    rule, and against the padded bat_get canvas of the same call."""
    rs = np.random.RandomState(4)
    cache = LookaheadCache(eos_ids=[None])
    phrases = [rs.randint(3, 80, size=rs.randint(3, 9)).tolist() for _ in range(20)]

    def text(n):
        out = []
        while len(out) < n:
            out.extend(phrases[rs.randint(0, len(phrases))])
        return out[:n]
    for i in range(30):
        cache.put(text(50), branch_length=13, mode='output', idx=-1)
    for i in range(4):
        cache.put(text(40), branch_length=13, mode='input', idx=i)
    for trial in range(60):
        bs = int(rs.choice([1, 2, 3, 4, 8]))
        dl = int(rs.choice([64, 128, 256, 16]))
        if dl // bs > 64:
            continue
        mode = ['input', 'output', 'mix'][rs.randint(0, 3)]
        fmt = 'hier' if rs.rand() < 0.7 else 'one'
        qs = [text(30)[-2:] for _ in range(bs)]
        idxs = [int(rs.randint(0, 4)) for _ in range(bs)]
        got = cache.bat_get_packed(qs, decoding_length=dl, branch_length=12, mode=mode, indices=idxs, decoding_mode=fmt)
        per = dl // bs
        cursors = [int(rs.randint(5, 9)) for _ in range(bs)]
        id_list, canvas, size_list = cache.bat_get(qs, decoding_length=dl, branch_length=12, decoding_cursors=cursors, mode=mode,
                                                   indices=idxs, decoding_mode=fmt)
        lo = min(cursors)
        for b in range(bs):
            ids, rows, sizes = got[b]
            if fmt == 'hier':
                e_ids, e_rows, _, e_sizes = cache.hier_get_packed(qs[b], decoding_length=per, branch_length=12, min_input_size=0,
                                                                  min_output_size=max(per // 2, 1), mode=mode, idx=idxs[b])
                assert ids.tolist() == e_ids.tolist() and rows.tolist() == e_rows.tolist() and sizes == e_sizes
            else:
                e_ids, e_mask, e_sizes = cache.one_get(qs[b], decoding_length=per, branch_length=12, mode=mode, idx=idxs[b])
                assert ids.tolist() == list(e_ids) and sizes == list(e_sizes)
                assert rows.tolist() == [(2 << i) - 1 for i in range(len(ids))]
            n = len(ids)
            assert id_list[b][:n] == ids.tolist()
            off = cursors[b] - lo
            own = canvas[b][:n, off:off + n]
            assert tr.rows_of(own) == [int(r) for r in rows]


def test_corrupt_snapshot_leaves_the_live_cache_untouched_and_save_is_atomic():
    """load_mem of a truncated file returns an error and the forest answers exactly as before (the reference keeps
    self.mem when unpickling fails, lookahead_cache.py:583-587); save_mem goes through a temp file + rename."""
    import tempfile
    rs = random.Random(5)
    cache = LookaheadCache(eos_ids=[2])
    for _ in range(40):
        cache.put([rs.randrange(3, 40) for _ in range(30)], branch_length=9, mode='output', idx=-1)
    queries = [[rs.randrange(3, 40), rs.randrange(3, 40)] for _ in range(30)]

    def answers(c):
        out = []
        for q in queries:
            ids, mask, sizes = c.hier_get(q, decoding_length=32, branch_length=8, min_output_size=16)
            out.append(([int(x) for x in ids], tr.rows_of(mask), list(sizes)))
        return out
    before, st = answers(cache), cache.stats()
    with tempfile.TemporaryDirectory() as d:
        good = os.path.join(d, 'snap.latrie')
        cache.save_mem(good)
        assert os.listdir(d) == ['snap.latrie']                      # no temp file left behind
        blob = open(good, 'rb').read()
        for cut in (len(blob) // 2, len(blob) - 3, 12):
            bad = os.path.join(d, f'cut{cut}.latrie')
            open(bad, 'wb').write(blob[:cut])
            with pytest.raises(Exception):
                cache.load_mem(bad)
            assert cache.stats() == st and answers(cache) == before
        with pytest.raises(Exception):
            cache.save_mem(os.path.join(d, 'no_such_dir', 'x.latrie'))
        assert open(good, 'rb').read() == blob
        cache.load_mem(good)
        assert answers(cache) == before


def test_in_place_mutation_of_stop_words_and_eos_is_seen_like_the_reference():
    """The reference consults self.stop_words / self.eos_ids live (lookahead_cache.py:352, 396, 422); callers mutate the
    containers in place.  Checked against the oracle, which reads them per call like the reference."""
    stop = {}
    eos = [2]
    native = LookaheadCache(eos_ids=eos, stop_words=stop)
    oracle = TrieOracle(eos_ids=eos, stop_words=stop)
    rs = random.Random(11)
    seqs = [[rs.randrange(3, 25) for _ in range(24)] for _ in range(30)]
    for c in (native, oracle):
        for s in seqs[:15]:
            c.put(s, branch_length=7, mode='output', idx=-1)
    stop[7] = 1            # in place: no setter runs
    stop[11] = 1
    eos.append(5)
    for c in (native, oracle):
        for s in seqs[15:]:
            c.put(s, branch_length=7, mode='output', idx=-1)
        c.stream_put(seqs[0] + seqs[1], branch_length=7, final=False, idx=3)
    for q in ([3, 7], [9, 11], [11, 7], [4, 5], [12, 13]):
        a = native.hier_get(q, decoding_length=24, branch_length=6, min_output_size=12)
        b = oracle.hier_get(q, decoding_length=24, branch_length=6, min_output_size=12)
        assert [int(x) for x in a[0]] == [int(x) for x in b[0]] and tr.rows_of(a[1]) == tr.rows_of(b[1]) \
            and list(a[2]) == list(b[2]), q
    assert native.stats()['n_nodes'] == oracle.n_nodes()


def _mirror_sync_host(cache, img):
    """Apply la_cache_mirror_image / la_cache_mirror_patch to numpy arrays standing in for the device image (the GPU applies the
    same patch with k_trie_patch)."""
    import ctypes as C
    from painlessinferenceacceleration_amd import _lib
    from painlessinferenceacceleration_amd._lib import check, lib
    pd = C.POINTER(C.c_double)
    n, full, ni, nd = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    check(lib.la_cache_mirror_state(cache._h, C.byref(n), C.byref(full), C.byref(ni), C.byref(nd)))
    if full.value or n.value > img['cap']:
        cap = img['cap'] = max(2 * n.value, 64)
        P = img['planes']
        img['tok'], img['cstart'], img['ccount'] = (np.zeros(cap, np.int32) for _ in range(3))
        img['fo'], img['fi'] = np.zeros(cap, np.float64), np.zeros(P * cap, np.float64)
        check(lib.la_cache_mirror_image(cache._h, cap, img['tok'].ctypes.data_as(_lib.pi32), img['fo'].ctypes.data_as(pd),
                                        img['fi'].ctypes.data_as(pd), img['cstart'].ctypes.data_as(_lib.pi32),
                                        img['ccount'].ctypes.data_as(_lib.pi32)))
        img['n'] = n.value
        return 'full'
    ip, dk, dv = np.zeros(3 * ni.value + 1, np.int32), np.zeros(2 * nd.value + 1, np.int32), np.zeros(nd.value + 1, np.float64)
    check(lib.la_cache_mirror_patch(cache._h, ip.ctypes.data_as(_lib.pi32), dk.ctypes.data_as(_lib.pi32), dv.ctypes.data_as(pd)))
    img.setdefault('ccap', {})
    arrs = [img['tok'], img['cstart'], img['ccount'], img['ccap']]      # array 3: block capacities (read by device-side updates only)
    seen = set()
    for k in range(ni.value):
        a, r, v = ip[3 * k:3 * k + 3]
        assert (int(a), int(r)) not in seen
        seen.add((int(a), int(r)))
        arrs[a][r] = v
    for k in range(nd.value):
        pl, r = dk[2 * k:2 * k + 2]
        if pl == 0:
            img['fo'][r] = dv[k]
        else:
            img['fi'][(pl - 1) * img['cap'] + r] = dv[k]
    img['n'] = n.value
    return 'patch'


def _walk(tok, fo, fi, cstart, ccount):
    """canonical form of a forest in the device layout: nested (token, fo, fi, children...) with tree roots sorted by token"""
    def rec(u):
        kids = [rec(cstart[u] + k) for k in range(ccount[u])]
        return (int(tok[u]), float(fo[u]), float(fi[u]), kids)
    roots = [rec(cstart[0] + k) for k in range(ccount[0])]
    return sorted(roots, key=lambda r: r[0])


def test_incremental_mirror_equals_a_fresh_export_after_every_update():
    """The incremental device mirror (patches of changed words; child blocks that grow at the arena end) must describe the same
    forest as a fresh la_cache_export after every put / stream_put / reset_input_freqs, for every mirrored input slot; squeeze and
    fresh fall back to a full image."""
    import ctypes as C
    import sys
    from painlessinferenceacceleration_amd import _lib
    from painlessinferenceacceleration_amd._lib import check, lib
    sys.setrecursionlimit(10000)
    pd = C.POINTER(C.c_double)
    rs = random.Random(17)
    cache = LookaheadCache(eos_ids=[2], max_node=64, max_output_node=32)
    planes = [0, 1, 2]
    arr = np.asarray(planes, dtype=np.int32)
    check(lib.la_cache_mirror_enable(cache._h, arr.ctypes.data_as(_lib.pi32), len(planes)))
    img = {'cap': 0, 'planes': len(planes), 'n': 0}
    kinds = {'full': 0, 'patch': 0}
    phrases = [[rs.randrange(3, 30) for _ in range(rs.randint(3, 9))] for _ in range(12)]

    def snapshot(idx):
        n = C.c_int32()
        check(lib.la_cache_export(cache._h, idx, 0, None, None, None, None, None, C.byref(n)))
        cap = n.value
        t, c1, c2 = (np.zeros(cap, np.int32) for _ in range(3))
        f1, f2 = np.zeros(cap, np.float64), np.zeros(cap, np.float64)
        check(lib.la_cache_export(cache._h, idx, cap, t.ctypes.data_as(_lib.pi32), f1.ctypes.data_as(pd), f2.ctypes.data_as(pd),
                                  c1.ctypes.data_as(_lib.pi32), c2.ctypes.data_as(_lib.pi32), C.byref(n)))
        return _walk(t, f1, f2, c1, c2)

    for step in range(400):
        r = rs.random()
        seq = sum((phrases[rs.randrange(12)] for _ in range(rs.randint(1, 4))), [])
        if r < 0.35:
            cache.put(seq, branch_length=rs.choice([4, 9, 13]), mode='output', idx=-1)
        elif r < 0.55:
            cache.put(seq, branch_length=9, mode='input', idx=rs.choice(planes + [7]))       # slot 7 is not mirrored
        elif r < 0.9:
            cache.stream_put(seq[:rs.randint(1, 13)], branch_length=9, final=rs.random() < 0.1, idx=rs.choice(planes))
        elif r < 0.96:
            cache.reset_input_freqs(rs.choice(planes))
        elif r < 0.98:
            cache.squeeze_branch_counts()
        else:
            cache.fresh()
        if step % 3 == 0 or r >= 0.9:
            kinds[_mirror_sync_host(cache, img)] += 1
            for p, idx in enumerate(planes):
                got = _walk(img['tok'], img['fo'], img['fi'][p * img['cap']:(p + 1) * img['cap']], img['cstart'], img['ccount'])
                assert got == snapshot(idx), (step, idx)
    assert kinds['patch'] > 80 and kinds['full'] >= 1, kinds


def test_mirror_ccap_discard_and_stream_buffer_for_device_side_updates():
    """Host side of the device-side trie update (la_trie_stream_put_dev): block capacities cover their blocks and leave room to grow
    after a rebuild, blocks never overlap, the hold-back buffers read back as stream_put left them (== the oracle's _output_ids), and
    la_cache_mirror_discard drops exactly the words a replayed stream_put logged (the next patch is empty, a later update patches
    on top of the discarded ones: the numpy image that skipped the discarded patch is stale exactly at the records they touched)."""
    import ctypes as C
    from oracle.trie_oracle import TrieOracle
    from painlessinferenceacceleration_amd import _lib
    from painlessinferenceacceleration_amd._lib import check, lib
    rs = random.Random(5)
    native, oracle = LookaheadCache(eos_ids=[2]), TrieOracle(eos_ids=[2])
    planes = np.asarray([0, 1], dtype=np.int32)
    check(lib.la_cache_mirror_enable(native._h, planes.ctypes.data_as(_lib.pi32), 2))
    for _ in range(30):
        s = [rs.randrange(3, 40) for _ in range(rs.randint(4, 30))]
        for c in (native, oracle):
            c.put(s, branch_length=9, mode='output', idx=-1)
    n, full, ni, nd = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()

    def state():
        check(lib.la_cache_mirror_state(native._h, C.byref(n), C.byref(full), C.byref(ni), C.byref(nd)))
        return n.value, full.value, ni.value, nd.value

    def image():
        k = state()[0]
        tok, cs, cc, cap = (np.zeros(k, np.int32) for _ in range(4))
        fo, fi = np.zeros(k, np.float64), np.zeros(2 * k, np.float64)
        pd = C.POINTER(C.c_double)
        check(lib.la_cache_mirror_image(native._h, k, tok.ctypes.data_as(_lib.pi32), fo.ctypes.data_as(pd), fi.ctypes.data_as(pd),
                                        cs.ctypes.data_as(_lib.pi32), cc.ctypes.data_as(_lib.pi32)))
        check(lib.la_cache_mirror_ccap(native._h, k, cap.ctypes.data_as(_lib.pi32)))
        return k, tok, cs, cc, cap, fo

    def check_blocks(k, cs, cc, cap):
        spans = []
        stack = [0]
        while stack:
            u = stack.pop()
            assert 0 <= cc[u] <= cap[u]
            if cap[u]:
                assert 0 < cs[u] and cs[u] + cap[u] <= k
                spans.append((int(cs[u]), int(cs[u] + cap[u])))
            stack.extend(range(cs[u], cs[u] + cc[u]))
        spans.sort()
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), 'child blocks overlap'
        return len(spans)

    k, tok, cs, cc, cap, fo = image()
    assert check_blocks(k, cs, cc, cap) > 50
    assert all(cap[u] >= cc[u] + 1 for u in range(k) if cc[u] > 0)          # a rebuilt block leaves room to grow (count / 4, >= 1)
    assert lib.la_cache_mirror_ccap(native._h, k - 1, cap.ctypes.data_as(_lib.pi32)) == -2                    # LA_E_RANGE
    # hold-back buffers
    for idx in (0, 1):
        held = [rs.randrange(3, 40) for _ in range(5 + idx)]
        for c in (native, oracle):
            c.stream_put(held, branch_length=9, final=False, idx=idx)
    for idx in (0, 1, 7):
        buf, m = np.zeros(64, np.int32), C.c_int32()
        check(lib.la_cache_stream_buffer(native._h, idx, 64, buf.ctypes.data_as(_lib.pi32), C.byref(m)))
        assert buf[:m.value].tolist() == [int(x) for x in oracle.pending.get(idx, [])]
    assert lib.la_cache_stream_buffer(native._h, 0, 2, buf.ctypes.data_as(_lib.pi32), C.byref(m)) == -2 and m.value == 5
    # a replayed update: logged words are dropped, the record count follows
    image()                                                          # sync point: log empty
    assert state()[2:] == (0, 0)
    toks = [rs.randrange(3, 40) for _ in range(20)]
    native.stream_put(toks, branch_length=9, final=False, idx=0)
    k1, _, ni1, nd1 = state()
    assert ni1 > 0 and nd1 > 0 and k1 > k
    nrec = C.c_int32()
    check(lib.la_cache_mirror_discard(native._h, C.byref(nrec)))
    assert nrec.value == k1 and state()[2:] == (0, 0)
    native.put([rs.randrange(3, 40) for _ in range(12)], branch_length=9, mode='input', idx=1)
    assert state()[2] > 0                                            # later host updates patch on top
    k2, tok2, cs2, cc2, cap2, fo2 = image()
    assert check_blocks(k2, cs2, cc2, cap2) > 50
    # deletions (fresh / squeeze / load) make the mirror stale: discard refuses (the device image can no longer be the host's)
    native.fresh()
    assert lib.la_cache_mirror_discard(native._h, None) == -4        # LA_E_STATE


def test_stream_put_many_equals_sequential_stream_puts():
    """la_cache_stream_put_many (a batch step's accepted tokens in one native call) == the same stream_put calls one by one: same
    forest, same hold-back buffers, same answers (incl. eos cut, empty lists, a repeated slot, the final flush)."""
    rs = random.Random(11)
    a, b = LookaheadCache(eos_ids=[2]), LookaheadCache(eos_ids=[2])
    hist = [rs.randrange(3, 60) for _ in range(40)]
    for step in range(300):
        puts = []
        for idx in rs.sample(range(6), rs.randint(1, 6)):
            toks = [rs.choice(hist) if rs.random() < 0.6 else rs.randrange(2, 60) for _ in range(rs.randint(0, 14))]
            puts.append((idx, toks))
        if step % 17 == 0:
            puts.append(puts[0])                                  # the same slot twice in one call: applied in order
        final = step % 50 == 49
        a.stream_put_many(puts, branch_length=9, final=final)
        for idx, toks in puts:
            b.stream_put(toks, branch_length=9, final=final, idx=idx)
        if step % 10 == 0:
            q = [rs.choice(hist), rs.choice(hist)]
            ra, rb = a.hier_get(q, 32, 8, 0, 16, 'mix', 0), b.hier_get(q, 32, 8, 0, 16, 'mix', 0)
            assert ra[0] == rb[0] and np.array_equal(ra[1], rb[1]) and ra[2] == rb[2]
    assert a.stats() == b.stats()
    a.stream_put_many([], branch_length=9)


def test_par_get_c_abi_equals_python_layout_and_oracle():
    """la_cache_par_get (round 5: the re-layout of lookahead_cache.py:441-488 behind the C ABI) against (a) the Python
    re-layout of the native hier draft (LookaheadCache.par_layout, what round 4 shipped), (b) the oracle's par_get — ids, float64
    mask, sizes — on random tries at budgets below and above one 64-row word, and the packed row masks against the dense mask."""
    import ctypes as C
    from oracle.trie_oracle import TrieOracle
    from painlessinferenceacceleration_amd import _lib
    rng = random.Random(21)
    nat, ora = LookaheadCache(eos_ids=[None]), TrieOracle(eos_ids=[None])
    phrases = [[rng.randrange(3, 40) for _ in range(rng.randint(3, 30))] for _ in range(40)]
    for _ in range(300):
        seq = []
        while len(seq) < 60:
            seq.extend(phrases[rng.randrange(40)])
        bl_put = rng.choice([8, 13, 33])
        for c in (nat, ora):
            c.put(seq, branch_length=bl_put, mode='output', idx=-1)
    n_multi = 0
    for dl, bl in ((16, 8), (64, 12), (100, 20), (200, 32)):
        rngq = random.Random(dl)
        for _ in range(120):
            q = [rngq.randrange(3, 40) for _ in range(rngq.randint(1, 2))]
            ids, mask, sizes = nat.par_get(q, decoding_length=dl, branch_length=bl, min_output_size=dl // 2, mode='mix', idx=0)
            h_ids, h_mask, _ = nat.hier_get(q, decoding_length=dl, branch_length=bl, min_output_size=dl // 2, mode='mix', idx=0)
            p_ids, p_mask, p_sizes = LookaheadCache.par_layout(h_ids, h_mask)
            assert ids == p_ids and sizes == p_sizes and mask.dtype == np.float64 and np.array_equal(mask, p_mask), (dl, q)
            o_ids, o_mask, o_sizes = ora.par_get(q, decoding_length=dl, branch_length=bl, min_output_size=dl // 2, mode='mix', idx=0)
            assert ids == list(o_ids) and sizes == list(o_sizes) and np.array_equal(mask, np.asarray(o_mask)), (dl, q)
            # packed rows of the same call
            n = len(ids)
            W = (dl + 63) // 64
            rm = np.zeros(max(dl, 1) * W, dtype=np.uint64)
            out_ids = np.zeros(max(dl, 1), dtype=np.int32)
            sz, nsz, nn = (C.c_int32 * 2)(), C.c_int32(), C.c_int32()
            qa = (C.c_int32 * len(q))(*q)
            assert _lib.lib.la_cache_par_get(nat._h, qa, len(q), dl, bl, 0, dl // 2, 2, 0, max(dl, 1), out_ids.ctypes.data_as(_lib.pi32),
                                             rm.ctypes.data_as(_lib.pu64), None, sz, C.byref(nsz), C.byref(nn)) == 0
            assert nn.value == n and nsz.value == 1 and sz[0] == sizes[0] and out_ids[:n].tolist() == ids
            dense = np.array([[(int(rm[i * W + (j >> 6)]) >> (j & 63)) & 1 for j in range(n)] for i in range(n)], dtype=np.float64)
            assert np.array_equal(dense, mask), (dl, q)
            n_multi += int(n > 2 and mask[-1, 1] == 0)
    assert n_multi > 20           # several chains were laid out (block mask, not one lower triangle)
    # empty query: nothing to lay out
    assert nat.par_get([], decoding_length=16, branch_length=8)[0] == []
