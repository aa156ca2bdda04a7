# -*- coding: utf-8 -*-
"""CPU, world_size=2, gloo: the accepted-token all-gather keeps every rank's trie replica identical to a
single-process cache that is fed all sequences in global batch-index order (the N>1 path of bench.py)."""
import os
import random
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _streams(world, steps):
    rng = random.Random(3)
    phrases = [[rng.randrange(3, 60) for _ in range(rng.randint(3, 9))] for _ in range(10)]
    out = []
    for r in range(world):
        rr = random.Random(100 + r)
        seq = []
        for _ in range(steps):
            n = rr.randint(1, 13)
            step = []
            while len(step) < n:
                step.extend(phrases[rr.randrange(10)])
            seq.append(step[:n])
        out.append(seq)
    return out


def _worker(rank, world, port, steps, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from painlessinferenceacceleration_amd.distributed import AcceptedTokenGather
    from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
    cache = LookaheadCache(eos_ids=[None])
    b_loc = int(os.environ.get('LA_TEST_BLOC', '1'))
    g = AcceptedTokenGather('cpu', b_loc=b_loc)
    allseq = _streams(world * b_loc, steps)
    if b_loc == 1:
        mine = allseq[rank]
    else:      # this rank owns the global sequences b = i * world + rank; one token list per owned sequence and step
        mine = [[allseq[g.global_index(i)][s] for i in range(b_loc)] for s in range(steps)]
    seen = []
    if os.environ.get('LA_TEST_SPLIT_PHASE'):
        # split-phase form used by bench.py at N > 1: the gather of step s is collected during step s + 1
        pending = False
        for s in range(steps):
            if pending:
                seen.append(g.finish_into_trie(cache, branch_length=12))
            g.begin(mine[s])
            pending = True
        seen.append(g.finish_into_trie(cache, branch_length=12))
        g.begin([[] for _ in range(b_loc)] if b_loc > 1 else [])
        g.finish_into_trie(cache, branch_length=12, final=True)
    else:
        for s in range(steps):
            seen.append(g.update_trie(cache, mine[s], branch_length=12))
        g.update_trie(cache, [[] for _ in range(b_loc)] if b_loc > 1 else [], branch_length=12, final=True)
    res = []
    rng = random.Random(9)
    for _ in range(200):
        qy = [rng.randrange(3, 60) for _ in range(2)]
        ids, mask, sizes = cache.hier_get(qy, decoding_length=64, branch_length=12, min_output_size=32, mode='mix', idx=rank)
        res.append((ids, mask.tolist(), sizes))
    q.put((rank, seen, res, cache.stats()))
    dist.barrier()
    dist.destroy_process_group()


def _replicas_vs_single_process(world, b_loc, split_phase, steps):
    os.environ['LA_TEST_BLOC'] = str(b_loc)
    if split_phase:
        os.environ['LA_TEST_SPLIT_PHASE'] = '1'
    else:
        os.environ.pop('LA_TEST_SPLIT_PHASE', None)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = {}
    for _ in range(world):
        rank, seen, res, stats = q.get(timeout=600)
        outs[rank] = (seen, res, stats)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # every rank saw every sequence's tokens, in global batch-index order
    B = world * b_loc
    streams = _streams(B, steps)
    for r in range(world):
        for s in range(steps):
            assert outs[r][0][s] == [streams[k][s] for k in range(B)]
    # replicas agree with each other (queries carry idx=rank, but no input freqs exist, so drafts coincide)
    for r in range(1, world):
        assert outs[0][1] == outs[r][1], r
    # ... and with a single-process cache fed in global batch-index order
    from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
    ref = LookaheadCache(eos_ids=[None])
    for s in range(steps):
        for r in range(B):
            ref.stream_put(streams[r][s], branch_length=13, final=False, idx=r)
    for r in range(B):
        ref.stream_put([], branch_length=13, final=True, idx=r)
    assert all(ref.stats()['n_nodes'] == outs[r][2]['n_nodes'] for r in range(world))
    rng = random.Random(9)
    for i in range(200):
        qy = [rng.randrange(3, 60) for _ in range(2)]
        ids, mask, sizes = ref.hier_get(qy, decoding_length=64, branch_length=12, min_output_size=32, mode='mix', idx=0)
        assert (ids, mask.tolist(), sizes) == outs[0][1][i]


@pytest.mark.parametrize('b_loc', [1, 2])
@pytest.mark.parametrize('split_phase', [False, True])
def test_two_rank_trie_replicas_stay_identical(split_phase, b_loc):
    """world 2 x b_loc sequences per rank: every rank's trie replica == a single-process cache fed all B sequences in global
    batch-index order, in the strict and in the split-phase mode."""
    _replicas_vs_single_process(2, b_loc, split_phase, 40)


def test_eight_ranks_four_sequences_each_is_config4s_layout_in_single_process_order():
    """BASELINE config 4's exact layout — world 8, B_loc = 4: bs 32 batch-sharded over 8 ranks, rank r owning the global batch
    indices b = i * 8 + r — in strict mode: all 8 replicas == the single-process cache fed in batch-index order (the order of the
    reference's batch loop, pretrained_model_batch.py:1254-1259)."""
    _replicas_vs_single_process(8, 4, False, 10)


def _worker_native_fails(rank, world, port, q):
    """native transport asked for by auto-detection fails on ONE rank: all ranks must agree to fall back"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import warnings
    from painlessinferenceacceleration_amd import distributed as D

    def fake_init(self):
        if self.rank == 1:
            raise RuntimeError('la_comm_create: simulated failure')
        self._comm = None          # "succeeds" on rank 0 (no real communicator on CPU)

    D.AcceptedTokenGather._init_native = fake_init
    g = D.AcceptedTokenGather('cpu', b_loc=1)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        g._init_native_agreed(strict=False)
    out = g.gather([10 + rank, 20 + rank])
    q.put((rank, g._comm is None, len(w) == 1, out))
    dist.barrier()
    dist.destroy_process_group()


def test_native_transport_failure_on_one_rank_falls_back_everywhere():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_native_fails, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, no_comm, warned, out in got:
        assert no_comm and warned and out == [[10, 20], [11, 21]], (rank, no_comm, warned, out)


# ---------------------------------------------------------------------------------------------------------------------------
# Product-level sharded decoding (round 5): lookahead_generation() of both loops under decoding_kwargs['gather'].
# ---------------------------------------------------------------------------------------------------------------------------
def _sharded_setup(world, b_loc):
    """prompts + greedy truths of all B = world * b_loc sequences on the tiny decisive model (every rank computes the same)"""
    from types import SimpleNamespace
    from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
    from tests.oracle_engine import OracleBatchEngine, OracleEngine
    from tests.tiny_model import tiny_decisive_weights, tiny_shape
    shape = tiny_shape()
    B, P, n_new = world * b_loc, 20, 60
    rs = np.random.RandomState(17)
    prompts = rs.randint(3, shape.vocab, size=(B, P)).astype(np.int64)
    gen = SimpleNamespace(eos_token_id=None, pad_token_id=0, return_dict_in_generate=True)
    if b_loc == 1:
        from painlessinferenceacceleration_amd.pretrained_model import LookaheadPreTrainedModel as Mixin

        class M(Mixin):
            def __init__(self):
                self.engine = OracleEngine(shape, tiny_decisive_weights(0, torch.float32), max_length=256)
                self.generation_config = gen
                self.lookahead_cache = LookaheadCache(eos_ids=[None])
        m = M()
        truths = [m.greedy_search(torch.from_numpy(prompts[b:b + 1]), P + n_new, eos_token_id=None)[0].tolist()[P:] for b in range(B)]
    else:
        from painlessinferenceacceleration_amd.pretrained_model_batch import LookaheadPreTrainedModel as Mixin

        class M(Mixin):
            def __init__(self):
                self.engine = OracleBatchEngine(shape, tiny_decisive_weights(0, torch.float32), max_length=256, n_slots=max(b_loc, B), max_blocks=max(b_loc, 2))
                self.generation_config = gen
                self.lookahead_cache = LookaheadCache(eos_ids=[None])
        m = M()
        truths = m.greedy_search(torch.from_numpy(prompts), P + n_new, eos_token_id=None)[:, P:].tolist()
    return shape, prompts, truths, M, P, n_new


def _warm(cache, prompts, truths, vocab):
    from tests.tiny_model import noisy_copies
    for b in range(len(truths)):                  # bench.py's warm-up: different noise per sequence -> different accept lengths per rank
        for c in noisy_copies(prompts[b, -2:].tolist() + truths[b], 6, 0.25 + 0.1 * (b % 2), vocab, seed=40 + b):
            cache.put(c, branch_length=13, mode='output', idx=-1)


def _queries(cache, B, vocab):
    rng = random.Random(5)
    res = []
    for _ in range(150):
        qy = [rng.randrange(3, vocab) for _ in range(2)]
        ids, mask, sizes = cache.hier_get(qy, decoding_length=64, branch_length=12, min_output_size=32, mode='mix', idx=rng.randrange(B))
        res.append((list(ids), mask.tolist(), list(sizes)))
    return res


def _worker_sharded(rank, world, port, b_loc, mode, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from painlessinferenceacceleration_amd.distributed import AcceptedTokenGather
    shape, prompts, truths, M, P, n_new = _sharded_setup(world, b_loc)
    m = M()
    _warm(m.lookahead_cache, prompts, truths, shape.vocab)
    g = AcceptedTokenGather('cpu', b_loc=b_loc, branch_length=12, mode=mode)
    record, inner = [], g.finish

    def finish():
        per = inner()
        record.append([list(t) for t in per])
        return per
    g.finish = finish
    mine = [g.global_index(i) for i in range(b_loc)]
    dk = {'use_lookahead': True, 'decoding_mode': 'hier', 'decoding_length': 64, 'branch_length': 12, 'max_query_length': 2,
          'stop_words': {}, 'gather': g}
    if b_loc > 1:
        dk['per_sample_budget'] = True
    out = m.lookahead_generation(torch.from_numpy(prompts[mine]), stopping_criteria=P + n_new, eos_token_id=[None], pad_token_id=0,
                                 return_dict_in_generate=True, decoding_kwargs=dk)
    q.put((rank, out.sequences.cpu().numpy().tolist(), out.kwargs['edls'], record, _queries(m.lookahead_cache, world * b_loc, shape.vocab),
           m.lookahead_cache.stats()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('b_loc,mode', [(1, 'strict'), (1, 'split-phase'), (2, 'strict'), (2, 'split-phase')])
def test_lookahead_generation_sharded_over_two_ranks(b_loc, mode):
    """decoding_kwargs['gather'] in BOTH product loops (bs=1: pretrained_model.py, B_loc=2: pretrained_model_batch.py), 2 gloo ranks:
    (1) every rank's output == plain greedy decoding of its own prompts (lookahead is lossless, sharded or not);
    (2) the ranks finish at different steps and still leave together: same number of collectives on both, the last one all-DONE;
    (3) the trie replicas are identical (150 queries over every input-frequency plane, node counts);
    (4) they equal a single-process cache that received the same prompts and the same per-step token lists in global batch-index
        order (the order of pretrained_model_batch.py:1254-1259) and the final flush of all B sequences."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded, args=(r, world, port, b_loc, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = {}
    for _ in range(world):
        got = q.get(timeout=900)
        outs[got[0]] = got[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    shape, prompts, truths, M, P, n_new = _sharded_setup(world, b_loc)
    B = world * b_loc
    for r in range(world):
        seqs = outs[r][0]
        for i in range(b_loc):
            b = i * world + r
            assert seqs[i][:P] == prompts[b].tolist() and seqs[i][P:P + n_new] == truths[b][:n_new], (r, i)
        assert np.mean(outs[r][1][b_loc:]) > 2.0                  # drafts were accepted: the steps carried real trees
    rec0, rec1 = outs[0][2], outs[1][2]
    assert rec0 == rec1 and len(rec0) >= 3                        # same collectives, same contents, on both ranks
    # a rank that had finished kept serving the collective with empty lists while the other one still emitted tokens (drain)
    assert any(any(len(t) == 0 for t in per) and any(len(t) > 0 for t in per) for per in rec0)
    assert outs[0][3] == outs[1][3] and outs[0][4]['n_nodes'] == outs[1][4]['n_nodes']
    # (4) single-process cache, global batch-index order
    from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
    ref = LookaheadCache(eos_ids=[None])
    _warm(ref, prompts, truths, shape.vocab)
    for b in range(B):
        ref.put(prompts[b, 1:].tolist() if b_loc == 1 else prompts[b, 1:-1].tolist(), branch_length=13, mode='input', idx=b)
    for per in rec0:
        assert len(per) == B
        for b, toks in enumerate(per):
            ref.stream_put(toks, branch_length=13, final=False, mode='output', idx=b)
    for b in range(B):
        ref.stream_put([], branch_length=13, final=True, mode='output', idx=b)
    assert ref.stats()['n_nodes'] == outs[0][4]['n_nodes']
    assert _queries(ref, B, shape.vocab) == outs[0][3]


# ---------------------------------------------------------------------------------------------------------------------------
# Failure path of a sharded request (round 6, advisor finding): one rank raises mid-request.
# ---------------------------------------------------------------------------------------------------------------------------
def _worker_sharded_failure(rank, world, port, b_loc, mode, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from painlessinferenceacceleration_amd.distributed import AcceptedTokenGather
    shape, prompts, truths, M, P, n_new = _sharded_setup(world, b_loc)
    m = M()
    _warm(m.lookahead_cache, prompts, truths, shape.vocab)
    g = AcceptedTokenGather('cpu', b_loc=b_loc, branch_length=12, mode=mode, timeout_s=120)
    mine = [g.global_index(i) for i in range(b_loc)]
    dk = {'use_lookahead': True, 'decoding_mode': 'hier', 'decoding_length': 64, 'branch_length': 12, 'max_query_length': 2,
          'stop_words': {}, 'gather': g}
    if b_loc > 1:
        dk['per_sample_budget'] = True
    calls = {'n': 0}
    if rank == 1:                                   # rank 1's engine fails on its third verify step
        eng = m.engine
        name = 'step' if b_loc == 1 else 'mstep'
        if b_loc > 1 and hasattr(eng, 'mstep_async'):
            name = 'mstep_async'
        inner = getattr(eng, name)

        def failing(*a, **kw):
            calls['n'] += 1
            if calls['n'] == 3:
                raise RuntimeError('simulated engine failure on rank 1')
            return inner(*a, **kw)
        setattr(eng, name, failing)
    err, seqs = None, None
    try:
        out = m.lookahead_generation(torch.from_numpy(prompts[mine]), stopping_criteria=P + n_new, eos_token_id=[None], pad_token_id=0,
                                     return_dict_in_generate=True, decoding_kwargs=dk)
        seqs = out.sequences.cpu().numpy().tolist()
    except RuntimeError as e:
        err = str(e)
    failed_seen = list(g.failed_ranks)
    # the SAME gather object serves a second request on both ranks: nothing of the failed one may leak into it
    for name in ('step', 'mstep', 'mstep_async'):
        if rank == 1 and calls['n'] and hasattr(m.engine, name) and getattr(m.engine, name).__name__ == 'failing':
            delattr(m.engine, name)
    out2 = m.lookahead_generation(torch.from_numpy(prompts[mine]), stopping_criteria=P + n_new, eos_token_id=[None], pad_token_id=0,
                                  return_dict_in_generate=True, decoding_kwargs=dict(dk))
    q.put((rank, err, seqs, out2.sequences.cpu().numpy().tolist(), (failed_seen, list(g.failed_ranks)), _queries(m.lookahead_cache, world * b_loc, shape.vocab),
           m.lookahead_cache.stats()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('b_loc,mode', [(1, 'strict'), (1, 'split-phase'), (2, 'strict'), (2, 'split-phase')])
def test_sharded_request_survives_a_rank_that_raises_mid_request(b_loc, mode):
    """One rank's engine raises on its third verify step (both product loops, both gather modes).  The failing rank must keep its side of
    the per-step collective (DONE | FAILED contributions until every rank has finished, then the flush of all B sequences) before it
    re-raises, so that (1) the healthy rank finishes its sequences == plain greedy instead of blocking in the collective, (2) it learns
    which rank failed, (3) both processes exit, (4) a SECOND request through the same gather object runs cleanly on both ranks
    (no stale pending gather / all_done), == plain greedy on both, and (5) the replicas still answer identically afterwards."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded_failure, args=(r, world, port, b_loc, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = {}
    for _ in range(world):
        got = q.get(timeout=900)
        outs[got[0]] = got[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    shape, prompts, truths, M, P, n_new = _sharded_setup(world, b_loc)
    err0, seqs0, second0, failed0, res0, st0 = outs[0]
    err1, seqs1, second1, failed1, res1, st1 = outs[1]
    assert err0 is None and err1 is not None and 'simulated engine failure' in err1
    for i in range(b_loc):
        b = i * world
        assert seqs0[i][:P] == prompts[b].tolist() and seqs0[i][P:P + n_new] == truths[b][:n_new]
    for r, second in ((0, second0), (1, second1)):
        for i in range(b_loc):
            b = i * world + r
            assert second[i][:P] == prompts[b].tolist() and second[i][P:P + n_new] == truths[b][:n_new], (r, i)
    assert failed0[0] == [1] and failed1[0] == [1]         # every rank learnt which rank failed (FAILED bit of its count words) ...
    assert failed0[1] == [] and failed1[1] == []           # ... and the second request's begin_request() cleared it
    assert res0 == res1 and st0['n_nodes'] == st1['n_nodes']
