# -*- coding: utf-8 -*-
"""Batch sharding across the GPUs of one node: the only exchange on the path.

Sequences are independent: with B sequences and N ranks, rank r owns the B_loc = B / N sequences with global
batch index b = i * N + r (b mod N == r), a full model replica, their KV caches and a trie replica.  Per verify step every
rank contributes int32[B_loc][SLOT] = {n, tokens...} per sequence; ONE all-gather hands every rank every sequence's accepted
tokens, which are then applied to the local trie in GLOBAL batch-index order — the order in which the reference's
single-process batch loop calls stream_put (common/pretrained_model_batch.py:1254-1259) — so all replicas stay identical to
that run.  The message is 64 B per sequence: latency-bound, never bandwidth-bound.

Transport: on GPUs the all-gather is la_gather_accepted (C ABI, ncclAllGather of RCCL over xGMI on the engine's stream,
communicator created from an id broadcast through torch.distributed); on CPU (gloo tests) and as a fallback it is
torch.distributed.all_gather_into_tensor.  Two modes, label every number with the one it was measured in:
  strict       gather(...) / update_trie(...): blocking, every rank's drafts see every token of the step — trie state at query
               time equals the reference's single-process order (bit-exact retrieval);
  split-phase  begin(...) after step k, finish_into_trie(...) during step k+1: the gather and the host-side trie updates
               overlap the next verify step; drafts see a step's tokens one step later (emitted tokens are unaffected:
               verification is lossless), replicas still identical to each other.

Product entry point (round 5): both decoding loops honour `decoding_kwargs['gather'] = AcceptedTokenGather(..., mode=...)` —
`lookahead_generation()` of pretrained_model.py (one sequence per rank) and of pretrained_model_batch.py (B_loc sequences per
rank) then key every trie call with the GLOBAL batch index, replace their per-step stream_put by step_update() / overlap(), keep
serving the collective after their own sequences have finished (drain(): every rank makes the same number of collective calls; a
DONE bit rides in the count word) and flush all B sequences in batch-index order at the end (flush()).  bench.py --gpus N drives
the same four calls from its timed loop.
"""
import ctypes as C
import datetime
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from ._lib import check, lib


def slot_words(branch_length):
    """int32 words per sequence: count + up to branch_length + 1 tokens, rounded up to 16 words (64-byte messages)."""
    return max(16, (int(branch_length) + 2 + 15) // 16 * 16)


DONE_BIT = 1 << 30          # count word: this rank has finished all its sequences (it keeps contributing empty lists)
FAILED_BIT = 1 << 29        # count word: this rank's request FAILED (it is draining like a finished rank; the others learn which rank it was)


class GatherTimeout(RuntimeError):
    """A per-step collective did not complete within the gather's timeout: a peer is gone.  The gather is marked broken — no
    further collective is attempted, the loops flush their local trie state and re-raise."""


class AcceptedTokenGather(object):
    def __init__(self, device, group=None, b_loc=1, branch_length=12, native=None, mode='split-phase', timeout_s=None):
        assert mode in ('strict', 'split-phase')
        self.mode = mode
        # every wait on a collective is bounded (a rank that died mid-request must not park the others forever in drain())
        self.timeout_s = float(timeout_s if timeout_s is not None else os.environ.get('LA_GATHER_TIMEOUT_S', '300'))
        self.broken = False                                # a collective timed out / failed: no further collective is attempted
        self.failed_ranks = []                             # ranks whose FAILED bit was seen during this request
        self.stats = {'collectives': 0, 'wait_s': 0.0, 'wait_s_max': 0.0}     # host time spent waiting for collectives (finish())
        self.branch_length = int(branch_length)
        self.all_done = False                              # every rank's DONE bit was set in the last collected gather
        self._pending = False
        self.group = group
        self.local_only = not dist.is_initialized()        # a 1-rank process group still runs the collective
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device(device)
        self.b_loc = int(b_loc)
        self.slot = slot_words(branch_length)
        self.max_tokens = self.slot - 1
        n = self.b_loc * self.slot
        self._in = torch.zeros(n, dtype=torch.int32, device=self.device)
        self._out = torch.zeros(n * self.world, dtype=torch.int32, device=self.device)
        pin = self.device.type == 'cuda'
        self._stage = torch.zeros(n, dtype=torch.int32, pin_memory=pin)
        self._host = torch.zeros(n * self.world, dtype=torch.int32, pin_memory=pin)
        self._work = None
        self._comm = None
        self._stream = None
        native_given = native is not None
        if native is None:
            native = self.device.type == 'cuda' and not self.local_only and dist.get_backend(group) == 'nccl'
        if native:
            self._init_native_agreed(strict=native_given)

    # ---- native RCCL communicator (la_comm_*): id from rank 0, distributed through torch.distributed ---------------------
    def _init_native_agreed(self, strict):
        """Create the native communicator on every rank, then AGREE on it: one all-reduce(MIN) of the success flags through
        torch.distributed.  If any rank failed, every rank drops its communicator and the gather runs through
        torch.distributed (still RCCL on GPUs) — a mixed set of transports would deadlock in the first collective.
        strict (native=True was asked for explicitly): a failure raises instead."""
        err = None
        try:
            self._init_native()
        except Exception as e:                      # noqa: BLE001 - any failure means "no native transport on this rank"
            err = e
            self._comm = None
        if not self.local_only:
            ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=self.device if dist.get_backend(self.group) == 'nccl' else 'cpu')
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            all_ok = bool(int(ok.item()))
        else:
            all_ok = err is None
        if all_ok:
            return
        if self._comm:
            lib.la_comm_destroy(self._comm)
            self._comm = None
        if strict:
            raise err if err is not None else _lib.LookaheadHipError('la_comm_create failed on another rank')
        import warnings
        warnings.warn(f'native RCCL communicator unavailable ({err if err else "failed on another rank"}): '
                      'accepted-token gather falls back to torch.distributed.all_gather_into_tensor')

    def _init_native(self):
        # 128 id bytes + 1 "rank 0 has an id" flag.  Rank 0 ALWAYS takes part in the broadcast, also when it could not make an
        # id (librccl missing): every rank must run the same collectives before the agreement all-reduce, or the others
        # would sit in the broadcast while rank 0 is already in the all-reduce.
        idbuf = torch.zeros(129, dtype=torch.uint8)
        id_err = None
        if self.rank == 0:
            try:
                arr = (C.c_uint8 * 128)()
                check(lib.la_comm_unique_id(arr), 'comm_unique_id')
                idbuf[:128] = torch.tensor(list(arr), dtype=torch.uint8)
                idbuf[128] = 1
            except Exception as e:                  # noqa: BLE001 - reported after the broadcast
                id_err = e
        if not self.local_only:
            t = idbuf.to(self.device) if dist.get_backend(self.group) == 'nccl' else idbuf
            dist.broadcast(t, src=0, group=self.group)
            idbuf = t.cpu()
        if id_err is not None:
            raise id_err
        if int(idbuf[128]) != 1:
            raise _lib.LookaheadHipError('rank 0 could not create an RCCL unique id')
        arr = (C.c_uint8 * 128)(*idbuf[:128].tolist())
        torch.cuda.set_device(self.device)
        self._comm = lib.la_comm_create(arr, self.world, self.rank)
        if not self._comm:
            raise _lib.LookaheadHipError(f'la_comm_create: {_lib.last_error()}')
        self._stream = torch.cuda.Stream(self.device)
        self._done = torch.cuda.Event()

    @property
    def transport(self):
        """which code path the all-gather runs through (recorded by bench.py as config.gather_transport)"""
        if self.local_only:
            return 'none (single process)'
        return 'la_gather_accepted(rccl)' if self._comm else f'torch.distributed({dist.get_backend(self.group)})'

    def __del__(self):
        if getattr(self, '_comm', None):
            lib.la_comm_destroy(self._comm)
            self._comm = None

    # ---- packing -------------------------------------------------------------------------------------------------------
    def _lists(self, tokens):
        if self.b_loc == 1 and (len(tokens) == 0 or not isinstance(tokens[0], (list, tuple, np.ndarray))):
            tokens = [tokens]                  # bs=1 per rank: a flat token list
        assert len(tokens) == self.b_loc, f'{len(tokens)} token lists for b_loc={self.b_loc}'
        return [list(t) for t in tokens]

    def _pack(self, lists, done=False, failed=False):
        buf = self._stage
        for t in lists:                        # validated BEFORE anything is staged: a refusal leaves no half-written message behind
            if len(t) > self.max_tokens:
                raise ValueError(f'{len(t)} accepted tokens do not fit the {self.slot}-word gather slot; construct '
                                 f'AcceptedTokenGather with branch_length >= {len(t) - 1}')
        buf.zero_()
        v = buf.view(self.b_loc, self.slot)
        for i, t in enumerate(lists):
            v[i, 0] = len(t) | (DONE_BIT if done else 0) | (FAILED_BIT if failed else 0)
            if t:
                v[i, 1:1 + len(t)] = torch.tensor(t, dtype=torch.int32)
        return buf

    def _unpack(self, host):
        """-> token lists in GLOBAL batch-index order b = i * world + r (and self.all_done: every rank flagged DONE)."""
        allv = host.view(self.world, self.b_loc, self.slot)
        self.all_done = all((int(allv[r, 0, 0]) & DONE_BIT) != 0 for r in range(self.world))
        for r in range(self.world):
            if (int(allv[r, 0, 0]) & FAILED_BIT) and r not in self.failed_ranks:
                self.failed_ranks.append(r)
        return [allv[r, i, 1:1 + (int(allv[r, i, 0]) & (FAILED_BIT - 1))].tolist() for i in range(self.b_loc) for r in range(self.world)]

    def global_index(self, i):
        """global batch index of this rank's i-th sequence"""
        return i * self.world + self.rank

    # ---- strict (blocking) form -------------------------------------------------------------------------------------------
    def gather(self, tokens, done=False, failed=False):
        """tokens: this rank's accepted tokens of the step (b_loc lists; a flat list when b_loc == 1)
        -> token lists of all B sequences in global batch-index order."""
        self.begin(tokens, done=done, failed=failed)
        return self.finish()

    # ---- split-phase form: the gather of step k overlaps the verify step k+1 ---------------------------------------------
    def begin(self, tokens, done=False, failed=False):
        """Start the all-gather of this rank's accepted tokens (asynchronous; one outstanding gather at a time).  done: this rank
        has no live sequence left (it keeps calling with empty lists until all_done, see drain())."""
        assert self._work is None, 'one outstanding gather at a time'
        if self.broken:
            raise GatherTimeout('the accepted-token gather is broken (an earlier collective timed out)')
        lists = self._lists(tokens)
        if self.local_only:
            self._mine = lists
            self._work = True
            self._mine_done = bool(done)
            return
        self._pack(lists, done, failed)        # raises before any state changes when a list does not fit its slot
        self._mine = lists
        if self._comm:
            with torch.cuda.stream(self._stream):
                self._in.copy_(self._stage, non_blocking=True)
                check(lib.la_gather_accepted(self._comm, C.c_void_p(self._stream.cuda_stream), self._in.data_ptr(), self.b_loc,
                                             self.slot, self._out.data_ptr()), 'gather_accepted')
                self._host.copy_(self._out, non_blocking=True)
                self._done.record(self._stream)
            self._work = True
        else:
            self._in.copy_(self._stage, non_blocking=True)
            self._work = dist.all_gather_into_tensor(self._out, self._in, group=self.group, async_op=True)

    def finish(self):
        """-> token lists (global batch-index order) of the gather started by begin()."""
        assert self._work is not None
        work, self._work = self._work, None
        if self.local_only:
            self.all_done = self._mine_done
            return self._mine
        t0 = time.perf_counter()
        try:
            if self._comm:
                deadline = t0 + self.timeout_s
                while not self._done.query():              # bounded wait: a dead peer must not hang this rank
                    if time.perf_counter() > deadline:
                        raise GatherTimeout(f'la_gather_accepted did not complete within {self.timeout_s:.0f} s')
                    time.sleep(0) if time.perf_counter() - t0 < 1e-3 else time.sleep(2e-4)
            else:
                ok = work.wait(datetime.timedelta(seconds=self.timeout_s))
                if ok is False:
                    raise GatherTimeout(f'all_gather_into_tensor did not complete within {self.timeout_s:.0f} s')
                self._host.copy_(self._out)
        except Exception as e:
            self.broken = True
            self._pending = False
            if isinstance(e, GatherTimeout):
                raise
            raise GatherTimeout(f'accepted-token gather failed: {e!r}') from e
        dt = time.perf_counter() - t0
        self.stats['collectives'] += 1
        self.stats['wait_s'] += dt
        self.stats['wait_s_max'] = max(self.stats['wait_s_max'], dt)
        return self._unpack(self._host)

    def finish_into_trie(self, cache, branch_length, final=False):
        per_seq = self.finish()
        for b, toks in enumerate(per_seq):
            cache.stream_put(toks, branch_length=branch_length + 1, final=final, mode='output', idx=b)
        return per_seq

    def update_trie(self, cache, tokens, branch_length, final=False, done=False, failed=False):
        """strict mode: all-gather + stream_put for every sequence (idx = global batch index) in batch-index order."""
        per_seq = self.gather(tokens, done=done, failed=failed)
        for b, toks in enumerate(per_seq):
            cache.stream_put(toks, branch_length=branch_length + 1, final=final, mode='output', idx=b)
        return per_seq

    # ---- what a decoding loop calls (lookahead_generation of both loops, bench.py): mode-independent call sites ---------------
    @property
    def n_sequences(self):
        return self.world * self.b_loc

    def begin_request(self):
        """Start of a request (both decoding loops call it before anything else): a gather object that is reused must not carry the
        previous request's state into this one — an un-collected split-phase gather, a stale all_done, the failed-rank list."""
        if self.broken:
            raise GatherTimeout('the accepted-token gather is broken (an earlier collective timed out): build a new one')
        if self._work is not None:             # a gather nobody collected (the previous request ended abnormally): collect and drop it
            try:
                self.finish()
            except GatherTimeout:
                raise
        self._pending = False
        self.all_done = False
        self.failed_ranks = []
        self.stats = {'collectives': 0, 'wait_s': 0.0, 'wait_s_max': 0.0}

    def abort_request(self, cache, branch_length=None):
        """The failure path of a sharded request (called from the loops' exception handlers, then the exception is re-raised): this
        rank stops decoding but keeps its side of the protocol — it flags DONE | FAILED with empty contributions, keeps serving the
        per-step collective until every rank has finished (drain), and runs the final flush for all B sequences — so the other ranks
        finish their sequences normally, every replica ends the request in the same state, and nobody waits for a rank that left.
        If the collective itself is what failed (timeout: a peer is gone) only the local flush runs."""
        bl = self.branch_length if branch_length is None else branch_length
        try:
            if not self.broken:
                if self._pending or self._work is not None:
                    self._pending = False
                    if self._work is not None:
                        self.finish_into_trie(cache, bl)
                if not self.all_done:
                    empty = [[] for _ in range(self.b_loc)] if self.b_loc > 1 else []
                    self.update_trie(cache, empty, bl, done=True, failed=True)
                    while not self.all_done:
                        self.update_trie(cache, empty, bl, done=True, failed=True)
        except GatherTimeout:
            pass
        finally:
            self._pending = False
            self._work = None
            self.flush(cache, bl)

    def exchange_prompts(self, local_prompts):
        """Once per request, off the hot path: every rank's prompt token lists -> all B lists in global batch-index order, so that
        each replica holds every sequence's input frequencies (the reference's single-process loop puts all prompts,
        pretrained_model_batch.py:1204-1207) and node counts / squeeze decisions agree across replicas."""
        local_prompts = [list(map(int, p)) for p in local_prompts]
        assert len(local_prompts) == self.b_loc
        if self.local_only:
            return local_prompts
        got = [None] * self.world
        dist.all_gather_object(got, local_prompts, group=self.group)
        return [got[r][i] for i in range(self.b_loc) for r in range(self.world)]

    def step_update(self, cache, tokens, branch_length=None, done=False):
        """After a verify step: this rank's accepted tokens (b_loc lists; [] for a sequence that emitted nothing or has retired).
        strict: blocking all-gather + every sequence's stream_put in global batch-index order, now.  split-phase: start the
        all-gather; overlap() — called once the NEXT verify pass is queued — collects it and runs the puts while the GPU works."""
        bl = self.branch_length if branch_length is None else branch_length
        if self.mode == 'strict':
            self.update_trie(cache, tokens, bl, done=done)
        else:
            if self._pending:                  # a step that had no overlap point (e.g. it finished the sequence): collect now
                self.overlap(cache, bl)
            self.begin(tokens, done=done)
            self._pending = True

    def overlap(self, cache, branch_length=None):
        """split-phase: collect the gather begun after the previous step and apply it (call right after queueing a verify pass)."""
        if self._pending:
            self._pending = False
            self.finish_into_trie(cache, self.branch_length if branch_length is None else branch_length)

    def drain(self, cache, branch_length=None):
        """This rank's sequences are finished (its last step_update carried done=True): keep taking part in the per-step
        collective with empty contributions, applying the other ranks' tokens, until every rank has flagged DONE — all ranks
        make the same number of collective calls and leave together."""
        bl = self.branch_length if branch_length is None else branch_length
        self.overlap(cache, bl)
        empty = [[] for _ in range(self.b_loc)] if self.b_loc > 1 else []
        while not self.all_done:
            self.update_trie(cache, empty, bl, done=True)

    def flush(self, cache, branch_length=None):
        """End of the request on every rank: the final flush of all B sequences in batch-index order
        (pretrained_model_batch.py:1288-1290)."""
        bl = self.branch_length if branch_length is None else branch_length
        for b in range(self.n_sequences):
            cache.stream_put([], branch_length=bl + 1, final=True, mode='output', idx=b)
