# -*- coding: utf-8 -*-
"""Batch sharding across the GPUs of one node: the only exchange on the path.

Sequences are independent: rank r owns sequence r, a full model replica, its KV cache and a trie
replica.  Per verify step every rank contributes `int32[branch_length + 2]` = {n, tokens...}; one all-gather
(RCCL over xGMI on GPU, gloo in the CPU tests) hands every rank every sequence's accepted tokens, which are then
applied to the local trie in GLOBAL batch-index order — exactly the order in which the reference's single-process
batch loop calls stream_put (common/pretrained_model_batch.py:1254-1259) — so all replicas stay identical.
The message is 56 B per sequence: latency-bound, never bandwidth-bound.
"""
import torch
import torch.distributed as dist

SLOT = 16      # int32 words per sequence: count + up to branch_length+1 (<= 13) tokens, padded


class AcceptedTokenGather(object):
    def __init__(self, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.local_only = not dist.is_initialized()        # a 1-rank process group still runs the collective
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device(device)
        self._in = torch.zeros(SLOT, dtype=torch.int32, device=self.device)
        self._out = torch.zeros(SLOT * self.world, dtype=torch.int32, device=self.device)

    def gather(self, tokens):
        """tokens: this rank's accepted tokens of the step -> list (per rank, in rank order) of token lists."""
        assert len(tokens) < SLOT
        if self.local_only:
            return [list(tokens)]
        buf = torch.zeros(SLOT, dtype=torch.int32)
        buf[0] = len(tokens)
        buf[1:1 + len(tokens)] = torch.tensor(tokens, dtype=torch.int32)
        self._in.copy_(buf)
        dist.all_gather_into_tensor(self._out, self._in, group=self.group)
        allv = self._out.cpu().view(self.world, SLOT)
        return [allv[r, 1:1 + int(allv[r, 0])].tolist() for r in range(self.world)]

    # ---- split-phase form: the gather of step k overlaps the verify step k+1 -------------------------------------
    def begin(self, tokens):
        """Start the all-gather of this rank's accepted tokens (asynchronous; one outstanding gather at a time)."""
        assert len(tokens) < SLOT and getattr(self, '_work', None) is None
        self._mine = list(tokens)
        if self.local_only:
            self._work = True
            return
        if not hasattr(self, '_stage'):
            pin = self.device.type == 'cuda'
            self._stage = torch.zeros(SLOT, dtype=torch.int32, pin_memory=pin)
            self._host = torch.zeros(SLOT * self.world, dtype=torch.int32, pin_memory=pin)
        self._stage.zero_()
        self._stage[0] = len(tokens)
        if tokens:
            self._stage[1:1 + len(tokens)] = torch.tensor(tokens, dtype=torch.int32)
        self._in.copy_(self._stage, non_blocking=True)
        self._work = dist.all_gather_into_tensor(self._out, self._in, group=self.group, async_op=True)

    def finish(self):
        """-> per-rank token lists of the gather started by begin()."""
        assert getattr(self, '_work', None) is not None
        work, self._work = self._work, None
        if self.local_only:
            return [self._mine]
        work.wait()
        self._host.copy_(self._out)
        allv = self._host.view(self.world, SLOT)
        return [allv[r, 1:1 + int(allv[r, 0])].tolist() for r in range(self.world)]

    def finish_into_trie(self, cache, branch_length, final=False):
        per_rank = self.finish()
        for r, toks in enumerate(per_rank):
            cache.stream_put(toks, branch_length=branch_length + 1, final=final, mode='output', idx=r)
        return per_rank

    def update_trie(self, cache, tokens, branch_length, final=False):
        """all-gather + stream_put for every sequence (idx = global batch index) in rank order."""
        per_rank = self.gather(tokens)
        for r, toks in enumerate(per_rank):
            cache.stream_put(toks, branch_length=branch_length + 1, final=final, mode='output', idx=r)
        return per_rank
