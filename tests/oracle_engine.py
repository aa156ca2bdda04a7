# -*- coding: utf-8 -*-
"""TEST-ONLY stand-in for LlamaVerifyEngine built on the oracle forward (CPU).  It lets the `-m "not gpu"` suite
exercise the host logic of pretrained_model.py (draft retrieval, step bookkeeping, stop conditions, trie updates)
without a GPU.  It is never importable from the product package."""
import numpy as np
import torch

from oracle import llama_oracle as lo


class OracleEngine(object):
    def __init__(self, shape, state_dict, max_length=512):
        self.shape = shape
        self.model = lo.OracleLlama(shape, state_dict)
        self.max_keys = ((max_length + 65 + 31) // 32) * 32
        self.reset()

    def reset(self):
        self.past = None
        self.n_keys = 0
        self.last_argmax = None

    def _forward(self, ids, mask_rows):
        T = len(ids)
        tree = np.array([[(int(mask_rows[i]) >> j) & 1 for j in range(T)] for i in range(T)], dtype=np.int64)
        full = torch.cat([torch.ones((T, self.n_keys), dtype=torch.long), torch.from_numpy(tree)], 1)
        logits, past = self.model.forward(torch.tensor([int(x) for x in ids]), full, self.past)
        return logits, past, tree

    def prefill(self, prompt_ids, eager=False, fast=None):
        tok = None
        for s in range(0, len(prompt_ids), 64):
            blk = prompt_ids[s:s + 64]
            rows = [(2 << t) - 1 for t in range(len(blk))]
            logits, past, _ = self._forward(blk, rows)
            self._logits = logits
            self.past = past
            self.n_keys += len(blk)
            tok = int(torch.argmax(logits[-1].float()))
        return tok

    device = torch.device('cpu')

    def verify_only(self, ids, rowmask, eager=False):
        self._logits, self._pending, _ = self._forward(ids, rowmask)

    def logits(self):
        return self._logits

    def commit(self, rows):
        keep = list(range(self.n_keys)) + [self.n_keys + int(r) for r in rows]
        idx = torch.tensor(keep, dtype=torch.long)
        self.past = [(k[:, idx], v[:, idx]) for k, v in self._pending]
        self.n_keys += len(rows)

    def step(self, ids, rowmask, mode=0, eager=False):
        logits, past, tree = self._forward(ids, rowmask)
        self._logits = logits
        am = [int(x) for x in torch.argmax(logits.float(), -1)]
        self.last_argmax = am
        toks, rows = lo.accept_scan([int(x) for x in ids], tree, am)
        keep = list(range(self.n_keys)) + [self.n_keys + r for r in rows]
        idx = torch.tensor(keep, dtype=torch.long)
        self.past = [(k[:, idx], v[:, idx]) for k, v in past]
        self.n_keys += len(rows)
        return toks, len(rows)


class OracleBatchEngine(object):
    """TEST-ONLY stand-in for LlamaVerifyEngine(n_slots=N): the cursor-batch surface (reset_slot / bprefill_many /
    bstep) on the oracle forward, one KV list per slot."""

    def __init__(self, shape, state_dict, max_length=512, n_slots=4):
        self.shape = shape
        self.model = lo.OracleLlama(shape, state_dict)
        self.max_keys = ((max_length + 65 + 31) // 32) * 32
        self.n_slots = n_slots
        self.reset_slot(-1)

    def reset_slot(self, slot):
        if slot < 0:
            self.past = [None] * self.n_slots
            self.slot_keys = [0] * self.n_slots
        else:
            self.past[slot] = None
            self.slot_keys[slot] = 0

    def bstep(self, segments, eager=False):
        assert sum(len(s[1]) for s in segments) <= 64
        out = {}
        for slot, ids, rowmask, mode, limit in segments:
            T, nk = len(ids), self.slot_keys[slot]
            tree = np.array([[(int(rowmask[i]) >> j) & 1 for j in range(T)] for i in range(T)], dtype=np.int64)
            full = torch.cat([torch.ones((T, nk), dtype=torch.long), torch.from_numpy(tree)], 1)
            logits, past = self.model.forward(torch.tensor([int(x) for x in ids]), full, self.past[slot])
            am = [int(x) for x in torch.argmax(logits.float(), -1)]
            if mode == 1:
                toks, rows = [am[-1]], list(range(T))
            else:
                toks, rows = lo.accept_scan_limited([int(x) for x in ids], tree, am, max(1, min(16, int(limit))))
            idx = torch.tensor(list(range(nk)) + [nk + r for r in rows], dtype=torch.long)
            self.past[slot] = [(k[:, idx], v[:, idx]) for k, v in past]
            self.slot_keys[slot] = nk + len(rows)
            out[slot] = toks
        return out

    def bprefill_many(self, prompts, eager=False):
        return {s: self.bstep([(s, p, [(2 << t) - 1 for t in range(len(p))], 1, 1)])[s][0] for s, p in prompts.items()}
