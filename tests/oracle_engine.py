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
        rm = np.asarray(mask_rows, dtype=np.uint64)
        if rm.ndim == 1:
            rm = rm[:, None]
        words = [sum(int(rm[i, w]) << (64 * w) for w in range(rm.shape[1])) for i in range(T)]     # wide trees: several words per row
        tree = np.array([[(words[i] >> j) & 1 for j in range(T)] for i in range(T)], dtype=np.int64)
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

    # wide-tree surface of LlamaVerifyEngine (max_blocks > 0): the oracle forward has no block structure, so these are aliases
    max_blocks = 4

    def mprefill(self, slot, prompt_ids, eager=False):
        assert slot == 0
        return self.prefill(prompt_ids)

    def mlogits(self):
        return self._logits

    def tstep(self, ids, rowmask, slot=0, mode=0, limit=None, eager=False):
        assert len(ids) <= 256
        if mode == 2:
            self.verify_only(ids, rowmask)
            return [], 0
        return self.step(ids, rowmask)

    def tcommit(self, rows, n_rows):
        self.commit(rows)

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

    # asynchronous surface (LlamaVerifyEngine.step_async / step_finish): the stand-in computes at once, the result waits
    def step_async(self, ids, rowmask, mode=0, eager=False):
        self._async = self.step(ids, rowmask, mode)

    def step_finish(self):
        out, self._async = self._async, None
        return out


class OracleBatchEngine(object):
    """TEST-ONLY stand-in for LlamaVerifyEngine(n_slots=N[, max_blocks=M]): the cursor-batch surface (reset_slot / bprefill /
    bprefill_many / bstep / bcommit) and the multi-block surface (mprefill / mprefill_many / mstep / mcommit) on the oracle
    forward, one KV list per slot.  mode 2 = forward only, the host decides the commit (sequential accept path)."""
    device = torch.device('cpu')

    def __init__(self, shape, state_dict, max_length=512, n_slots=4, max_blocks=0):
        self.shape = shape
        self.model = lo.OracleLlama(shape, state_dict)
        self.max_keys = ((max_length + 65 + 31) // 32) * 32
        self.n_slots = n_slots
        self.max_blocks = max_blocks
        self.reset_slot(-1)

    def reset_slot(self, slot):
        if slot < 0:
            self.past = [None] * self.n_slots
            self.slot_keys = [0] * self.n_slots
        else:
            self.past[slot] = None
            self.slot_keys[slot] = 0

    def _segment(self, slot, ids, rowmask, mode, limit, tok_cap=16):
        T, nk = len(ids), self.slot_keys[slot]
        if np.ndim(rowmask) == 2:              # multi-word row masks of a wide tree: word w = tree columns 64 w .. 64 w + 63
            rowmask = [sum(int(rowmask[i][w]) << (64 * w) for w in range(len(rowmask[i]))) for i in range(T)]
        tree = np.array([[(int(rowmask[i]) >> j) & 1 for j in range(T)] for i in range(T)], dtype=np.int64)
        full = torch.cat([torch.ones((T, nk), dtype=torch.long), torch.from_numpy(tree)], 1)
        logits, past = self.model.forward(torch.tensor([int(x) for x in ids]), full, self.past[slot])
        am = [int(x) for x in torch.argmax(logits.float(), -1)]
        if mode == 2:
            self._pending[slot] = past
            return [], logits
        if mode == 1:
            toks, rows = [am[-1]], list(range(T))
        else:
            toks, rows = lo.accept_scan_limited([int(x) for x in ids], tree, am, max(1, min(tok_cap, int(limit))))
        self._keep(slot, past, rows)
        return toks, logits

    def _keep(self, slot, past, rows):
        nk = self.slot_keys[slot]
        idx = torch.tensor(list(range(nk)) + [nk + int(r) for r in rows], dtype=torch.long)
        self.past[slot] = [(k[:, idx], v[:, idx]) for k, v in past]
        self.slot_keys[slot] = nk + len(rows)

    def bstep(self, segments, eager=False):
        assert sum(len(s[1]) for s in segments) <= 64
        out, self._pending, self._first, blocks, row = {}, {}, {}, [], 0
        for slot, ids, rowmask, mode, limit in segments:
            out[slot], lg = self._segment(slot, ids, rowmask, mode, limit)
            self._first[slot] = row
            row += len(ids)
            blocks.append(lg)
        self._logits = torch.cat(blocks, 0)
        return out

    def bstep_rows(self):
        return dict(self._first)

    def logits(self):
        return self._logits

    def bcommit(self, kept):
        for slot, rows in kept.items():
            self._keep(slot, self._pending[slot], rows)

    def bprefill(self, slot, prompt_ids, eager=False):
        tok = None
        for s in range(0, len(prompt_ids), 64):
            blk = prompt_ids[s:s + 64]
            tok = self.bstep([(slot, blk, [(2 << t) - 1 for t in range(len(blk))], 1, 1)])[slot][0]
        return tok

    def bprefill_many(self, prompts, eager=False):
        return {s: self.bprefill(s, p) for s, p in prompts.items()}

    # ---- multi-block surface: every block a full tree of its own -----------------------------------------------
    def mstep(self, blocks, eager=False):
        assert self.max_blocks and len(blocks) <= self.max_blocks
        out, self._pending, lgs, self._mslots = [], {}, [], [b[0] for b in blocks]
        for slot, ids, rowmask, mode, limit in blocks:
            toks, lg = self._segment(slot, ids, rowmask, mode, limit)
            out.append(toks)
            lgs.append(torch.cat([lg, torch.zeros((64 - lg.shape[0], lg.shape[1]), dtype=lg.dtype)], 0))
        self._mlogits = torch.cat(lgs, 0)
        return out

    def mstep_async(self, blocks, eager=False):
        self._masync = self.mstep(blocks)

    def mstep_finish(self):
        out, self._masync = self._masync, None
        return out

    def mstep_trees(self, trees, eager=False):
        """several sequences' trees of up to 256 rows, ceil(T / 64) blocks each, one pass (LlamaVerifyEngine.mstep_trees)"""
        assert self.max_blocks and sum((len(t[1]) + 63) // 64 for t in trees) <= self.max_blocks
        out, self._pending, lgs, self._mslots = [], {}, [], [t[0] for t in trees]
        for slot, ids, rowmask, mode, limit in trees:
            toks, lg = self._segment(slot, ids, rowmask, mode, limit, tok_cap=40)
            out.append(toks)
            pad = (-lg.shape[0]) % 64
            lgs.append(torch.cat([lg, torch.zeros((pad, lg.shape[1]), dtype=lg.dtype)], 0))
        self._mlogits = torch.cat(lgs, 0)
        return out

    def mcommit_trees(self, kept, n_rows):
        for slot, rows in zip(self._mslots, kept):
            self._keep(slot, self._pending[slot], rows)

    def mlogits(self):
        return self._mlogits

    def mcommit(self, kept):
        for slot, rows in zip(self._mslots, kept):
            self._keep(slot, self._pending[slot], rows)

    def mprefill(self, slot, prompt_ids, eager=False):
        tok, per = None, 64 * self.max_blocks
        for s in range(0, len(prompt_ids), per):
            piece = prompt_ids[s:s + per]
            blocks = [(slot, piece[i:i + 64], [(2 << t) - 1 for t in range(len(piece[i:i + 64]))], 1, 1)
                      for i in range(0, len(piece), 64)]
            tok = self.mstep(blocks)[-1][0]
        return tok

    def mprefill_many(self, prompts, eager=False):
        return {s: self.mprefill(s, p) for s, p in prompts.items()}
