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

    def prefill(self, prompt_ids):
        tok = None
        for s in range(0, len(prompt_ids), 64):
            blk = prompt_ids[s:s + 64]
            rows = [(2 << t) - 1 for t in range(len(blk))]
            logits, past, _ = self._forward(blk, rows)
            self.past = past
            self.n_keys += len(blk)
            tok = int(torch.argmax(logits[-1].float()))
        return tok

    def step(self, ids, rowmask, mode=0, eager=False):
        logits, past, tree = self._forward(ids, rowmask)
        am = [int(x) for x in torch.argmax(logits.float(), -1)]
        self.last_argmax = am
        toks, rows = lo.accept_scan([int(x) for x in ids], tree, am)
        keep = list(range(self.n_keys)) + [self.n_keys + r for r in rows]
        idx = torch.tensor(keep, dtype=torch.long)
        self.past = [(k[:, idx], v[:, idx]) for k, v in past]
        self.n_keys += len(rows)
        return toks, len(rows)
