# -*- coding: utf-8 -*-
"""DeviceTrie — hier_get on the GPU over a mirrored snapshot of a LookaheadCache (la_cache_export +
la_trie_hier_get_dev, csrc/la_trie_dev.hip).  One wavefront per query; bit-identical to LookaheadCache.hier_get.

The host trie stays the owner of all updates (put / stream_put / squeeze); a mirror is a read-only snapshot for one
input-frequency slot `idx`.  On the bs=1 path the host query (10-40 us) is faster than any device pointer chase, so
lookahead_generation() keeps using it; this class is the batched retrieval building block (B queries in one launch).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import lib, check

_MODES = {'input': 0, 'output': 1, 'mix': 2}
_pd = C.POINTER(C.c_double)


class DeviceTrie(object):
    def __init__(self, cache, idx=0, device='cuda:0'):
        if not torch.cuda.is_available():
            raise RuntimeError('DeviceTrie needs an MI355X (no CPU fallback)')
        self.device = torch.device(device)
        n = C.c_int32()
        check(lib.la_cache_export(cache._h, int(idx), 0, None, None, None, None, None, C.byref(n)), 'export(size)')
        cap = n.value
        tok = np.zeros(cap, np.int32); fo = np.zeros(cap, np.float64); fi = np.zeros(cap, np.float64)
        cs = np.zeros(cap, np.int32); cc = np.zeros(cap, np.int32)
        check(lib.la_cache_export(cache._h, int(idx), cap, tok.ctypes.data_as(_lib.pi32), fo.ctypes.data_as(_pd),
                                  fi.ctypes.data_as(_pd), cs.ctypes.data_as(_lib.pi32), cc.ctypes.data_as(_lib.pi32),
                                  C.byref(n)), 'export')
        self.n_nodes = n.value
        up = lambda a: torch.from_numpy(a[:self.n_nodes].copy()).to(self.device)
        self.tok, self.fo, self.fi, self.cstart, self.ccount = up(tok), up(fo), up(fi), up(cs), up(cc)
        self.stop_words = [int(x) for x in cache.stop_words]

    def hier_get(self, queries, decoding_length=64, branch_length=8, min_input_size=0, min_output_size=0, mode='mix'):
        """queries: list of token lists (each <= 8 tokens).  -> list of (ids list, uint64 row masks, sizes list)."""
        assert mode in _MODES and decoding_length <= _lib.LA_TREE_MAX
        B = len(queries)
        q = np.zeros((B, 8), np.int32); nq = np.zeros(B, np.int32)
        for b, toks in enumerate(queries):
            assert len(toks) <= 8
            q[b, :len(toks)] = toks; nq[b] = len(toks)
        dq, dnq = torch.from_numpy(q).to(self.device), torch.from_numpy(nq).to(self.device)
        stop = torch.tensor(self.stop_words or [0], dtype=torch.int32, device=self.device)
        sq = torch.empty(B * self.n_nodes, dtype=torch.int32, device=self.device)
        sv = torch.empty(B * 2 * self.n_nodes, dtype=torch.float64, device=self.device)
        ids = torch.zeros(B * 64, dtype=torch.int32, device=self.device)
        rm = torch.zeros(B * 64, dtype=torch.int64, device=self.device)
        on = torch.zeros(B, dtype=torch.int32, device=self.device)
        osz = torch.zeros(B * 2, dtype=torch.int32, device=self.device)
        ons = torch.zeros(B, dtype=torch.int32, device=self.device)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr())
        check(lib.la_trie_hier_get_dev(st, p(self.tok), p(self.fo), p(self.fi), p(self.cstart), p(self.ccount), self.n_nodes,
                                       p(dq), p(dnq), B, int(decoding_length), int(branch_length), int(min_input_size),
                                       int(min_output_size), _MODES[mode], p(stop), len(self.stop_words), p(sq), p(sv),
                                       p(ids), p(rm), p(on), p(osz), p(ons)), 'trie_hier_get_dev')
        torch.cuda.synchronize(self.device)
        ids, rm, on, osz, ons = ids.cpu().numpy().reshape(B, 64), rm.cpu().numpy().view(np.uint64).reshape(B, 64), \
            on.cpu().numpy(), osz.cpu().numpy().reshape(B, 2), ons.cpu().numpy()
        return [(ids[b, :on[b]].tolist(), rm[b, :on[b]].copy(), osz[b, :ons[b]].tolist()) for b in range(B)]
