# -*- coding: utf-8 -*-
"""DeviceTrie — hier_get on the GPU over an INCREMENTAL mirror of a LookaheadCache (csrc/la_trie.cpp Mirror +
csrc/la_trie_wg.hip / la_trie_dev.hip).  One WORKGROUP per query (round 6, the default: level-synchronous expansion, radix-select
cut-offs, per-level rank / size / position passes in LDS; draft trees of up to 256 rows with multi-word row masks) or one wavefront
per query (`algo='wave'`, rounds 1-5: ordered DFS, <= 64 rows); both bit-identical to LookaheadCache.hier_get
(lookahead_cache.py:408-439).

The host trie stays the owner of all updates (put / stream_put / reset_input_freqs / squeeze); it logs every word an update
changes in the device layout.  sync() ships that log — a few hundred bytes per verify step — as one pinned H2D copy plus one
patch kernel on the query stream; only fresh / load / squeeze (request boundaries) and arena growth past the device capacity
cost a full image upload.  Input-frequency slots (`idxs`, one per batch index) are mirrored as fi planes, so one launch serves
every sequence of a batch step with its own input frequencies.

Product use: pretrained_model_batch.lookahead_generation(decoding_kwargs={'device_trie': True}) retrieves the drafts of all
active samples with one launch here instead of one host query per sample.

Device-side UPDATES (put_vocab=...): stream_put_dev() applies LookaheadCache.stream_put(final=False, mode='output')
(lookahead_cache.py:369-406) to the device image on the device, from tokens that are already in HBM (a verify step's accepted
tokens); replay() then repeats the same puts on the host trie, whose mirror grows by the same rule in the same order, and drops
the words it logged — the two images stay identical word for word and those updates never cross PCIe.
"""
import ctypes as C
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import lib, check

_MODES = {'input': 0, 'output': 1, 'mix': 2}
_pd = C.POINTER(C.c_double)


class DeviceTrie(object):
    def __init__(self, cache, idx=None, device='cuda:0', idxs=None, max_queries=64, put_vocab=None, cap_slack=4096, algo='wg',
                 max_rows=64):
        """put_vocab: enable device-side updates; = the model's vocabulary size (length of the token -> tree root table).
        cap_slack: free records behind a fresh image (the image is re-allocated 1.5x larger when an update passes it).
        algo: 'wg' (one workgroup per query, trees up to 256 rows) | 'wave' (one wavefront per query, <= 64 rows).
        max_rows: rows per query of the result block (64: the layout la_llama_mstep_trie chains; grows to 256 on the first wider query)."""
        assert algo in ('wg', 'wave')
        self.algo = algo
        self.wg_limits = (0, 0, 0)       # la_trie_query.lds_level_cap / lds_cand_cap / one_wave_cap (0 = the library's; tests shrink them)
        self.rows = 64 if int(max_rows) <= 64 else _lib.LA_TREE_WIDE_MAX
        if not torch.cuda.is_available():
            raise RuntimeError('DeviceTrie needs an MI355X (no CPU fallback)')
        self.device = torch.device(device)
        self.cache = cache
        self.idxs = [int(i) for i in (idxs if idxs is not None else [0 if idx is None else idx])]
        self.plane = {v: k for k, v in enumerate(self.idxs)}
        # la_cache_mirror_enable REPLACES the cache's mirror: an older DeviceTrie on the same cache would apply its next patch
        # to the wrong image.  One live owner per cache: the previous one is revoked and raises on its next use.
        prev = getattr(cache, '_mirror_owner', None)
        if prev is not None and prev() is not None:
            prev()._revoked = True
        self._revoked = False
        cache._mirror_owner = weakref.ref(self)
        self._h2d_done = torch.cuda.Event()
        self._h2d_pending = False
        self.put_vocab = int(put_vocab) if put_vocab else 0
        self.cap_slack = int(cap_slack)
        if self.put_vocab:
            n_idx = max(self.idxs) + 1
            dev = self.device
            self.n_idx = n_idx
            self.obuf = torch.zeros(n_idx * _lib.LA_TRIE_OBUF, dtype=torch.int32, device=dev)
            self.olen = torch.zeros(n_idx, dtype=torch.int32, device=dev)
            self._h_obuf = torch.zeros(n_idx * (_lib.LA_TRIE_OBUF + 1), dtype=torch.int32).pin_memory()
            self.meta = torch.zeros(4, dtype=torch.int32, device=dev)
            self._h_meta = torch.zeros(4, dtype=torch.int32).pin_memory()
            self.root_of = torch.full((self.put_vocab,), -1, dtype=torch.int32, device=dev)
            self._items = torch.zeros(64 * _lib.LA_MOUT_TOKS * 2, dtype=torch.int32, device=dev)
            self._put_idx = torch.zeros(64, dtype=torch.int32, device=dev)
            self._h_put_idx = torch.zeros(64, dtype=torch.int32).pin_memory()
            self._put_idx_list = None
            self._eos_list = self._stopw_list = None
            self.stats_put = {'calls': 0, 'replays': 0}
        self._unreplayed = 0
        arr = np.asarray(self.idxs, dtype=np.int32)
        check(lib.la_cache_mirror_enable(cache._h, arr.ctypes.data_as(_lib.pi32), len(self.idxs)), 'mirror_enable')
        self.cap = 0
        self.n_records = 0
        self.stats = {'full_uploads': 0, 'patches': 0, 'patch_words': 0}
        self._qcap = 0
        self._alloc_queries(max_queries)
        self._patch_cap = 0
        self.sync()

    # ---- device image ------------------------------------------------------------------------------------------------
    def _alloc_image(self, n):
        self.cap = max(int(n * 1.5) + self.cap_slack, 64)
        P = max(len(self.idxs), 1)
        dev = self.device
        self.tok = torch.empty(self.cap, dtype=torch.int32, device=dev)
        self.cstart = torch.empty(self.cap, dtype=torch.int32, device=dev)
        self.ccount = torch.empty(self.cap, dtype=torch.int32, device=dev)
        self.fo = torch.empty(self.cap, dtype=torch.float64, device=dev)
        self.fi = torch.zeros(P * self.cap, dtype=torch.float64, device=dev)
        self._h_tok = torch.empty(self.cap, dtype=torch.int32).pin_memory()
        self._h_cstart = torch.empty(self.cap, dtype=torch.int32).pin_memory()
        self._h_ccount = torch.empty(self.cap, dtype=torch.int32).pin_memory()
        self._h_fo = torch.empty(self.cap, dtype=torch.float64).pin_memory()
        self._h_fi = torch.zeros(P * self.cap, dtype=torch.float64).pin_memory()
        self._scratch = None
        if self.put_vocab:
            self.ccap = torch.zeros(self.cap, dtype=torch.int32, device=dev)
            self._h_ccap = torch.zeros(self.cap, dtype=torch.int32).pin_memory()

    def _alloc_queries(self, B, rows=None):
        rows = self.rows if rows is None else max(self.rows, 64 if rows <= 64 else _lib.LA_TREE_WIDE_MAX)
        if B <= self._qcap and rows == self.rows:
            return
        B = max(B, self._qcap)
        self._qcap, self.rows = B, rows
        R, W = rows, rows // 64
        self.mask_words = W
        dev = self.device
        self._hq = torch.zeros(B * 8 + 3 * B, dtype=torch.int32).pin_memory()      # queries [B][8], nq [B], plane [B], bl [B]
        self._dq = torch.zeros(B * 8 + 3 * B, dtype=torch.int32, device=dev)
        # ONE result block [row masks (8-byte aligned) | ids | n | sizes | nsizes]: hier_get reads it back with a single D2H copy
        # (round 2: five synchronous .cpu() calls, ~130 us of the 730 us a bs=1 query cost — profiles/r03_trie_device_profile.txt).
        # R rows per query, W = R / 64 mask words per row (R = 64: the layout the chained verify step reads, la_llama_mstep_trie)
        nm = 2 * R * W
        words = B * (nm + R + 1 + 2 + 1)
        self._out = torch.zeros(words, dtype=torch.int32, device=dev)
        self._h_out = torch.zeros(words, dtype=torch.int32).pin_memory()
        self.out_rm = self._out[:B * nm].view(torch.int64)
        self.out_ids = self._out[B * nm:B * (nm + R)]
        self.out_n = self._out[B * (nm + R):B * (nm + R + 1)]
        self.out_sizes = self._out[B * (nm + R + 1):B * (nm + R + 3)]
        self.out_nsizes = self._out[B * (nm + R + 3):B * (nm + R + 4)]
        self._scratch = None

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check_owner(self):
        if self._revoked:
            raise RuntimeError('this DeviceTrie was superseded by a newer mirror of the same LookaheadCache')

    def _staging_free(self):
        """The pinned staging buffers are rewritten by the host on every call: wait until the H2D copies queued from them by
        the previous call have been executed (an event behind the last copy; microseconds, normally already signalled)."""
        if self._h2d_pending:
            self._h2d_done.synchronize()
            self._h2d_pending = False

    def _staging_queued(self):
        self._h2d_done.record(torch.cuda.current_stream(self.device))
        self._h2d_pending = True

    def sync(self):
        """Bring the device image up to date with the host trie (patch, or full image when due).  Enqueued on the current
        stream; returns the kind of sync that happened."""
        self._check_owner()
        if getattr(self, '_unreplayed', 0):
            # the device applied updates the host trie has not repeated yet: a host image / patch computed now would number its
            # records differently from the device's
            raise RuntimeError('DeviceTrie.sync(): replay() the updates queued with stream_put_dev() first')
        self._staging_free()
        n, full, ni, nd = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        check(lib.la_cache_mirror_state(self.cache._h, C.byref(n), C.byref(full), C.byref(ni), C.byref(nd)), 'mirror_state')
        self.stop_words = [int(x) for x in self.cache.stop_words]
        if full.value or n.value > self.cap:
            if n.value > self.cap:
                self._alloc_image(n.value)
            P = len(self.idxs)
            check(lib.la_cache_mirror_image(self.cache._h, self.cap, C.cast(self._h_tok.data_ptr(), _lib.pi32),
                                            C.cast(self._h_fo.data_ptr(), _pd), C.cast(self._h_fi.data_ptr(), _pd),
                                            C.cast(self._h_cstart.data_ptr(), _lib.pi32),
                                            C.cast(self._h_ccount.data_ptr(), _lib.pi32)), 'mirror_image')
            k = n.value
            self.tok[:k].copy_(self._h_tok[:k], non_blocking=True)
            self.cstart[:k].copy_(self._h_cstart[:k], non_blocking=True)
            self.ccount[:k].copy_(self._h_ccount[:k], non_blocking=True)
            self.fo[:k].copy_(self._h_fo[:k], non_blocking=True)
            for p in range(P):
                self.fi[p * self.cap:p * self.cap + k].copy_(self._h_fi[p * self.cap:p * self.cap + k], non_blocking=True)
            self.n_records = k
            self.stats['full_uploads'] += 1
            if self.put_vocab:
                check(lib.la_cache_mirror_ccap(self.cache._h, self.cap, C.cast(self._h_ccap.data_ptr(), _lib.pi32)), 'mirror_ccap')
                self.ccap[:k].copy_(self._h_ccap[:k], non_blocking=True)
                self._after_host_update(k, reset=True)
            self._staging_queued()
            return 'full'
        self.n_records = n.value
        if ni.value == 0 and nd.value == 0:
            return 'clean'
        need = 3 * ni.value + 2 * nd.value
        if need > self._patch_cap or nd.value > getattr(self, '_patch_dcap', 0):
            self._patch_cap = max(2 * need, 4096)
            self._patch_dcap = max(2 * nd.value, 2048)
            self._h_pi = torch.zeros(self._patch_cap, dtype=torch.int32).pin_memory()
            self._d_pi = torch.zeros(self._patch_cap, dtype=torch.int32, device=self.device)
            self._h_pd = torch.zeros(self._patch_dcap, dtype=torch.float64).pin_memory()
            self._d_pd = torch.zeros(self._patch_dcap, dtype=torch.float64, device=self.device)
        ip = C.cast(self._h_pi.data_ptr(), _lib.pi32)
        dk = C.cast(self._h_pi.data_ptr() + 4 * 3 * ni.value, _lib.pi32)
        check(lib.la_cache_mirror_patch(self.cache._h, ip, dk, C.cast(self._h_pd.data_ptr(), _pd)), 'mirror_patch')
        self._d_pi[:need].copy_(self._h_pi[:need], non_blocking=True)
        if nd.value:
            self._d_pd[:nd.value].copy_(self._h_pd[:nd.value], non_blocking=True)
        check(lib.la_trie_patch_dev(self._stream(), self.tok.data_ptr(), self.fo.data_ptr(), self.fi.data_ptr(), self.cap,
                                    self.cstart.data_ptr(), self.ccount.data_ptr(), self.ccap.data_ptr() if self.put_vocab else None,
                                    self._d_pi.data_ptr(), ni.value,
                                    self._d_pi.data_ptr() + 4 * 3 * ni.value, self._d_pd.data_ptr(), nd.value), 'trie_patch_dev')
        if self.put_vocab:
            self._after_host_update(n.value, reset=False)
        self._staging_queued()
        self.stats['patches'] += 1
        self.stats['patch_words'] += ni.value + nd.value
        return 'patch'

    # ---- device-side updates ---------------------------------------------------------------------------------------------
    def _image(self):
        img = _lib.TrieImageC()
        img.tok, img.fo, img.fi, img.fi_stride = self.tok.data_ptr(), self.fo.data_ptr(), self.fi.data_ptr(), self.cap
        img.n_planes = len(self.idxs)
        img.cstart, img.ccount, img.ccap = self.cstart.data_ptr(), self.ccount.data_ptr(), self.ccap.data_ptr()
        img.meta, img.cap = self.meta.data_ptr(), self.cap
        img.root_of, img.n_root_of = self.root_of.data_ptr(), self.put_vocab
        return img

    def _after_host_update(self, n_records, reset):
        """A host image / patch just went to the device: the record count the device inserts continue from, and the token ->
        root table (a patch may have added trees or moved the root block)."""
        self._h_meta[0] = n_records
        if reset:
            self._h_meta[1:] = 0
            self.meta.copy_(self._h_meta, non_blocking=True)
        else:
            self.meta[:1].copy_(self._h_meta[:1], non_blocking=True)
        img = self._image()
        check(lib.la_trie_root_index_dev(self._stream(), C.byref(img), int(n_records)), 'trie_root_index_dev')

    def load_stream_buffers(self):
        """Upload the host's hold-back buffers (_output_ids[idx], lookahead_cache.py:369) of the mirrored slots: from here on the
        device rolls them itself (stream_put_dev) and the host's copies follow through replay()."""
        assert self.put_vocab, 'DeviceTrie(put_vocab=...) enables device-side updates'
        self._check_owner()
        self._staging_free()
        W = _lib.LA_TRIE_OBUF
        h = self._h_obuf.numpy()
        h[:] = 0
        n = C.c_int32()
        for idx in self.idxs:
            check(lib.la_cache_stream_buffer(self.cache._h, int(idx), W, h[idx * W:].ctypes.data_as(_lib.pi32), C.byref(n)),
                  'stream_buffer')
            h[self.n_idx * W + idx] = n.value
        self.obuf.copy_(self._h_obuf[:self.n_idx * W], non_blocking=True)
        self.olen.copy_(self._h_obuf[self.n_idx * W:], non_blocking=True)
        self._staging_queued()

    def _small_dev_list(self, name, values):
        """device copy of a short int list (eos ids, stop words), re-uploaded only when it changes"""
        key = [int(v) for v in values]
        if getattr(self, name + '_list', None) != key or getattr(self, name + '_dev', None) is None:
            setattr(self, name + '_list', key)
            setattr(self, name + '_dev', torch.tensor(key or [0], dtype=torch.int32, device=self.device))
        return getattr(self, name + '_dev'), len(key)

    def stream_put_dev(self, src_tok_ptr, src_stride, src_cnt_ptr, idxs, branch_length):
        """Queue the device-side stream_put of len(idxs) sequences on the current stream: put k appends the src_cnt_ptr[k] int32
        tokens at src_tok_ptr + 4 * k * src_stride (device pointers, e.g. a verify step's output block) to the hold-back buffer of
        slot idxs[k].  The host must replay() the same puts (same order) before its next own update of the trie."""
        assert self.put_vocab, 'DeviceTrie(put_vocab=...) enables device-side updates'
        self._check_owner()
        idxs = [int(i) for i in idxs]
        assert len(set(idxs)) == len(idxs) <= 64 and all(0 <= i < self.n_idx for i in idxs)
        if idxs != self._put_idx_list:
            self._staging_free()
            self._h_put_idx[:len(idxs)] = torch.tensor(idxs, dtype=torch.int32)
            self._put_idx.copy_(self._h_put_idx, non_blocking=True)
            self._staging_queued()
            self._put_idx_list = idxs
        eos = [e for e in (self.cache.eos_ids or []) if e is not None]
        eos_dev, n_eos = self._small_dev_list('_eos', eos)
        stop_dev, n_stop = self._small_dev_list('_stopw', sorted(self.cache.stop_words or []))
        img = self._image()
        check(lib.la_trie_stream_put_dev(self._stream(), C.byref(img), self.obuf.data_ptr(), self.olen.data_ptr(),
                                         C.c_void_p(src_tok_ptr), int(src_stride), C.c_void_p(src_cnt_ptr), self._put_idx.data_ptr(),
                                         len(idxs), int(branch_length), stop_dev.data_ptr(), n_stop, eos_dev.data_ptr(), n_eos,
                                         self._items.data_ptr()), 'trie_stream_put_dev')
        self.stats_put['calls'] += 1
        self._unreplayed += 1

    def replay(self, puts, branch_length, calls=1):
        """Repeat on the host trie what `calls` consecutive stream_put_dev() calls did on the device: puts = [(idx, tokens)] in their order
        (calls are replayed in the order they were queued; the next call may already be queued on the device — round 4: the loops
        replay step N while the GPU runs step N + 1).  The words the host logs for them are already in the device image and are dropped.  -> False when the host's image left the
        device's behind (capacity passed, or a squeeze): the next sync() uploads a full image."""
        n_pending = C.c_int32()
        n, full, ni, nd = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        check(lib.la_cache_mirror_state(self.cache._h, C.byref(n), C.byref(full), C.byref(ni), C.byref(nd)), 'mirror_state')
        assert not full.value and ni.value == 0 and nd.value == 0, 'replay() needs a synced mirror (call sync() after host updates)'
        for idx, toks in puts:
            self.cache.stream_put([int(t) for t in toks if t != -1], branch_length=branch_length, final=False, mode='output', idx=int(idx))
        self.stats_put['replays'] += 1
        self._unreplayed = max(0, self._unreplayed - int(calls))       # `calls` stream_put_dev calls are covered by these puts; newer ones may already be queued
        rc = lib.la_cache_mirror_discard(self.cache._h, C.byref(n_pending))
        if rc != 0:
            return False
        self.n_records = n_pending.value
        return n_pending.value <= self.cap

    # ---- queries -------------------------------------------------------------------------------------------------------
    def hier_get_dev(self, queries, idxs=None, branch_lengths=None, decoding_length=64, branch_length=8, min_input_size=0,
                     min_output_size=0, mode='mix', sync=True):
        """One launch for all queries; results stay on the device: out_ids int32[B][64], out_rm uint64[B][64], out_n int32[B]
        (views of buffers reused by the next call).  queries: token lists (<= 8 tokens each); idxs: the input slot of each
        query (default: the first mirrored slot); branch_lengths: per-query branch length (default: branch_length)."""
        assert mode in _MODES and decoding_length <= (_lib.LA_TREE_WIDE_MAX if self.algo == 'wg' else _lib.LA_TREE_MAX), \
            'the device retrieval emits <= 256 rows per query (one workgroup per query) / <= 64 (one wavefront per query)'
        B = len(queries)
        self._check_owner()
        self._alloc_queries(B, int(decoding_length))
        if sync:
            self.sync()
        self._staging_free()
        h = self._hq.numpy()
        q = h[:B * 8].reshape(B, 8)
        q[:] = 0
        for b, toks in enumerate(queries):
            assert len(toks) <= 8
            q[b, :len(toks)] = toks
            h[B * 8 + b] = len(toks)
            h[B * 9 + b] = self.plane[int(idxs[b])] if idxs is not None else 0
            h[B * 10 + b] = int(branch_lengths[b]) if branch_lengths is not None else int(branch_length)
        self._dq[:B * 11].copy_(self._hq[:B * 11], non_blocking=True)
        self._staging_queued()
        if self._scratch is None or self._scratch[0].numel() < B * 16 * self.cap:
            self._scratch = (torch.empty(B * 16 * self.cap, dtype=torch.int32, device=self.device),
                             torch.empty(B * 3 * self.cap, dtype=torch.float64, device=self.device))
        stop = getattr(self, '_stop_dev', None)
        if stop is None or self._stop_list != self.stop_words:
            self._stop_list = list(self.stop_words)
            self._stop_dev = stop = torch.tensor(self.stop_words or [0], dtype=torch.int32, device=self.device)
        base = self._dq.data_ptr()
        if self.algo == 'wg':
            q = _lib.TrieQueryC()
            q.tok, q.fo, q.fi, q.fi_stride = self.tok.data_ptr(), self.fo.data_ptr(), self.fi.data_ptr(), self.cap
            q.cstart, q.ccount, q.n_records = self.cstart.data_ptr(), self.ccount.data_ptr(), self.cap
            q.root_of, q.n_root_of = (self.root_of.data_ptr(), self.put_vocab) if self.put_vocab else (None, 0)
            q.queries, q.nq, q.plane, q.branch_lengths, q.B = base, base + 4 * B * 8, base + 4 * B * 9, base + 4 * B * 10, B
            q.decoding_length, q.branch_length, q.min_in, q.min_out = int(decoding_length), int(branch_length), int(min_input_size), int(min_output_size)
            q.mode, q.stop, q.n_stop = _MODES[mode], stop.data_ptr(), len(self.stop_words)
            q.scratch_i, q.scratch_v = self._scratch[0].data_ptr(), self._scratch[1].data_ptr()
            q.out_ids, q.out_rowmask, q.row_stride, q.mask_words = self.out_ids.data_ptr(), self.out_rm.data_ptr(), self.rows, self.mask_words
            q.out_n, q.out_sizes, q.out_nsizes = self.out_n.data_ptr(), self.out_sizes.data_ptr(), self.out_nsizes.data_ptr()
            q.lds_level_cap, q.lds_cand_cap, q.one_wave_cap = self.wg_limits
            check(lib.la_trie_hier_get_wg(self._stream(), C.byref(q)), 'trie_hier_get_wg')
            return B
        assert self.rows == 64, 'the one-wavefront kernel writes 64-row result blocks'
        check(lib.la_trie_hier_get_dev2(self._stream(), self.tok.data_ptr(), self.fo.data_ptr(), self.fi.data_ptr(), self.cap,
                                        self.cstart.data_ptr(), self.ccount.data_ptr(), self.cap, base, base + 4 * B * 8,
                                        base + 4 * B * 9, base + 4 * B * 10, B, int(decoding_length), int(branch_length),
                                        int(min_input_size), int(min_output_size), _MODES[mode], stop.data_ptr(),
                                        len(self.stop_words), self._scratch[0].data_ptr(), self._scratch[1].data_ptr(),
                                        self.out_ids.data_ptr(), self.out_rm.data_ptr(), self.out_n.data_ptr(),
                                        self.out_sizes.data_ptr(), self.out_nsizes.data_ptr()), 'trie_hier_get_dev2')
        return B

    def one_get_dev(self, queries, idxs=None, branch_lengths=None, decoding_length=64, branch_length=8, mode='mix', sync=True):
        """LookaheadCache.one_get (lookahead_cache.py:490-517) for all queries in one launch: per query the single most frequent
        chain; results in the same device buffers as hier_get_dev (row masks lower-triangular), so a chained verify step
        (LlamaVerifyEngine.mstep_trie) can take them as they are."""
        assert mode in _MODES and int(branch_length) + 1 <= _lib.LA_TREE_MAX
        assert self.rows == 64, 'one_get_dev writes 64-row result blocks (a DeviceTrie that served wide hier_get queries keeps 256-row blocks)'
        B = len(queries)
        self._check_owner()
        self._alloc_queries(B)
        if sync:
            self.sync()
        self._staging_free()
        h = self._hq.numpy()
        q = h[:B * 8].reshape(B, 8)
        q[:] = 0
        for b, toks in enumerate(queries):
            assert len(toks) <= 8
            q[b, :len(toks)] = toks
            h[B * 8 + b] = len(toks)
            h[B * 9 + b] = self.plane[int(idxs[b])] if idxs is not None else 0
            h[B * 10 + b] = int(branch_lengths[b]) if branch_lengths is not None else int(branch_length)
        self._dq[:B * 11].copy_(self._hq[:B * 11], non_blocking=True)
        self._staging_queued()
        stop = getattr(self, '_stop_dev', None)
        if stop is None or self._stop_list != self.stop_words:
            self._stop_list = list(self.stop_words)
            self._stop_dev = stop = torch.tensor(self.stop_words or [0], dtype=torch.int32, device=self.device)
        base = self._dq.data_ptr()
        check(lib.la_trie_one_get_dev2(self._stream(), self.tok.data_ptr(), self.fo.data_ptr(), self.fi.data_ptr(), self.cap,
                                       self.cstart.data_ptr(), self.ccount.data_ptr(), self.cap, base, base + 4 * B * 8,
                                       base + 4 * B * 9, base + 4 * B * 10, B, int(decoding_length), int(branch_length),
                                       _MODES[mode], stop.data_ptr(), len(self.stop_words), self.out_ids.data_ptr(),
                                       self.out_rm.data_ptr(), self.out_n.data_ptr(), self.out_sizes.data_ptr(),
                                       self.out_nsizes.data_ptr()), 'trie_one_get_dev2')
        return B

    def one_get(self, queries, decoding_length=64, branch_length=8, mode='mix', idxs=None, branch_lengths=None):
        """-> list of (ids list, uint64 row masks, sizes list), like LookaheadCache.one_get per query (masks packed)."""
        B = self.one_get_dev(queries, idxs=idxs, branch_lengths=branch_lengths, decoding_length=decoding_length,
                             branch_length=branch_length, mode=mode)
        return self._read_results(B)

    def _read_results(self, B):
        self._h_out.copy_(self._out, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        Q, R, W = self._qcap, self.rows, self.mask_words
        nm = 2 * R * W
        h = self._h_out.numpy()
        rm = h[:Q * nm].view(np.uint64).reshape(Q, R, W)
        ids = h[Q * nm:Q * (nm + R)].reshape(Q, R)
        on = h[Q * (nm + R):Q * (nm + R + 1)]
        osz = h[Q * (nm + R + 1):Q * (nm + R + 3)].reshape(Q, 2)
        ons = h[Q * (nm + R + 3):Q * (nm + R + 4)]
        if (ons[:B] < 0).any():
            raise RuntimeError('device trie: a queried subtree is deeper than the 128 levels the workgroup kernel follows')
        return [(ids[b, :on[b]].tolist(), (rm[b, :on[b], 0] if W == 1 else rm[b, :on[b]]).copy(), osz[b, :ons[b]].tolist()) for b in range(B)]

    def hier_get(self, queries, decoding_length=64, branch_length=8, min_input_size=0, min_output_size=0, mode='mix', idxs=None,
                 branch_lengths=None):
        """-> list of (ids list, uint64 row masks, sizes list), like LookaheadCache.hier_get per query; the masks are uint64[T] while the
        result block holds 64 rows per query and uint64[T][4] (word w = tree columns 64 w ..) once a query asked for more."""
        B = self.hier_get_dev(queries, idxs=idxs, branch_lengths=branch_lengths, decoding_length=decoding_length,
                              branch_length=branch_length, min_input_size=min_input_size, min_output_size=min_output_size, mode=mode)
        return self._read_results(B)
