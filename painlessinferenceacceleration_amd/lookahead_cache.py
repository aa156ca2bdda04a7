# -*- coding: utf-8 -*-
"""LookaheadCache — the reference's trie-cache API on top of the native trie (la_cache_* C ABI).

Mirrors class LookaheadCache of lookahead/lookahead/common/lookahead_cache.py:336-587 (same
constructor, method names, argument meaning, return types and assertion behaviour) so that the
reference's model wrappers, examples and benchmarks can use it unchanged.  All trie state lives
in liblookahead_hip.so (csrc/la_trie.cpp); this file only converts arguments and results.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, check

_MODES = {'input': _lib.LA_MODE_INPUT, 'output': _lib.LA_MODE_OUTPUT, 'mix': _lib.LA_MODE_MIX}


def _as_i32(token_ids):
    arr = np.ascontiguousarray(np.asarray(token_ids, dtype=np.int32).reshape(-1))
    return arr, arr.ctypes.data_as(_lib.pi32), int(arr.shape[0])


class _MemView(object):
    """Read-only stand-in for the reference's `cache.mem` dict (callers use len() / `in` / truthiness)."""

    def __init__(self, cache):
        self._cache = cache

    def __len__(self):
        return self._cache.stats()['n_trees']

    def __contains__(self, token_id):
        n = C.c_int64()
        return lib.la_cache_tree_counters(self._cache._h, int(token_id), C.byref(n), None) == 0

    def __bool__(self):
        return len(self) > 0


class LookaheadCache(object):
    def __init__(self, debug=False, eos_ids=(2,), stop_words=None, max_node=65536, max_output_node=512):
        self.debug = debug
        self._h = lib.la_cache_create(int(max_node), int(max_output_node))
        if not self._h:
            raise MemoryError('la_cache_create failed')
        self._max_node = int(max_node)
        self._max_output_node = int(max_output_node)
        self._cap = 0
        self._alloc(64)
        self.eos_ids = eos_ids if eos_ids is not None else [None]
        self.stop_words = stop_words if stop_words is not None else {}
        self.default_mask = np.ones((1, 1), dtype=np.int64)

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            lib.la_cache_destroy(h)
            self._h = None

    # ---- attributes the reference's callers mutate (pretrained_model.py:1088-1089, benchmark.py:271-272)
    @property
    def eos_ids(self):
        return self._eos_ids

    @eos_ids.setter
    def eos_ids(self, value):
        self._eos_ids = value if value is not None else [None]
        self._push_eos()

    def _push_eos(self):
        ids = [int(x) for x in self._eos_ids if x is not None]
        self._eos_pushed = tuple(ids)
        arr, p, n = _as_i32(ids)
        check(lib.la_cache_set_eos(self._h, p, n), 'set_eos')

    @property
    def stop_words(self):
        return self._stop_words

    @stop_words.setter
    def stop_words(self, value):
        self._stop_words = value if value is not None else {}
        self._push_stop_words()

    def _push_stop_words(self):
        ids = sorted(int(x) for x in self._stop_words)
        self._stop_pushed = tuple(ids)
        arr, p, n = _as_i32(ids)
        check(lib.la_cache_set_stop_words(self._h, p, n), 'set_stop_words')

    def _sync_live(self):
        """The reference reads `self.stop_words` / `self.eos_ids` live on every call (lookahead_cache.py:352, 396, 422), so
        callers may mutate the containers in place (cache.stop_words.add(x), or the dict they passed through
        decoding_kwargs).  The native trie holds a copy: re-push it whenever the live contents differ from the last push
        (a tuple compare of a handful of ids per call)."""
        if len(self._stop_words) != len(self._stop_pushed) or \
                (self._stop_pushed and tuple(sorted(int(x) for x in self._stop_words)) != self._stop_pushed):
            self._push_stop_words()
        if tuple(int(x) for x in self._eos_ids if x is not None) != self._eos_pushed:
            self._push_eos()

    @property
    def max_node(self):
        return self._max_node

    @max_node.setter
    def max_node(self, v):
        self._max_node = int(v)
        check(lib.la_cache_set_limits(self._h, self._max_node, self._max_output_node))

    @property
    def max_output_node(self):
        return self._max_output_node

    @max_output_node.setter
    def max_output_node(self, v):
        self._max_output_node = int(v)
        check(lib.la_cache_set_limits(self._h, self._max_node, self._max_output_node))

    @property
    def mem(self):
        return _MemView(self)

    @mem.setter
    def mem(self, value):
        # the reference resets with `cache.mem = {}` (== fresh()); any other assignment is unsupported
        if value:
            raise TypeError('LookaheadCache.mem can only be reset to an empty dict; use load_mem()')
        self.fresh()

    def _alloc(self, cap):
        if cap <= self._cap:
            return
        self._cap = cap
        self._ids = np.zeros(cap, dtype=np.int32)
        self._parent = np.zeros(cap, dtype=np.int32)
        self._rowmask = np.zeros(cap * ((cap + 63) // 64), dtype=np.uint64)       # W = ceil(decoding_length / 64) words per row
        self._mask = np.zeros(cap * cap, dtype=np.int64)
        self._sizes = np.zeros(2, dtype=np.int32)
        self._n = C.c_int32()
        self._nsizes = C.c_int32()
        # the ctypes views of the result buffers are built once per allocation, not once per query (a query on the bs=1 critical
        # path spent more time in these conversions than in the trie walk)
        self._p_ids, self._p_parent = self._ids.ctypes.data_as(_lib.pi32), self._parent.ctypes.data_as(_lib.pi32)
        self._p_rowmask, self._p_mask = self._rowmask.ctypes.data_as(_lib.pu64), self._mask.ctypes.data_as(_lib.pi64)
        self._p_sizes = self._sizes.ctypes.data_as(_lib.pi32)
        self._r_n, self._r_nsizes = C.byref(self._n), C.byref(self._nsizes)
        if not hasattr(self, '_q'):
            self._q = np.zeros(64, dtype=np.int32)
            self._p_q = self._q.ctypes.data_as(_lib.pi32)

    # ---- updates -------------------------------------------------------------------------------------
    def put(self, token_ids, branch_length=8, final=False, mode='output', idx=0):
        """lookahead_cache.py:349-373."""
        self._sync_live()
        assert mode in ('input', 'output')
        arr, p, n = _as_i32(token_ids)
        check(lib.la_cache_put(self._h, p, n, int(branch_length), int(bool(final)), _MODES[mode], int(idx)), 'put')

    def stream_put(self, token_ids, branch_length=8, final=False, mode='output', idx=0):
        """lookahead_cache.py:375-406."""
        self._sync_live()
        assert mode == 'output' and idx >= 0
        if isinstance(token_ids, (list, tuple)) and len(token_ids) <= 64:     # the per-step update: no array / pointer objects per call
            n = len(token_ids)
            self._q[:n] = token_ids
            p = self._p_q
        else:
            arr, p, n = _as_i32(token_ids)
        rc = lib.la_cache_stream_put(self._h, p, n, int(branch_length), int(bool(final)), int(idx))
        if rc:
            check(rc, 'stream_put')

    def stream_put_many(self, puts, branch_length=8, final=False):
        """stream_put for several sequences in one native call: puts = [(idx, token list)], applied in order (the batch loop's
        per-step update, pretrained_model_batch.py:1254-1259)."""
        self._sync_live()
        n = len(puts)
        if n == 0:
            return
        offs = np.zeros(n + 1, dtype=np.int32)
        for k, (_, toks) in enumerate(puts):
            offs[k + 1] = offs[k] + len(toks)
        flat = np.zeros(max(int(offs[-1]), 1), dtype=np.int32)
        for k, (_, toks) in enumerate(puts):
            flat[offs[k]:offs[k + 1]] = toks
        idxs = np.asarray([int(i) for i, _ in puts], dtype=np.int32)
        assert (idxs >= 0).all()
        check(lib.la_cache_stream_put_many(self._h, flat.ctypes.data_as(_lib.pi32), offs.ctypes.data_as(_lib.pi32),
                                           idxs.ctypes.data_as(_lib.pi32), n, int(branch_length), int(bool(final))), 'stream_put_many')

    # ---- retrieval -----------------------------------------------------------------------------------
    def _hier_raw(self, token_ids, decoding_length, branch_length, min_input_size, min_output_size, mode, idx,
                  want_mask):
        assert mode in ('input', 'output', 'mix')
        self._sync_live()
        self._alloc(max(int(decoding_length), 1))
        if isinstance(token_ids, (list, tuple)) and len(token_ids) <= 64:
            n = len(token_ids)
            self._q[:n] = token_ids
            p = self._p_q
        else:
            arr, p, n = _as_i32(token_ids)
        rc = lib.la_cache_hier_get(self._h, p, n, int(decoding_length), int(branch_length), int(min_input_size),
                                   int(min_output_size), _MODES[mode], int(idx), self._cap, self._p_ids, self._p_parent,
                                   self._p_rowmask, self._p_mask if want_mask else None, self._p_sizes, self._r_nsizes, self._r_n)
        if rc:
            check(rc, 'hier_get')
        return self._n.value, self._nsizes.value

    def hier_get(self, token_ids, decoding_length=64, branch_length=8, min_input_size=0, min_output_size=0,
                 mode='mix', idx=0):
        """lookahead_cache.py:408-439 -> (list ids, np.int64 [T,T] mask, list sizes)."""
        n, nsizes = self._hier_raw(token_ids, decoding_length, branch_length, min_input_size, min_output_size,
                                   mode, idx, True)
        ids = self._ids[:n].tolist()
        sizes = self._sizes[:nsizes].tolist()
        if n == 0:
            return ids, self.default_mask, sizes
        mask = self._mask[:n * n].reshape(n, n).copy()
        return ids, mask, sizes

    def hier_get_packed(self, token_ids, decoding_length=64, branch_length=8, min_input_size=0, min_output_size=0,
                        mode='mix', idx=0):
        """Device-path form of hier_get: (ids int32[T], rowmask, parent int32[T], sizes list).  decoding_length <= 64:
        rowmask uint64[T], bit j of rowmask[i] == mask[i][j].  Wide trees (decoding_length <= 256, the reference's
        decoding_length=128 / branch_length=32 setting, lookahead/README.md:100): rowmask uint64[T][W], W = ceil(decoding_length /
        64), word w of row i = columns 64 w .. 64 w + 63."""
        assert decoding_length <= _lib.LA_TREE_WIDE_MAX, 'device path handles <= 256 tree tokens'
        n, nsizes = self._hier_raw(token_ids, decoding_length, branch_length, min_input_size, min_output_size,
                                   mode, idx, False)
        W = (max(int(decoding_length), 1) + 63) // 64
        if W == 1:
            rm = self._rowmask[:n]
        elif n <= 1:                                   # root only / the token_ids[-1:] fallbacks: one word is written
            rm = np.zeros((n, W), dtype=np.uint64)
            rm[:, 0] = 1
        else:
            rm = self._rowmask[:n * W].reshape(n, W)
        return self._ids[:n], rm, self._parent[:n], self._sizes[:nsizes].tolist()

    def one_get(self, token_ids, decoding_length=64, branch_length=8, min_input_size=0, min_output_size=0,
                mode='mix', idx=0):
        """lookahead_cache.py:490-517 -> single greedy chain, lower-triangular mask."""
        assert mode in ('input', 'output', 'mix')
        self._sync_live()
        self._alloc(max(int(branch_length) + 1, 1))
        arr, p, n = _as_i32(token_ids)
        check(lib.la_cache_one_get(self._h, p, n, int(decoding_length), int(branch_length), _MODES[mode], int(idx),
                                   self._cap, self._ids.ctypes.data_as(_lib.pi32),
                                   self._sizes.ctypes.data_as(_lib.pi32), C.byref(self._nsizes), C.byref(self._n)),
              'one_get')
        n_out, nsizes = self._n.value, self._nsizes.value
        ids = self._ids[:n_out].tolist()
        sizes = self._sizes[:nsizes].tolist()
        if nsizes == 0 or n_out == 0:
            return ids, self.default_mask, sizes
        if nsizes == 2:     # no-match fallback of get_one_branch (:175-177)
            return ids, np.ones((1, 1), dtype=np.int64), sizes
        return ids, np.tril(np.ones((n_out, n_out), dtype=np.int64), 0), sizes

    def par_get(self, token_ids, decoding_length=16, branch_length=8, min_input_size=0, min_output_size=0,
                mode='mix', idx=0):
        """lookahead_cache.py:441-488: re-lay the hierarchical draft as independent root-to-leaf chains.

        Walk rows from last to first, keep a row's ancestor set unless an already kept set covers it (so only
        maximal paths survive), lay the kept paths out one after another (truncated to the draft budget) under
        a block mask in which every chain sees the root and itself only."""
        assert mode in ('input', 'output', 'mix')
        self._sync_live()
        self._alloc(max(int(decoding_length), 1))
        arr, p, n = _as_i32(token_ids)
        # retrieval + re-layout in one native call (la_cache_par_get); par_layout() below is the same re-layout for a draft that
        # came from somewhere else (the device trie)
        check(lib.la_cache_par_get(self._h, p, n, int(decoding_length), int(branch_length), int(min_input_size), int(min_output_size),
                                   _MODES[mode], int(idx), self._cap, self._p_ids, self._p_rowmask, self._p_mask, self._p_sizes,
                                   self._r_nsizes, self._r_n), 'par_get')
        n_out = self._n.value
        if n_out == 0:
            return [], np.tril(np.ones((1, 1)), 0), [0]
        return (self._ids[:n_out].tolist(), self._mask[:n_out * n_out].reshape(n_out, n_out).astype(np.float64),      # float64, as :480
                self._sizes[:1].tolist())

    @staticmethod
    def par_layout(output_ids, decoding_masks):
        """The re-layout step of par_get (lookahead_cache.py:449-488) on a hierarchical draft (ids, 0/1 mask [T][T]) — also applied
        to a draft the DEVICE trie retrieved (pretrained_model.lookahead_prepare_inputs_for_generation)."""
        budget = len(output_ids) - 1
        kept = []
        for row in range(budget, 0, -1):
            members = set(np.nonzero(decoding_masks[row, 1:])[0].tolist())
            if not any(members <= other for other in kept):
                kept.append(members)
        kept.reverse()
        ids = [output_ids[0]] if len(output_ids) > 0 else []
        spans = []
        used = 0
        for members in kept:
            cols = sorted(members)[:budget - used]
            used += len(cols)
            spans.append(len(cols))
            ids.extend(output_ids[c + 1] for c in cols)
            if used >= budget:
                break
        masks = np.tril(np.ones((used + 1, used + 1)), 0)      # float64, as in the reference (:480)
        start = 1
        for length in spans:
            masks[start:start + length, 1:start] = 0
            start += length
        return ids, masks, [start - 1]

    def bat_get(self, token_id_list, decoding_length=64, branch_length=8, decoding_cursors=None, mode='output',
                indices=None, decoding_mode='hier'):
        """lookahead_cache.py:519-561: per-sample drafts, right-padded, mask aligned to the cursors."""
        assert mode in ('input', 'output', 'mix')
        assert decoding_mode in ('hier', 'one')
        bs = len(token_id_list)
        assert bs == len(decoding_cursors) and bs == len(indices), \
            f'{bs=} {len(decoding_cursors)=} {len(indices)=}'
        getter = self.hier_get if decoding_mode == 'hier' else self.one_get
        per_sample = decoding_length // bs                     # the budget is split again here
        id_list, mask_list, size_list = [], [], []
        for sub_idx, token_ids in enumerate(token_id_list):
            ids, masks, sizes = getter(token_ids, decoding_length=per_sample, branch_length=branch_length,
                                       min_input_size=0, min_output_size=max(per_sample // 2, 1), mode=mode,
                                       idx=indices[sub_idx])
            id_list.append(ids)
            mask_list.append(masks)
            size_list.append(sizes)
        lo, hi = min(decoding_cursors), max(decoding_cursors)
        width = max(len(x) for x in id_list)
        out = np.zeros((bs, width, hi - lo + width), dtype=np.int64)
        for i, ids in enumerate(id_list):
            n = len(ids)
            if width > n:
                ids.extend([0] * (width - n))
            off = decoding_cursors[i] - lo
            out[i, :n, off:off + n] = mask_list[i]
            out[i, :, :off + 1] = 1
        return id_list, out, size_list

    def bat_get_packed(self, token_id_list, decoding_length=64, branch_length=8, mode='output', indices=None,
                       decoding_mode='hier'):
        """Device-path form of bat_get: the same per-sample drafts (same budget rule, :534-541) as
        [(ids int32[T_b], rowmask uint64[T_b], sizes)], unpadded and without the [bs,T,W] canvas — the batch engine
        takes each sample's rows as they are.  One native call for the whole batch (la_cache_bat_get_packed)."""
        assert mode in ('input', 'output', 'mix')
        assert decoding_mode in ('hier', 'one')
        self._sync_live()
        bs = len(token_id_list)
        assert bs == len(indices), f'{bs=} {len(indices)=}'
        per_sample = decoding_length // bs
        assert per_sample <= _lib.LA_TREE_MAX or per_sample <= 1, 'device path handles <= 64 tree tokens per sample'
        cap = _lib.LA_TREE_MAX
        qmax = max(1, max(len(q) for q in token_id_list))
        q = np.zeros((bs, qmax), dtype=np.int32)
        nq = np.zeros(bs, dtype=np.int32)
        for b, toks in enumerate(token_id_list):
            nq[b] = len(toks)
            q[b, :len(toks)] = toks
        ids = np.zeros((bs, cap), dtype=np.int32)
        rows = np.zeros((bs, cap), dtype=np.uint64)
        n = np.zeros(bs, dtype=np.int32)
        sizes = np.zeros((bs, 2), dtype=np.int32)
        nsizes = np.zeros(bs, dtype=np.int32)
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        check(lib.la_cache_bat_get_packed(self._h, q.ctypes.data_as(_lib.pi32), nq.ctypes.data_as(_lib.pi32), qmax, bs,
                                          int(decoding_length), int(branch_length), _MODES[mode], idx.ctypes.data_as(_lib.pi32),
                                          1 if decoding_mode == 'one' else 0, cap, ids.ctypes.data_as(_lib.pi32),
                                          rows.ctypes.data_as(_lib.pu64), n.ctypes.data_as(_lib.pi32),
                                          sizes.ctypes.data_as(_lib.pi32), nsizes.ctypes.data_as(_lib.pi32)), 'bat_get_packed')
        return [(ids[b, :n[b]], rows[b, :n[b]], sizes[b, :nsizes[b]].tolist()) for b in range(bs)]

    # ---- maintenance ---------------------------------------------------------------------------------
    def fresh(self):
        """lookahead_cache.py:563-564."""
        check(lib.la_cache_fresh(self._h), 'fresh')

    def reset_input_freqs(self, idx):
        check(lib.la_cache_reset_input_freqs(self._h, int(idx)), 'reset_input_freqs')

    def squeeze_branch_counts(self):
        check(lib.la_cache_squeeze(self._h), 'squeeze_branch_counts')

    def save_mem(self, save_dir):
        """lookahead_cache.py:578-582 (portable binary snapshot instead of pickled Python objects)."""
        check(lib.la_cache_save(self._h, str(save_dir).encode()), 'save_mem')

    def load_mem(self, load_dir):
        check(lib.la_cache_load(self._h, str(load_dir).encode()), 'load_mem')

    def load_reference_mem(self, load_dir):
        """Import a trie written by the REFERENCE's save_mem (lookahead_cache.py:578-582: pickle.dumps(self.mem) as a
        latin-1 string inside JSON).  The pickle is read with an unpickler that only admits the two record classes
        (Node / Tree of lookahead.common.lookahead_cache) and is re-encoded as a native snapshot, preserving child
        insertion order (= dict order) and every per-idx frequency."""
        import io
        import json
        import pickle
        import struct
        import tempfile

        class _Node(object):
            __slots__ = ['freqs', 'children']

        class _Tree(object):
            pass

        class _Restricted(pickle.Unpickler):
            def find_class(self, module, name):
                if module.endswith('lookahead_cache') and name == 'Node':
                    return _Node
                if module.endswith('lookahead_cache') and name == 'Tree':
                    return _Tree
                raise pickle.UnpicklingError(f'refusing to load {module}.{name}')

        with open(load_dir, 'r') as f:
            payload = json.loads(json.load(f)).encode('latin-1')
        mem = _Restricted(io.BytesIO(payload)).load()
        buf = io.BytesIO()
        buf.write(b'LATRIE01')
        buf.write(struct.pack('<q', len(mem)))
        for token, tree in mem.items():
            recs = []
            stack = [(tok, node, 1) for tok, node in reversed(list(tree.nodes.items()))]
            while stack:
                tok, node, depth = stack.pop()
                recs.append((tok, depth, node.freqs))
                stack.extend((t, n, depth + 1) for t, n in reversed(list(node.children.items())))
            buf.write(struct.pack('<6q', int(token), int(tree.max_node), int(tree.max_output_node), int(tree.n_node),
                                  int(tree.n_output_node), len(recs)))
            for tok, depth, freqs in recs:
                fi = [(int(k), float(v)) for k, v in freqs.items() if k != -1]
                buf.write(struct.pack('<3i', int(tok), depth, len(fi)))
                buf.write(struct.pack('<d', float(freqs.get(-1, 0.0))))
                for k, v in fi:
                    buf.write(struct.pack('<id', k, v))
        with tempfile.NamedTemporaryFile(suffix='.latrie', delete=True) as tmp:
            tmp.write(buf.getvalue())
            tmp.flush()
            self.load_mem(tmp.name)

    def stats(self):
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        check(lib.la_cache_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {'n_trees': a.value, 'n_nodes': b.value, 'n_dirty_trees': c.value, 'n_dirty_input_trees': d.value}

    def tree_counters(self, token_id):
        a, b = C.c_int64(), C.c_int64()
        rc = lib.la_cache_tree_counters(self._h, int(token_id), C.byref(a), C.byref(b))
        if rc != 0:
            return None
        return a.value, b.value
