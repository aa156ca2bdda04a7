# -*- coding: utf-8 -*-
"""ORACLE (test infrastructure, not product code): CPU restatement of the reference trie cache.

Restates lookahead/lookahead/common/lookahead_cache.py of alipay/PainlessInferenceAcceleration
(file:line cited per function) on flat Python lists — no Node/Tree objects, no recursion on dicts
of objects — so it is an independent second implementation of the same algorithm.

Pinning: oracle/gen_golden.py runs the reference itself (imported from /root/reference in the
build container) on seeded operation traces and on the reference's own two known-answer tests
(lookahead/tests/test_lookahead_cache.py:16-45); tests/test_oracle_trie.py replays the committed
traces (tests/golden/trie_*.json) through this file and requires identical outputs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

BIG = 1e9


class TrieOracle(object):
    """Same public surface as the reference LookaheadCache, minus persistence."""

    def __init__(self, eos_ids=(2,), stop_words=None, max_node=65536, max_output_node=512):
        self.eos_ids = eos_ids if eos_ids is not None else [None]      # the live object, as lookahead_cache.py:339
        self.stop_words = stop_words if stop_words is not None else {}
        self.max_node = max_node
        self.max_output_node = max_output_node
        self._reset_arena()
        self.pending = {}            # _output_ids: idx -> rolling list
        self.dirty = set()           # _update_trees (tree uids; dead uids keep counting after fresh())
        self.dirty_input = set()     # _update_input_trees
        self.next_uid = 1

    def _reset_arena(self):
        # node arrays
        self.tok, self.kids, self.fo, self.fi = [], [], [], []   # kids: ordered child ids; fi: {idx: f}
        self.find = {}               # (parent, token) -> child
        # trees
        self.tree_of = {}            # token -> tree record [root, max_node, max_out, n_node, n_out, uid]
        self.by_uid = {}

    # ---------------------------------------------------------------- arena helpers
    def _new_node(self, token):
        self.tok.append(token); self.kids.append([]); self.fo.append(0.0); self.fi.append({})
        return len(self.tok) - 1

    def _freq(self, n, idx):
        return self.fo[n] if idx == -1 else self.fi[n].get(idx, 0.0)

    def _bump(self, n, idx):
        if idx == -1:
            self.fo[n] += 1.0
        else:
            self.fi[n][idx] = self.fi[n].get(idx, 0.0) + 1.0

    def _tree(self, token):
        rec = self.tree_of.get(token)
        if rec is not None:
            return rec, False
        rec = [self._new_node(token), self.max_node, self.max_output_node, 0, 0, self.next_uid]
        self.next_uid += 1
        self.tree_of[token] = rec
        self.by_uid[rec[5]] = rec
        return rec, True

    # ---------------------------------------------------------------- Tree.put/_put/_pack (:33-63)
    def _insert(self, rec, toks, output_mode, idx):
        if output_mode:
            idx = -1
        cur = rec[0]
        for i, t in enumerate(toks):
            ch = self.find.get((cur, t))
            if ch is None:
                rest = toks[i:]
                for t2 in rest:
                    nn = self._new_node(t2)
                    self.kids[cur].append(nn)
                    self.find[(cur, t2)] = nn
                    self._bump(nn, idx)
                    cur = nn
                rec[3] += len(rest)
                if output_mode:
                    rec[4] += len(rest)
                return
            self._bump(ch, idx)
            cur = ch

    def _cut_eos(self, token_ids):
        token_ids = list(token_ids)
        for eos in self.eos_ids:
            if eos in token_ids:
                token_ids = token_ids[:token_ids.index(eos)]
        return token_ids

    # ---------------------------------------------------------------- put (:349-373)
    def put(self, token_ids, branch_length=8, final=False, mode='output', idx=0):
        assert mode in ('input', 'output')
        toks = self._cut_eos(token_ids)
        if len(toks) >= 2:
            for i in range(len(toks) - 1):
                rec, created = self._tree(toks[i])
                self._insert(rec, toks[i + 1:i + 1 + branch_length], mode == 'output', idx)
                if not created:
                    self.dirty.add(rec[5])
                if mode == 'input':
                    self.dirty_input.add(rec[5])
        if final:
            self.reset_input_freqs(idx)
            self.squeeze_branch_counts()

    # ---------------------------------------------------------------- stream_put (:375-406)
    def stream_put(self, token_ids, branch_length=8, final=False, mode='output', idx=0):
        assert mode == 'output' and idx >= 0
        buf = self.pending.setdefault(idx, [])
        buf.extend(self._cut_eos(token_ids))
        ts = len(buf)
        need = 1 if final else branch_length
        if ts > need:
            for i in range(ts - need):
                if buf[i] in self.stop_words:
                    continue
                rec, _ = self._tree(buf[i])
                self._insert(rec, buf[i + 1:i + 1 + branch_length], True, idx)
                self.dirty.add(rec[5])
            if not final:
                self.pending[idx] = buf[ts - branch_length:]
        if final:
            self.pending[idx] = []
            self.reset_input_freqs(idx)
            self.squeeze_branch_counts()

    # ---------------------------------------------------------------- Tree._match (:224-246)
    def _descend(self, rec, query, mode, idx):
        """-> (node whose children are the candidate set, or None for an empty set; last query token or None)."""
        cur, last = rec[0], None
        for t in query:
            last = t
            ch = None if cur is None else self.find.get((cur, t))
            if ch is None:
                return None, last
            fi, fo = self._freq(ch, idx), self.fo[ch]
            live = fi > 0 if mode == 'input' else fo > 0 if mode == 'output' else (fi > 0 or fo > 0)
            cur = ch if live else None
        return cur, last

    # ---------------------------------------------------------------- Tree.get (:65-144) + _dfs_get_freqs + _ravel
    def _draft(self, rec, query, max_size, max_length, min_input_size, min_output_size, mode, idx):
        at, last = self._descend(rec, query, mode, idx)
        if at is None or len(self.kids[at]) == 0:
            tok = query[-1] if len(query) > 0 else self.tok[rec[0]]
            return [tok], np.ones((1, 1), dtype=np.int64), [0, 0]

        # live sub-forest statistics (:146-154)
        fis, fos = [], []
        stack = list(self.kids[at])
        while stack:
            n = stack.pop()
            fi, fo = self._freq(n, idx), self.fo[n]
            if fi > 0 or fo > 0:
                fis.append(fi); fos.append(fo)
                stack.extend(self.kids[n])

        def kth(vals, k):              # sorted(desc)[k-1]; k == 0 hits Python's index -1 = the minimum
            return sorted(vals, reverse=True)[k - 1]

        w = 1e-4
        lo_in = lo_out = lo_mix = BIG
        if mode == 'input':
            w = 0.0
            lo_in = kth(fis, min_input_size) if sum(1 for f in fis if f > 0) > max_size else 0.0
        elif mode == 'output':
            w = 1.0
            lo_out = kth(fos, min_output_size) if sum(1 for f in fos if f > 0) > max_size else 0.0
        elif len(fis) > max_size:
            # rows carry None as their index (:152) => the `indices` set is {None} and the mix cut-off loop
            # (:111-123) never fires: lo_mix stays 1e9
            if min_input_size > 0:
                lo_in = kth(fis, min_input_size)
            if min_output_size > 0:
                lo_out = kth(fos, min_output_size)
        else:
            lo_mix = 0.0

        ids = [last if last else self.tok[rec[0]]]          # `match_token_id or self.token_id` (:129)
        rows = [1]                                           # bit masks; row 0 = {0}
        sizes = [0, 0]
        # explicit stack of (iterator over sorted children, parent row, remaining depth)
        def ordered(node):
            ks = self.kids[node]
            keyed = [((1.0 - w) * self._freq(k, idx) + w * self.fo[k], k) for k in ks]
            order = sorted(range(len(ks)), key=lambda i: -keyed[i][0])   # stable: ties keep insertion order
            return iter([keyed[i] for i in order])

        if max_length > 0 and len(ids) < max_size:
            frames = [(ordered(at), -1, max_length)]
            while frames:
                it, prow, depth_left = frames[-1]
                if len(ids) >= max_size:
                    break
                nxt = next(it, None)
                if nxt is None:
                    frames.pop()
                    continue
                fm, n = nxt
                fi, fo = self._freq(n, idx), self.fo[n]
                if mode == 'mix':
                    if fi < lo_in and fo < lo_out and fm < lo_mix:
                        continue
                elif mode == 'input':
                    if fi < lo_in:
                        continue
                elif fo < lo_out:
                    continue
                if fi > 0.0:
                    sizes[0] += 1
                if fo > 0.0:
                    sizes[1] += 1
                ids.append(self.tok[n])
                r = len(ids) - 1
                rows.append((rows[prow] if prow > -1 else 1) | (1 << r))
                if self.kids[n] and depth_left - 1 > 0 and len(ids) < max_size:
                    frames.append((ordered(n), r, depth_left - 1))
        T = len(ids)
        mask = np.zeros((T, T), dtype=np.int64)
        for i, bits in enumerate(rows):
            for j in range(T):
                mask[i, j] = (bits >> j) & 1
        return ids, mask, sizes

    # ---------------------------------------------------------------- hier_get (:408-439)
    def hier_get(self, token_ids, decoding_length=64, branch_length=8, min_input_size=0, min_output_size=0,
                 mode='mix', idx=0):
        assert mode in ('input', 'output', 'mix')
        token_ids = list(token_ids)
        if decoding_length <= 1 or branch_length == 0:
            return token_ids[-1:], np.ones((1, 1), dtype=np.int64), []
        out = None
        for i, t in enumerate(token_ids):
            rec = self.tree_of.get(t)
            if rec is None:
                continue
            rest = token_ids[i + 1:]
            if t in self.stop_words and len(rest) == 0:
                continue
            out = self._draft(rec, rest, decoding_length, branch_length, min_input_size, min_output_size, mode, idx)
            if len(out[0]) >= branch_length:
                break                                         # otherwise a later suffix overwrites this result
        if out is None:
            return token_ids[-1:], np.ones((1, 1), dtype=np.int64), [0, 0]
        return out

    # ---------------------------------------------------------------- one_get / get_one_branch (:171-222, 490-517)
    def _one_branch(self, rec, query, max_length, mode, idx):
        """Greedy single chain below the matched prefix: at every level the live child with the largest key, first child
        winning ties (strict '>'); keys: input/output mode = that frequency, mix = 10000 * output + input (the reference
        names them the other way round, :188-191, the arithmetic is this)."""
        parent, matched = self._descend(rec, query, mode, idx)
        if parent is None or not self.kids[parent]:
            return [query[-1] if len(query) > 0 else self.tok[rec[0]]], np.ones((1, 1), dtype=np.int64), [0, 0]
        ids = [matched or self.tok[rec[0]]]
        length = 0
        while self.kids[parent] and length < max_length:
            best, best_key = None, 0.0
            for ch in self.kids[parent]:
                fi, fo = self._freq(ch, idx), self._freq(ch, -1)
                if mode == 'mix':
                    key = 10000 * fo + fi if (fi > 0 or fo > 0) else 0.0
                elif mode == 'input':
                    key = fi
                else:
                    key = fo
                if key > 0 and key > best_key:
                    best, best_key = ch, key
            if best is None:
                break
            ids.append(self.tok[best])
            parent = best
            length += 1
        return ids, np.tril(np.ones((length + 1, length + 1), dtype=np.int64), 0), [length]

    def one_get(self, token_ids, decoding_length=64, branch_length=8, min_input_size=0, min_output_size=0, mode='mix',
                idx=0):
        assert mode in ('input', 'output', 'mix')
        token_ids = list(token_ids)
        mask = np.ones((1, 1), dtype=np.int64)
        if decoding_length <= 1 or branch_length == 0:
            return token_ids[-1:], mask, []
        ids, sizes = None, [0, 0]
        for i, t in enumerate(token_ids):
            rec = self.tree_of.get(t)
            if rec is None:
                continue
            rest = token_ids[i + 1:]
            if t in self.stop_words and len(rest) == 0:
                continue
            ids, mask, sizes = self._one_branch(rec, rest, branch_length, mode, idx)
            if len(ids) >= branch_length // 2:
                break
        if ids is None:
            ids = token_ids[-1:]
        return ids, mask, sizes

    # ---------------------------------------------------------------- par_get (:441-488)
    def par_get(self, token_ids, decoding_length=16, branch_length=8, min_input_size=0, min_output_size=0, mode='mix',
                idx=0):
        """The hierarchical draft re-laid as independent root-to-leaf chains: walking rows from last to first, a row's
        ancestor set is kept unless a kept set already contains it; kept paths are concatenated (cut at the draft budget)
        under a mask in which each chain sees the root and itself."""
        out_ids, masks, _ = self.hier_get(token_ids, decoding_length=decoding_length, branch_length=branch_length,
                                          min_input_size=min_input_size, min_output_size=min_output_size, mode=mode, idx=idx)
        budget = len(out_ids) - 1
        kept = []
        for row in range(budget, 0, -1):
            members = set(np.nonzero(masks[row, 1:])[0].tolist())
            if not any(len(members - other) == 0 for other in kept):
                kept.append(members)
        kept.reverse()
        ids, spans, used = [out_ids[0]], [], 0
        for members in kept:
            cols = sorted(members)[:budget - used]
            used += len(cols)
            spans.append(len(cols))
            ids.extend(out_ids[c + 1] for c in cols)
            if used >= budget:
                break
        m = np.tril(np.ones((used + 1, used + 1)), 0)
        start = 1
        for n in spans:
            m[start:start + n, 1:start] = 0
            start += n
        return ids, m, [start - 1]

    # ---------------------------------------------------------------- bat_get (:519-561)
    def bat_get(self, token_id_list, decoding_length=64, branch_length=8, decoding_cursors=None, mode='output',
                indices=None, decoding_mode='hier'):
        """Per-sample hierarchical drafts with the budget `decoding_length // bs` (the caller has already divided
        once, :534 divides again), right-padded with id 0, masks laid on a [bs, W, max_cur-min_cur+W] canvas at the
        sample's cursor offset; every column up to and including the sample's own root column is forced to 1."""
        assert decoding_mode in ('hier', 'one')
        getter = self.hier_get if decoding_mode == 'hier' else self.one_get
        bs = len(token_id_list)
        assert bs == len(decoding_cursors) == len(indices)
        per = decoding_length // bs
        got = [getter(q, decoding_length=per, branch_length=branch_length, min_input_size=0,
                             min_output_size=max(per // 2, 1), mode=mode, idx=indices[i])
               for i, q in enumerate(token_id_list)]
        lo, hi = min(decoding_cursors), max(decoding_cursors)
        W = max(len(g[0]) for g in got)
        canvas = np.zeros((bs, W, hi - lo + W), dtype=np.int64)
        id_list = []
        for i, (ids, m, _) in enumerate(got):
            n, off = len(ids), decoding_cursors[i] - lo
            id_list.append(list(ids) + [0] * (W - n))
            canvas[i, :n, off:off + n] = m
            canvas[i, :, :off + 1] = 1
        return id_list, canvas, [g[2] for g in got]

    # ---------------------------------------------------------------- maintenance (:295-333, 563-576)
    def fresh(self):
        self._reset_arena()

    def reset_input_freqs(self, idx):
        for uid in self.dirty_input:
            rec = self.by_uid.get(uid)
            if rec is None:
                continue
            stack = [rec[0]]
            while stack:
                p = stack.pop()
                for ch in self.kids[p]:
                    if self._freq(ch, idx) == 0.0:
                        continue
                    if idx == -1:
                        self.fo[ch] = 0.0
                    else:
                        self.fi[ch][idx] = 0.0
                    stack.append(ch)
        self.dirty_input.clear()

    def _halve_or_drop(self, parent):
        keep = []
        for ch in self.kids[parent]:
            if self.fo[ch] > 1.0:
                self.fo[ch] *= 0.5
                self._halve_or_drop(ch)
                keep.append(ch)
            else:
                self._unindex(parent, ch)
        self.kids[parent] = keep

    def _unindex(self, parent, ch):
        del self.find[(parent, self.tok[ch])]
        for g in self.kids[ch]:
            self._unindex(ch, g)
        self.kids[ch] = []

    def _count(self, parent):
        return sum(1 + self._count(ch) for ch in self.kids[parent])

    def squeeze_branch_counts(self):
        if len(self.dirty) >= 1024:
            for uid in self.dirty:
                rec = self.by_uid.get(uid)
                if rec is None:
                    continue
                if rec[3] > rec[1] or rec[4] > rec[2]:
                    self._halve_or_drop(rec[0])
                    rec[3] = rec[4] = self._count(rec[0])
            self.dirty.clear()

    # ---------------------------------------------------------------- introspection for tests
    def n_trees(self):
        return len(self.tree_of)

    def n_nodes(self):
        return sum(self._count(rec[0]) for rec in self.tree_of.values())
