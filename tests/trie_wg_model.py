# -*- coding: utf-8 -*-
"""CPU model of the workgroup-per-query device retrieval (csrc/la_trie_wg.hip, k_trie_hier_get_wg) over the mirror image arrays.

LookaheadCache.hier_get (lookahead_cache.py:408-439) with Tree.get / _match / _dfs_get_freqs / _ravel (:65-154, 224-293) restated as the
LEVEL-SYNCHRONOUS passes the HIP kernel runs — no recursion, no ordered DFS:

  S1  prefix match (as the reference).
  S2  breadth-first expansion of the matched node's descendants, level by level, every node's children appended as ONE contiguous run
      (count pass, then write pass); per entry: record id, parent entry, insertion index k, fi, fo and inF = "reachable through live
      nodes" (the rows _dfs_get_freqs collects).  A node is expanded iff it is inF or shallower than max_length (nodes outside inF only
      matter to _ravel, which stops at depth max_length).
  S3  cut-offs: k-th largest fi / fo over the inF rows (the reference's sorts).
  S4  top-down per level: ok = passes the cut-off rule, depth <= max_length and its parent is ok; rank among its ok siblings by
      (fm desc, k asc); lower bound of its preorder position lb = lb(parent) + 1 + rank; lb >= max_size => the node can never be
      emitted: it is PRUNED (not expanded further, but it weighs "infinitely": every later node of the preorder is cut as well).
  S5  bottom-up per level: size = pruned ? BIG : 1 + sum of the ok children's sizes (saturating).
  S6  top-down per level: pos = pos(parent) + 1 + sum of the sizes of the better ok siblings — the node's position in the DFS preorder.
  S7  emit the nodes with pos < max_size: ids[pos] = token, row mask = ancestors' bits; sizes = emitted rows with fi > 0 / fo > 0.

The truncation of the reference's DFS (`if len(ids) >= max_size: return`) is a PREFIX of the untruncated preorder, which is why positions
computed from full subtree sizes reproduce it.  tests/test_trie_wg_model.py replays every golden trace through this model (CPU)."""
import numpy as np

BIG = 1 << 20
MODE = {'input': 0, 'output': 1, 'mix': 2}


class Image(object):
    def __init__(self, tok, fo, fi, cstart, ccount):
        self.tok, self.fo, self.fi, self.cstart, self.ccount = tok, fo, fi, cstart, ccount

    def find_child(self, u, token):
        cs, cc = int(self.cstart[u]), int(self.ccount[u])
        for k in range(cc):
            if int(self.tok[cs + k]) == token:
                return cs + k
        return -1


def _select_desc(vals, r):
    return sorted(vals, reverse=True)[r]


def tree_get(t, cur, root, q_rest_last, nrest, max_size, max_length, min_in, min_out, mode, form='levels'):
    """Tree.get below the matched node `cur` -> (ids, parent positions, sizes); cur < 0 or childless: the single-row answer.
    form: 'levels' = S4..S6 level by level (the kernel's global-scratch path); 'chains' = the same recurrences unrolled along each entry's
    ancestor chain (the kernel's LDS path, order_emit_chains): rank within the sibling run, lb = sum over the chain of (1 + rank),
    ok = lb(parent) < max_size, size = weights of the ok entries below (pruned: BIGC), position = sum over the chain of (1 + sizes of the
    better siblings)."""
    if cur < 0 or t.ccount[cur] == 0:
        return [q_rest_last if nrest > 0 else int(t.tok[root])], [-1], [0, 0]
    # ---- S2: level-synchronous expansion; entries: node, par (entry index or -1), k, fi, fo, inF; levels: [(start, end)]
    node, par, kk, fi, fo, inF, crun = [], [], [], [], [], [], []
    levels = []
    frontier = [(-1, cur, True)]                       # (entry index, record, expandable as an inF parent)
    depth = 0
    while frontier:
        depth += 1
        start = len(node)
        nxt = []
        for (pe, u, p_inF) in frontier:
            cs, cc = int(t.cstart[u]), int(t.ccount[u])
            c0 = len(node)
            for k in range(cc):
                c = cs + k
                cfi, cfo = float(t.fi[c]), float(t.fo[c])
                live = cfi > 0 or cfo > 0
                e_inF = live and p_inF
                node.append(c); par.append(pe); kk.append(k); fi.append(cfi); fo.append(cfo); inF.append(e_inF); crun.append((0, 0))
                if t.ccount[c] > 0 and (e_inF or depth < max_length):
                    nxt.append((len(node) - 1, c, e_inF))
            if pe >= 0:
                crun[pe] = (c0, len(node) - c0)
        levels.append((start, len(node)))
        frontier = nxt
    n = len(node)
    top_run = (0, levels[0][1])                        # the children of the matched node: entries [0, end of level 1)
    # ---- S3: cut-offs
    rows = sum(1 for e in range(n) if inF[e])
    vfi = [fi[e] if inF[e] else 0.0 for e in range(n)]
    vfo = [fo[e] if inF[e] else 0.0 for e in range(n)]
    TB = 1e9
    w, lo_in, lo_out, lo_mix = 1e-4, TB, TB, TB
    if mode == 0:
        w = 0.0
        cnt = sum(1 for e in range(n) if inF[e] and fi[e] > 0)
        lo_in = _select_desc(vfi, rows - 1 if min_in <= 0 else min(min_in - 1, rows - 1)) if cnt > max_size else 0.0
    elif mode == 1:
        w = 1.0
        cnt = sum(1 for e in range(n) if inF[e] and fo[e] > 0)
        lo_out = _select_desc(vfo, rows - 1 if min_out <= 0 else min(min_out - 1, rows - 1)) if cnt > max_size else 0.0
    elif rows > max_size:
        if min_in > 0:
            lo_in = _select_desc(vfi, min(min_in - 1, rows - 1))
        if min_out > 0:
            lo_out = _select_desc(vfo, min(min_out - 1, rows - 1))
    else:
        lo_mix = 0.0
    w1 = 1.0 - w
    fm = [w1 * fi[e] + w * fo[e] for e in range(n)]     # separately rounded multiplies and add (no FMA), as Python evaluates it

    def skip(e):
        if mode == 2:
            return fi[e] < lo_in and fo[e] < lo_out and fm[e] < lo_mix
        if mode == 0:
            return fi[e] < lo_in
        return fo[e] < lo_out

    def better(a, b):
        return fm[a] > fm[b] or (fm[a] == fm[b] and kk[a] < kk[b])

    def run_of_parent(e):
        return top_run if par[e] < 0 else crun[par[e]]

    if form == 'chains':
        return _order_chains(t, node, par, fi, fo, fm, levels, top_run, crun, skip, better, max_size, max_length, q_rest_last, nrest, root)
    # ---- S4: ok / rank / lower bound / prune, top-down
    ok, pruned, lb = [False] * n, [False] * n, [0] * n
    for d, (s, e_) in enumerate(levels, start=1):
        for e in range(s, e_):
            p = par[e]
            ok[e] = (d <= max_length) and not skip(e) and (p < 0 or (ok[p] and not pruned[p]))
        for e in range(s, e_):
            if not ok[e]:
                continue
            r0, rn = run_of_parent(e)
            rank = sum(1 for j in range(r0, r0 + rn) if j != e and ok[j] and better(j, e))
            lb[e] = (lb[par[e]] if par[e] >= 0 else 0) + 1 + rank
            pruned[e] = lb[e] >= max_size
    # ---- S5: sizes bottom-up
    size = [0] * n
    for (s, e_) in reversed(levels):
        for e in range(s, e_):
            if not ok[e]:
                continue
            if pruned[e]:
                size[e] = BIG
            else:
                c0, cn = crun[e]
                size[e] = min(BIG, 1 + sum(size[j] for j in range(c0, c0 + cn) if ok[j]))
    # ---- S6: preorder positions top-down
    pos = [BIG] * n
    for (s, e_) in levels:
        for e in range(s, e_):
            if not ok[e]:
                continue
            pp = pos[par[e]] if par[e] >= 0 else 0
            if pp >= BIG:
                continue
            r0, rn = run_of_parent(e)
            before = sum(size[j] for j in range(r0, r0 + rn) if j != e and ok[j] and better(j, e))
            pos[e] = min(BIG, pp + 1 + before)
    # ---- S7: emit
    n_out = 1 + sum(1 for e in range(n) if ok[e] and pos[e] < max_size)
    ids, ppos, sizes = [0] * n_out, [-1] * n_out, [0, 0]
    ids[0] = q_rest_last if (nrest > 0 and q_rest_last != 0) else int(t.tok[root])           # :129 (`match_token_id or self.token_id`)
    for e in range(n):
        if ok[e] and pos[e] < max_size:
            ids[pos[e]] = int(t.tok[node[e]])
            ppos[pos[e]] = pos[par[e]] if par[e] >= 0 else -1
            sizes[0] += fi[e] > 0
            sizes[1] += fo[e] > 0
    return ids, ppos, sizes


def _order_chains(t, node, par, fi, fo, fm, levels, top_run, crun, skip, better, max_size, max_length, q_rest_last, nrest, root):
    n = len(node)
    BIGC = 1 << 16
    depth = [0] * n
    for d, (s_, e_) in enumerate(levels, start=1):
        for e in range(s_, e_):
            depth[e] = d
    cand = [depth[e] <= max_length and not skip(e) for e in range(n)]

    def chain(e):                                   # e and its ancestors, or None when one of them is no candidate (an orphan)
        out = []
        while e >= 0:
            if not cand[e]:
                return None
            out.append(e)
            e = par[e]
        return out
    rank = [0] * n
    for e in range(n):
        if cand[e] and (par[e] < 0 or cand[par[e]]):
            r0, rn = top_run if par[e] < 0 else crun[par[e]]
            rank[e] = sum(1 for j in range(r0, r0 + rn) if j != e and cand[j] and better(j, e))
    lb, ok, pruned = [0] * n, [False] * n, [False] * n
    for e in range(n):
        ch = chain(e) if cand[e] else None
        if ch is None:
            continue
        lb[e] = sum(1 + rank[x] for x in ch)
        ok[e] = lb[e] - 1 - rank[e] < max_size
        pruned[e] = lb[e] >= max_size
    size = [0] * n
    for e in range(n):
        if ok[e]:
            for x in chain(e):
                size[x] += BIGC if pruned[e] else 1
    before = [0] * n
    for e in range(n):
        if ok[e]:
            r0, rn = top_run if par[e] < 0 else crun[par[e]]
            before[e] = sum(size[j] for j in range(r0, r0 + rn) if j != e and cand[j] and better(j, e))
    pos = [0] * n
    for e in range(n):
        if ok[e]:
            pos[e] = sum(1 + before[x] for x in chain(e))
    emit = [e for e in range(n) if ok[e] and pos[e] < max_size]
    ids, ppos, sizes = [0] * (1 + len(emit)), [-1] * (1 + len(emit)), [0, 0]
    ids[0] = q_rest_last if (nrest > 0 and q_rest_last != 0) else int(t.tok[root])
    for e in emit:
        ids[pos[e]] = int(t.tok[node[e]])
        ppos[pos[e]] = pos[par[e]] if par[e] >= 0 else -1
        sizes[0] += fi[e] > 0
        sizes[1] += fo[e] > 0
    return ids, ppos, sizes


def hier_get(t, q, decoding_length, branch_length, min_in, min_out, mode, stop_words=(), form='levels'):
    """-> (ids, row masks as Python ints, sizes list) like LookaheadCache.hier_get (masks as per-row integers)."""
    mode = MODE[mode] if isinstance(mode, str) else mode
    nq = len(q)
    if decoding_length <= 1 or branch_length == 0:
        return ([q[-1]] if nq else []), ([1] if nq else []), []
    have, out = False, None
    for i in range(nq):
        root = t.find_child(0, q[i])
        if root < 0:
            continue
        nrest = nq - (i + 1)
        if q[i] in stop_words and nrest == 0:
            continue
        have = True
        cur = root
        for k in range(nrest):
            ch = t.find_child(cur, q[i + 1 + k])
            if ch < 0:
                cur = -1
                break
            cfi, cfo = float(t.fi[ch]), float(t.fo[ch])
            live = cfi > 0 if mode == 0 else cfo > 0 if mode == 1 else (cfi > 0 or cfo > 0)
            cur = ch if live else -1
            if cur < 0:
                break
        ids, ppos, sizes = tree_get(t, cur, root, q[-1], nrest, decoding_length, branch_length, min_in, min_out, mode, form=form)
        out = (ids, ppos, sizes)
        if len(ids) >= branch_length:
            break
    if not have:
        return ([q[-1]] if nq else []), ([1] if nq else []), [0, 0]
    ids, ppos, sizes = out
    rows = []
    for r in range(len(ids)):
        rows.append((rows[ppos[r]] if ppos[r] > -1 else 1) | (1 << r))
    return ids, rows, sizes


def image_of(cache, plane_idx):
    """The mirror image arrays of a native LookaheadCache for ONE input slot (host side; what DeviceTrie uploads)."""
    import ctypes as C
    from painlessinferenceacceleration_amd import _lib
    from painlessinferenceacceleration_amd._lib import check, lib
    pd = C.POINTER(C.c_double)
    arr = np.asarray([int(plane_idx)], dtype=np.int32)
    check(lib.la_cache_mirror_enable(cache._h, arr.ctypes.data_as(_lib.pi32), 1))
    n, full, ni, nd = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    check(lib.la_cache_mirror_state(cache._h, C.byref(n), C.byref(full), C.byref(ni), C.byref(nd)))
    cap = max(n.value, 64)
    tok, cstart, ccount = (np.zeros(cap, np.int32) for _ in range(3))
    fo, fi = np.zeros(cap, np.float64), np.zeros(cap, np.float64)
    check(lib.la_cache_mirror_image(cache._h, cap, tok.ctypes.data_as(_lib.pi32), fo.ctypes.data_as(pd), fi.ctypes.data_as(pd),
                                    cstart.ctypes.data_as(_lib.pi32), ccount.ctypes.data_as(_lib.pi32)))
    return Image(tok, fo, fi, cstart, ccount)
