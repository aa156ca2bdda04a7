// la_trie_wg.hip — device-side trie retrieval, ONE WORKGROUP PER QUERY (round 6): LookaheadCache.hier_get (lookahead_cache.py:408-439) with
// Tree.get / _match / _dfs_get_freqs / _ravel (:65-154, 224-293) as level-synchronous passes of 256 threads — no recursion and no ordered DFS,
// draft trees of up to LA_TREE_WIDE_MAX rows with multi-word row masks.  tests/trie_wg_model.py is the CPU statement of the same passes
// (replayed over every golden trace by tests/test_trie_wg_model.py); the one-wavefront kernel of la_trie_dev.hip stays as the second opinion.
//
//   S1  prefix match                      : 256 children compared per step                                         (:224-246)
//   S2  breadth-first expansion           : the matched node's descendants level by level, every node's children appended as ONE contiguous run
//                                           (work item = child, its parent found by a binary search over the level's run starts in LDS); one
//                                           dependent HBM round trip per LEVEL instead of one per node; per entry {parent, token, fi, fo, depth,
//                                           inF = reachable through live nodes} streamed to a per-query scratch               (:146-154)
//   S3  cut-offs                          : radix select (8 x 8 bits) of the k-th largest fi / fo over the inF rows, values staged in LDS  (:78-125)
//   C   candidates                        : entries that pass the cut-off rule within max_length levels, compacted (stable: sibling runs stay
//                                           contiguous and in insertion order) into LDS; parent / child-run indices re-derived by binary search
//   S4  top-down per level                : ok = parent ok and not pruned; rank among the ok siblings by (fm desc, insertion asc); lower bound of the
//                                           preorder position lb = lb(parent) + 1 + rank; lb >= max_size: PRUNED (never emitted, weighs "infinitely")
//   S5  bottom-up per level               : subtree size = 1 + sizes of the ok children (saturating)
//   S6  top-down per level                : preorder position = pos(parent) + 1 + sizes of the better ok siblings               (:248-293)
//   S7  emit                              : rows with pos < max_size: ids[pos], ancestor mask = bits of the ancestors' positions
// The reference's DFS truncation (`if len(ids) >= max_size: return`) keeps a PREFIX of the untruncated preorder, which is why positions computed from
// full subtree sizes reproduce it.  Bit-exact to the host trie / the reference: fm = separately rounded fp64 multiplies and add (no FMA contraction).
// Sets that outgrow LDS (a level above LA_WG_LCAP entries, more than LA_WG_MCAP candidates) run the same code over the global scratch: the passes are
// templates over the pointer types (address_space(3) = ds_read / ds_write, generic = the scratch), not flat accesses that decide per instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "la_kernels.h"
#include "la_trie_dev.h"
#include "la_knobs.h"

#define WGT 256
#define NWV 4
#define LA_WG_LCAP 4096          // entries of one level whose run starts live in LDS
#define LA_WG_MCAP 3072          // candidate entries that live in LDS
#define LA_WG_SEL 4096           // values the radix select stages in LDS
#define LA_WG_ONEWAVE 256        // candidate sets up to here are ordered by wave 0 alone (no s_barrier in the level steps)
#define LA_WG_MAXLV 128          // deepest level followed (a put inserts branch_length + 1 tokens)
#define LA_WG_POOL (LA_WG_MCAP * 34)
#define WBIG (1 << 20)
#define LA_WG_BIGC (1 << 16)      // weight of a pruned entry in the chain form (M x LA_WG_BIGC < 2^31)
#define WTBIG 1e9
#define ORPHAN (-2)
#define F_FI 1
#define F_FO 2
#define F_OK 4
#define F_PRUNED 8
#define G_INF 1                  // g_fl: bit 0 = inF, bits 8.. = depth

static_assert(2 * LA_WG_LCAP * 8 <= LA_WG_POOL && LA_WG_SEL * 8 <= LA_WG_POOL, "the S2 / S3 buffers alias the candidate arrays");

template <typename T> using lds_ptr = __attribute__((address_space(3))) T*;

// workgroup barrier that orders LDS only: __syncthreads() also drains the wave's global stores (s_waitcnt vmcnt(0)), which the expansion streams
// to the scratch and nobody reads before the next full barrier
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <typename P> struct in_lds { static constexpr bool value = false; };
template <typename T> struct in_lds<lds_ptr<T>> { static constexpr bool value = true; };

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int x = __shfl_up(v, o, 64);
        if (lane >= o) v += x;
    }
    return v;
}

// value at position r (0-based) of the DESCENDING sort of {inF(i) ? vals[i] : 0} (all >= 0), i < n: radix select on the bit patterns.
// Bytes on which ALL values agree (OR == AND over the set: the low mantissa bytes of small integer counts, typically 5 of the 8) need no pass.
__device__ double select_desc_wg(const double* vals, const int* fl, int n, int r, int tid, unsigned* hist, lds_ptr<double> sv, int* s_sel,
                                 unsigned long long* s_oa) {
    const bool staged = n <= LA_WG_SEL;
    if (tid == 0) { s_oa[0] = 0ull; s_oa[1] = ~0ull; }
    __syncthreads();                                                        // the pool's previous tenants are done
    unsigned long long diff = ~0ull, vand = 0ull;
    if (staged) {
        unsigned long long o = 0ull, an = ~0ull;
        for (int i = tid; i < n; i += WGT) {
            const double v = (fl[i] & G_INF) ? vals[i] : 0.0;
            sv[i] = v;
            const unsigned long long bts = (unsigned long long)__double_as_longlong(v);
            o |= bts; an &= bts;
        }
#pragma unroll
        for (int k = 32; k > 0; k >>= 1) { o |= __shfl_xor(o, k, 64); an &= __shfl_xor(an, k, 64); }
        if ((tid & 63) == 0) { atomicOr(&s_oa[0], o); atomicAnd(&s_oa[1], an); }
        __syncthreads();
        vand = s_oa[1];
        diff = s_oa[0] ^ vand;
    }
    unsigned long long prefix = 0ull, mask = 0ull;
    int rank = r;
    for (int byte = 7; byte >= 0; --byte) {
        const int sh = byte * 8;
        if (((diff >> sh) & 0xffull) == 0ull) {                             // every value carries the same byte here
            prefix |= vand & (0xffull << sh);
            mask |= 0xffull << sh;
            continue;
        }
        hist[tid] = 0u;
        __syncthreads();
        for (int i = tid; i < n; i += WGT) {
            const double v = staged ? sv[i] : ((fl[i] & G_INF) ? vals[i] : 0.0);
            const unsigned long long bts = (unsigned long long)__double_as_longlong(v);
            if ((bts & mask) == prefix) atomicAdd(&hist[(unsigned)((bts >> sh) & 0xffull)], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            // lane l owns bins 255 - 4l .. 252 - 4l (descending): above = values in higher bins = exclusive prefix over lanes
            const int top = 255 - 4 * tid;
            const int h0 = (int)hist[top], h1 = (int)hist[top - 1], h2 = (int)hist[top - 2], h3 = (int)hist[top - 3];
            const int mine = h0 + h1 + h2 + h3;
            const int incl = wave_incl_scan(mine, tid);
            const int above = incl - mine;
            if (rank >= above && rank < incl) {                             // exactly one lane (rank < n = total)
                int acc = above, bin = top;
                if (acc + h0 <= rank) { acc += h0; bin = top - 1;
                    if (acc + h1 <= rank) { acc += h1; bin = top - 2;
                        if (acc + h2 <= rank) { acc += h2; bin = top - 3; } } }
                s_sel[0] = bin; s_sel[1] = rank - acc;
            }
        }
        __syncthreads();
        prefix |= (unsigned long long)s_sel[0] << sh;
        rank = s_sel[1];
        mask |= 0xffull << sh;
    }
    __syncthreads();
    return __longlong_as_double((long long)prefix);
}

// child of node u with token `token`, or -1 (uniform over the workgroup)
__device__ int find_child_wg(const TrieDev& t, int u, int token, int tid, int* s_found) {
    const int cs = t.cstart[u], cc = t.ccount[u];
    if (tid == 0) *s_found = 0x7fffffff;
    __syncthreads();
    for (int i = tid; i < cc; i += WGT)
        if (t.tok[cs + i] == token) atomicMin(s_found, cs + i);
    __syncthreads();
    const int r = *s_found;
    __syncthreads();
    return r == 0x7fffffff ? -1 : r;
}

// first index in [0, n) with a[i] >= key / > key (a non-decreasing)
template <typename P>
__device__ __forceinline__ int lower_bound_i(P a, int n, int key) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
template <typename P>
__device__ __forceinline__ int upper_bound_i(P a, int n, int key) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)a[mid] <= key) lo = mid + 1; else hi = mid; }
    return lo;
}

struct WgCtx {
    TrieDev t;
    int tid, lane, wv;
    int max_size, max_length, mode;
    double w, w1;
    int* g_par; int* g_tok; int* g_fl; double* g_fi; double* g_fo;
    int* s_w; int* s_c; int* s_lv; int* s_x;
    int* oid; unsigned long long* orm; int W;
};

// ---- S2, one level: the `total` children of the level's `np` parents P (run start = absolute entry index, child block | inF << 31) become the
// entries [ebase, ebase + total); Q receives their own {children to expand, child block | inF << 31}, then the run starts.  -> entries of the next level
template <typename PP, typename QP>
__device__ __forceinline__ int expand_level(const WgCtx& c, PP P, int np, int ps, QP Q, int ebase, int total, int depth,
                                            int& c_rows, int& c_fi, int& c_fo) {
    const TrieDev& t = c.t;
    auto level_barrier = [&]() { if (in_lds<QP>::value) lds_barrier(); else __syncthreads(); };    // what the barriers publish is Q (and s_w)
    for (int j = c.tid; j < total; j += WGT) {
        const int e = ebase + j;
        int lo = 0, hi = np;                                      // the parent: last p with P[p].x <= e
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (P[mid].x <= e) lo = mid; else hi = mid; }
        const int px = P[lo].x, py = P[lo].y;
        const int ch = (int)((unsigned)py & 0x7fffffffu) + (e - px);
        const double cfi = t.fi[ch], cfo = t.fo[ch];
        const int ccs = t.cstart[ch], ccc = t.ccount[ch], ctok = t.tok[ch];
        const bool inF = (cfi > 0 || cfo > 0) && py < 0;
        const bool expand = ccc > 0 && (inF || depth < c.max_length);
        c.g_par[e] = ps + lo; c.g_tok[e] = ctok; c.g_fi[e] = cfi; c.g_fo[e] = cfo; c.g_fl[e] = (depth << 8) | (inF ? G_INF : 0);
        Q[j].x = expand ? ccc : 0; Q[j].y = (int)((unsigned)ccs | (inF ? 0x80000000u : 0u));
        c_rows += inF; c_fi += inF && cfi > 0; c_fo += inF && cfo > 0;
    }
    level_barrier();
    // Q[i].x <- ebase + total + exclusive prefix of the counts; one barrier per 256 items
    const int base = ebase + total;
    int carry = 0, it = 0;
    for (int c0 = 0; c0 < total; c0 += WGT, ++it) {
        const int i = c0 + c.tid;
        const int v = i < total ? Q[i].x : 0;
        const int inc = wave_incl_scan(v, c.lane);
        int* sw = c.s_w + (it & 1) * NWV;
        if (c.lane == 63) sw[c.wv] = inc;
        level_barrier();
        int pre = 0, tot = 0;
#pragma unroll
        for (int x = 0; x < NWV; ++x) { const int y = sw[x]; pre += x < c.wv ? y : 0; tot += y; }
        if (i < total) Q[i].x = base + carry + pre + inc - v;
        carry += tot;
    }
    level_barrier();
    return carry;
}

template <typename IP, typename DP, typename CP>
struct Cand { IP par, c0, c1, A, B, tok; DP fm; CP fl, dep; };

// ---- C: entries that pass the cut-off rule within max_length levels, compacted in entry order (sibling runs stay contiguous and in insertion
// order); A = the old entry index, B = the old parent (both non-decreasing).  -> their number (only the first dcap are stored)
template <typename C>
__device__ __forceinline__ int compact_cands(const WgCtx& c, C d, int dcap, int n, double lo_in, double lo_out, double lo_mix) {
    __syncthreads();
    int M = 0, it = 0;
    for (int c0 = 0; c0 < n; c0 += 4 * WGT, ++it) {
        bool cand[4]; double fmv[4]; int fl4[4], dep4[4];
        unsigned long long bm[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = c0 + k * WGT + c.tid;
            cand[k] = false; fmv[k] = 0.0; fl4[k] = 0; dep4[k] = 0;
            if (e < n) {
                const int dep = c.g_fl[e] >> 8;
                const double cfi = c.g_fi[e], cfo = c.g_fo[e];
                const double fm = __dadd_rn(__dmul_rn(c.w1, cfi), __dmul_rn(c.w, cfo));       // :254, no FMA
                bool skip;
                if (c.mode == LA_MODE_MIX) skip = cfi < lo_in && cfo < lo_out && fm < lo_mix;   // :265
                else if (c.mode == LA_MODE_INPUT) skip = cfi < lo_in;
                else skip = cfo < lo_out;
                cand[k] = dep <= c.max_length && !skip;
                fmv[k] = fm; dep4[k] = dep; fl4[k] = (cfi > 0 ? F_FI : 0) | (cfo > 0 ? F_FO : 0);
            }
        }
        int* sw = c.s_c + (it & 1) * 4 * NWV;
#pragma unroll
        for (int k = 0; k < 4; ++k) { bm[k] = __ballot(cand[k]); if (c.lane == 0) sw[k * NWV + c.wv] = __popcll(bm[k]); }
        __syncthreads();
        int run = M;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int pre = 0, tot = 0;
#pragma unroll
            for (int x = 0; x < NWV; ++x) { const int v = sw[k * NWV + x]; pre += x < c.wv ? v : 0; tot += v; }
            const int m = run + pre + __popcll(bm[k] & ((1ull << c.lane) - 1ull));
            if (cand[k] && m < dcap) {
                const int e = c0 + k * WGT + c.tid;
                d.A[m] = e; d.B[m] = c.g_par[e]; d.tok[m] = c.g_tok[e]; d.fm[m] = fmv[k];
                d.fl[m] = (unsigned char)fl4[k]; d.dep[m] = (unsigned char)dep4[k];
            }
            run += tot;
        }
        M = run;
    }
    __syncthreads();
    return M;
}

// ---- S4 .. S7 over the M candidates, LEVEL form (any M: the form the scratch fallback runs; tests/trie_wg_model.py states it); the rows emitted
// behind the root row and their sizes are ADDED to s_x[6..8] (the caller's barrier publishes them).  ONEWAVE: the whole pass by wave 0 alone (the
// caller sends only that wave in) — LDS operations of one wave execute in order, so the level steps need no s_barrier, only a compiler fence.
template <bool ONEWAVE, typename C>
__device__ __forceinline__ void order_emit(const WgCtx& c, C d, int M, long long* stamp) {
    constexpr int NT = ONEWAVE ? 64 : WGT;
    const int tid = c.tid, max_size = c.max_size;
    auto bar = [&]() {
        if (ONEWAVE) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    };
    int* const s_lv = c.s_lv;
    const int Dl = min(c.max_length, LA_WG_MAXLV);
    // parents, child runs and level starts in the compacted numbering (old index / old parent are non-decreasing)
    for (int m = tid; m < M; m += NT) {
        const int op = d.B[m], me = d.A[m];
        int pp = -1;
        if (op >= 0) { pp = lower_bound_i(d.A, M, op); if (pp >= M || d.A[pp] != op) pp = ORPHAN; }
        d.par[m] = pp; d.c0[m] = lower_bound_i(d.B, M, me); d.c1[m] = upper_bound_i(d.B, M, me);
    }
    for (int x = tid + 1; x <= Dl + 1; x += NT) s_lv[x] = lower_bound_i(d.dep, M, x);
    bar();
    int Dn = 0;                                                       // levels that hold candidates
    for (int dd = 1; dd <= Dl; ++dd) { if (s_lv[dd] < s_lv[dd + 1]) Dn = dd; else break; }
    if (stamp && tid == 0) stamp[6] = wall_clock64();
    // Every candidate of one sibling run has the same parent, so the whole run is ok or not together (ok = the parent is ok and not
    // pruned): the rank / size / position loops read no sibling flags, and a level without an ok, unpruned entry ends the descent.
    // ---- S4: ok / rank / lower bound / prune, top-down
    for (int dd = 1; dd <= Dn; ++dd) {
        const int ls = s_lv[dd], le = s_lv[dd + 1];
        int alive = 0;
        for (int m = ls + tid; m < le; m += NT) {
            const int p = d.par[m];
            if (p == ORPHAN || (p >= 0 && (d.fl[p] & (F_OK | F_PRUNED)) != F_OK)) continue;
            const int r0 = p < 0 ? 0 : d.c0[p], r1 = p < 0 ? s_lv[2] : d.c1[p];
            const int lbp = p < 0 ? 0 : d.B[p];
            const int room = max_size - lbp - 1;                      // rank >= room: lb >= max_size
            const double fmm = d.fm[m];
            int rank = 0;
            for (int j = r0; j < r1 && rank < room; ++j) {
                const double fj = d.fm[j];
                rank += (fj > fmm) || (fj == fmm && j < m);
            }
            const int lb = lbp + 1 + rank;
            d.B[m] = lb;
            d.fl[m] = (unsigned char)(d.fl[m] | (lb >= max_size ? (F_OK | F_PRUNED) : F_OK));
            alive |= lb < max_size;
        }
        bool any;
        if (ONEWAVE) { any = __ballot(alive != 0) != 0ull; bar(); }
        else any = __syncthreads_or(alive) != 0;
        if (!any) { Dn = dd; break; }
    }
    // ---- S5: subtree sizes, bottom-up
    for (int dd = Dn; dd >= 1; --dd) {
        const int ls = s_lv[dd], le = s_lv[dd + 1];
        for (int m = ls + tid; m < le; m += NT) {
            const int f = d.fl[m];
            if (!(f & F_OK)) continue;
            int sz = WBIG;
            if (!(f & F_PRUNED)) {
                sz = 1;
                if (dd < Dn) {
                    const int r0 = d.c0[m], r1 = d.c1[m];
                    for (int j = r0; j < r1 && sz < WBIG; ++j) sz += d.A[j];
                }
                sz = min(sz, WBIG);
            }
            d.A[m] = sz;
        }
        bar();
    }
    // ---- S6: preorder positions, top-down
    for (int dd = 1; dd <= Dn; ++dd) {
        const int ls = s_lv[dd], le = s_lv[dd + 1];
        for (int m = ls + tid; m < le; m += NT) {
            if (!(d.fl[m] & F_OK)) continue;
            const int p = d.par[m];
            const int pp = p < 0 ? 0 : d.B[p];
            int pos = WBIG;
            if (pp < max_size - 1) {
                const int r0 = p < 0 ? 0 : d.c0[p], r1 = p < 0 ? s_lv[2] : d.c1[p];
                const int room = max_size - pp - 1;                   // before >= room: pos >= max_size
                const double fmm = d.fm[m];
                int before = 0;
                for (int j = r0; j < r1 && before < room; ++j) {
                    const double fj = d.fm[j];
                    if ((fj > fmm) || (fj == fmm && j < m)) before += d.A[j];
                }
                pos = min(WBIG, pp + 1 + before);
            }
            d.B[m] = pos;
        }
        bar();
    }
    // ---- S7: emit
    int e_cnt = 0, e_fi = 0, e_fo = 0;
    const int Mend = Dn > 0 ? s_lv[Dn + 1] : 0;
    const int W = c.W;
    for (int m = tid; m < Mend; m += NT) {
        const int f = d.fl[m];
        if (!(f & F_OK)) continue;
        const int pos = d.B[m];
        if (pos >= max_size) continue;
        c.oid[pos] = d.tok[m];
        unsigned long long m0 = 1ull, m1 = 0ull, m2 = 0ull, m3 = 0ull;
        for (int x = m; x >= 0; x = d.par[x]) {
            const int px = d.B[x];
            const unsigned long long bit = 1ull << (px & 63);
            const int wd = px >> 6;
            m0 |= wd == 0 ? bit : 0ull; m1 |= wd == 1 ? bit : 0ull; m2 |= wd == 2 ? bit : 0ull; m3 |= wd == 3 ? bit : 0ull;
        }
        unsigned long long* o = c.orm + (size_t)pos * W;
        o[0] = m0;
        if (W > 1) o[1] = m1;
        if (W > 2) o[2] = m2;
        if (W > 3) o[3] = m3;
        ++e_cnt; e_fi += (f & F_FI) ? 1 : 0; e_fo += (f & F_FO) ? 1 : 0;
    }
    atomicAdd(&c.s_x[6], e_cnt); atomicAdd(&c.s_x[7], e_fi); atomicAdd(&c.s_x[8], e_fo);
}

// ---- S4 .. S7 WITHOUT level steps (candidate sets that live in LDS, <= LA_WG_MCAP): the per-level recurrences unrolled along each entry's
// ancestor chain, so the pass count does not grow with the depth of the trie (the level form pays ~3 dependent LDS round trips x 3 passes per
// level: 22 us for a median of 34 candidates over 12 levels).
//   rank(m)   among its sibling run (one run = one parent = ok or not together)                               -> A
//   lb(m)     = sum over the chain of (1 + rank): increasing along a chain, so  ok(m) = no orphan above and lb(parent) < max_size,
//             pruned(m) = lb(m) >= max_size
//   size      every ok entry adds its weight (1, pruned: LA_WG_BIGC) to itself and all its ancestors (LDS atomics)   -> B
//   before(m) = sizes of the better siblings                                                                   -> A
//   pos(m)    = sum over the chain of (1 + before)                                                             -> c1
// Equal to the saturating level form wherever a position is below max_size (sums stay below 2^31: M x LA_WG_BIGC).
template <bool ONEWAVE, typename C>
__device__ __forceinline__ void order_emit_chains(const WgCtx& c, C d, int M, long long* stamp) {
    constexpr int NT = ONEWAVE ? 64 : WGT;
    const int tid = c.tid, max_size = c.max_size;
    auto bar = [&]() {
        if (ONEWAVE) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    };
    for (int m = tid; m < M; m += NT) {
        const int op = d.B[m], me = d.A[m];
        int pp = -1;
        if (op >= 0) { pp = lower_bound_i(d.A, M, op); if (pp >= M || d.A[pp] != op) pp = ORPHAN; }
        d.par[m] = pp; d.c0[m] = lower_bound_i(d.B, M, me); d.c1[m] = upper_bound_i(d.B, M, me);
    }
    const int top_end = lower_bound_i(d.dep, M, 2);                   // the run below the matched node
    bar();
    if (stamp && tid == 0) stamp[6] = wall_clock64();
    for (int m = tid; m < M; m += NT) {                               // rank
        const int p = d.par[m];
        int rank = 0;
        if (p != ORPHAN) {
            const int r0 = p < 0 ? 0 : d.c0[p], r1 = p < 0 ? top_end : d.c1[p];
            const double fmm = d.fm[m];
            for (int j = r0; j < r1; ++j) {
                const double fj = d.fm[j];
                rank += (fj > fmm) || (fj == fmm && j < m);
            }
        }
        d.A[m] = rank;
    }
    bar();
    for (int m = tid; m < M; m += NT) {                               // lb, ok, pruned
        int lb = 0;
        bool orphan = false;
        for (int x = m; x >= 0; x = d.par[x]) {
            if (d.par[x] == ORPHAN) { orphan = true; break; }
            lb += 1 + d.A[x];
        }
        const bool ok = !orphan && (lb - 1 - d.A[m]) < max_size;      // lb(parent) < max_size (0 for the top run)
        d.fl[m] = (unsigned char)(d.fl[m] | (ok ? (lb >= max_size ? (F_OK | F_PRUNED) : F_OK) : 0));
        d.B[m] = 0;
    }
    bar();
    for (int m = tid; m < M; m += NT) {                               // sizes
        const int f = d.fl[m];
        if (!(f & F_OK)) continue;
        const int wgt = (f & F_PRUNED) ? LA_WG_BIGC : 1;
        for (int x = m; x >= 0; x = d.par[x]) __hip_atomic_fetch_add(&d.B[x], wgt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    bar();
    for (int m = tid; m < M; m += NT) {                               // before
        if (!(d.fl[m] & F_OK)) continue;
        const int p = d.par[m];
        const int r0 = p < 0 ? 0 : d.c0[p], r1 = p < 0 ? top_end : d.c1[p];
        const double fmm = d.fm[m];
        int before = 0;
        for (int j = r0; j < r1; ++j) {
            const double fj = d.fm[j];
            if ((fj > fmm) || (fj == fmm && j < m)) before += d.B[j];
        }
        d.A[m] = before;
    }
    bar();
    for (int m = tid; m < M; m += NT) {                               // positions (c1 is free: the runs were last read above)
        if (!(d.fl[m] & F_OK)) continue;
        int pos = 0;
        for (int x = m; x >= 0; x = d.par[x]) pos += 1 + d.A[x];
        d.c1[m] = pos;
    }
    bar();
    int e_cnt = 0, e_fi = 0, e_fo = 0;
    const int W = c.W;
    for (int m = tid; m < M; m += NT) {                               // emit
        const int f = d.fl[m];
        if (!(f & F_OK)) continue;
        const int pos = d.c1[m];
        if (pos >= max_size) continue;
        c.oid[pos] = d.tok[m];
        unsigned long long m0 = 1ull, m1 = 0ull, m2 = 0ull, m3 = 0ull;
        for (int x = m; x >= 0; x = d.par[x]) {
            const int px = d.c1[x];
            const unsigned long long bit = 1ull << (px & 63);
            const int wd = px >> 6;
            m0 |= wd == 0 ? bit : 0ull; m1 |= wd == 1 ? bit : 0ull; m2 |= wd == 2 ? bit : 0ull; m3 |= wd == 3 ? bit : 0ull;
        }
        unsigned long long* o = c.orm + (size_t)pos * W;
        o[0] = m0;
        if (W > 1) o[1] = m1;
        if (W > 2) o[2] = m2;
        if (W > 3) o[3] = m3;
        ++e_cnt; e_fi += (f & F_FI) ? 1 : 0; e_fo += (f & F_FO) ? 1 : 0;
    }
    atomicAdd(&c.s_x[6], e_cnt); atomicAdd(&c.s_x[7], e_fi); atomicAdd(&c.s_x[8], e_fo);
}

__global__ __launch_bounds__(WGT) void k_trie_hier_get_wg(TrieWgArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pool[];
    __shared__ unsigned hist[256];
    __shared__ int s_w[2 * NWV];
    __shared__ int s_c[2 * 4 * NWV];
    __shared__ int s_lv[LA_WG_MAXLV + 3];
    __shared__ unsigned long long s_oa[2];
    __shared__ int s_x[12];          // 0 found | 1, 2 select | 3 rows, 4 fi > 0, 5 fo > 0 | 6 emitted, 7 sizes[0], 8 sizes[1]
    const int b = blockIdx.x, tid = threadIdx.x;
    WgCtx c;
    c.t = a.q.t;
    if (a.q.plane) c.t.fi += (size_t)a.q.plane[b] * (size_t)a.q.fi_stride;
    const TrieDev& t = c.t;
    const int branch_length = a.q.bl ? a.q.bl[b] : a.q.branch_length;
    const int* q = a.q.queries + b * 8;
    const int nq = a.q.nq[b];
    const int R = a.row_stride, W = a.mask_words;
    int* oid = a.q.out_ids + (size_t)b * R;
    unsigned long long* orm = a.q.out_rowmask + (size_t)b * R * W;
    const int max_size = a.q.decoding_length, max_length = branch_length, mode = a.q.mode;
    const int cap = t.n_nodes;
    int* gi = a.scr_i + (size_t)b * 16 * cap;
    double* gv = a.scr_v + (size_t)b * 3 * cap;
    c.tid = tid; c.lane = tid & 63; c.wv = tid >> 6;
    c.max_size = max_size; c.max_length = max_length; c.mode = mode;
    c.w = mode == LA_MODE_INPUT ? 0.0 : mode == LA_MODE_OUTPUT ? 1.0 : 1e-4;
    c.w1 = 1.0 - c.w;
    c.g_par = gi; c.g_tok = gi + cap; c.g_fl = gi + 2 * (size_t)cap; c.g_fi = gv; c.g_fo = gv + cap;
    c.s_w = s_w; c.s_c = s_c; c.s_lv = s_lv; c.s_x = s_x; c.oid = oid; c.orm = orm; c.W = W;
    long long* const stamp = a.q.dbg ? a.q.dbg + (size_t)b * 8 : nullptr;
    if (stamp && tid == 0) { stamp[0] = wall_clock64(); for (int k = 1; k < 8; ++k) stamp[k] = 0; }

    auto finish = [&](int n, int s0, int s1, int nsizes) {
        if (tid == 0) { a.q.out_n[b] = n; a.q.out_sizes[b * 2] = s0; a.q.out_sizes[b * 2 + 1] = s1; a.q.out_nsizes[b] = nsizes; }
    };
    auto one_row = [&](int token) {
        if (tid == 0) { oid[0] = token; orm[0] = 1ull; for (int x = 1; x < W; ++x) orm[x] = 0ull; }
    };
    if (a.q.decoding_length <= 1 || branch_length == 0) {                     // :413-414
        if (nq > 0) one_row(q[nq - 1]);
        finish(nq > 0 ? 1 : 0, 0, 0, 0);
        return;
    }
    bool have = false;
    int n_out = 0, sz0 = 0, sz1 = 0;
    for (int i = 0; i < nq; ++i) {
        int root;
        if (a.root_of) { const int tk = q[i]; root = (tk >= 0 && tk < a.n_root_of) ? a.root_of[tk] : -1; }
        else root = find_child_wg(t, 0, q[i], tid, &s_x[0]);
        if (root < 0) continue;
        const int nrest = nq - (i + 1);
        bool is_stop = false;
        for (int k = 0; k < a.q.n_stop; ++k) is_stop |= (a.q.stop[k] == q[i]);
        if (is_stop && nrest == 0) continue;                                  // :422-423
        have = true;
        // ---- S1: Tree._match
        int cur = root;
        for (int k = 0; k < nrest && cur >= 0; ++k) {
            const int ch = find_child_wg(t, cur, q[i + 1 + k], tid, &s_x[0]);
            if (ch < 0) { cur = -1; break; }
            const double cfi = t.fi[ch], cfo = t.fo[ch];
            const bool live = mode == LA_MODE_INPUT ? cfi > 0 : mode == LA_MODE_OUTPUT ? cfo > 0 : (cfi > 0 || cfo > 0);
            cur = live ? ch : -1;
        }
        sz0 = sz1 = 0;
        if (stamp && tid == 0) stamp[1] = wall_clock64();
        const int cc_cur = cur >= 0 ? t.ccount[cur] : 0;
        if (cur < 0 || cc_cur == 0) {                                         // :70-72
            one_row(nrest > 0 ? q[nq - 1] : t.tok[root]);
            n_out = 1;
        } else {
            // ---- S2: level-synchronous expansion
            const lds_ptr<int2> l_lv0 = (lds_ptr<int2>)pool; const lds_ptr<int2> l_lv1 = l_lv0 + LA_WG_LCAP;
            int2* const g_lv0 = (int2*)(gi + 4 * (size_t)cap); int2* const g_lv1 = (int2*)(gi + 6 * (size_t)cap);
            __syncthreads();                                                  // the pool may still be read by the previous suffix's emit
            if (tid == 0) { l_lv0[0].x = 0; l_lv0[0].y = (int)((unsigned)t.cstart[cur] | 0x80000000u); }
            if (tid < 12 && tid >= 3) s_x[tid] = 0;
            __syncthreads();
            int np = 1, ps = -1, ebase = 0, total = cc_cur, depth = 1;
            int c_rows = 0, c_fi = 0, c_fo = 0;
            bool bad = false, p_lds = true;
            while (total > 0) {
                if (depth > LA_WG_MAXLV || ebase + total > cap) { bad = true; break; }
                const bool q_lds = total <= a.lcap, odd = depth & 1;
                int ntot;
                if (p_lds && q_lds) ntot = expand_level(c, odd ? l_lv0 : l_lv1, np, ps, odd ? l_lv1 : l_lv0, ebase, total, depth, c_rows, c_fi, c_fo);
                else if (p_lds)     ntot = expand_level(c, odd ? l_lv0 : l_lv1, np, ps, odd ? g_lv1 : g_lv0, ebase, total, depth, c_rows, c_fi, c_fo);
                else if (q_lds)     ntot = expand_level(c, odd ? g_lv0 : g_lv1, np, ps, odd ? l_lv1 : l_lv0, ebase, total, depth, c_rows, c_fi, c_fo);
                else                ntot = expand_level(c, odd ? g_lv0 : g_lv1, np, ps, odd ? g_lv1 : g_lv0, ebase, total, depth, c_rows, c_fi, c_fo);
                p_lds = q_lds; np = total; ps = ebase; ebase += total; total = ntot; ++depth;
            }
            const int n = ebase;
            atomicAdd(&s_x[3], c_rows); atomicAdd(&s_x[4], c_fi); atomicAdd(&s_x[5], c_fo);
            __syncthreads();
            const int rows = s_x[3], cnt_fi = s_x[4], cnt_fo = s_x[5];
            if (stamp && tid == 0) { stamp[2] = wall_clock64(); stamp[5] = rows; }
            if (bad) {                                                        // deeper than LA_WG_MAXLV levels / a corrupt image: the 1-row answer, flagged
                one_row(nrest > 0 ? q[nq - 1] : t.tok[root]);
                finish(1, 0, 0, -1);
                return;
            }
            // ---- S3: cut-offs
            double lo_in = WTBIG, lo_out = WTBIG, lo_mix = WTBIG;
            const lds_ptr<double> sv = (lds_ptr<double>)pool;
            if (mode == LA_MODE_INPUT) {
                lo_in = cnt_fi > max_size ? select_desc_wg(c.g_fi, c.g_fl, n, a.q.min_in <= 0 ? rows - 1 : min(a.q.min_in - 1, rows - 1), tid, hist, sv, &s_x[1], s_oa) : 0.0;
            } else if (mode == LA_MODE_OUTPUT) {
                lo_out = cnt_fo > max_size ? select_desc_wg(c.g_fo, c.g_fl, n, a.q.min_out <= 0 ? rows - 1 : min(a.q.min_out - 1, rows - 1), tid, hist, sv, &s_x[1], s_oa) : 0.0;
            } else if (rows > max_size) {
                // rows carry None as their index (:152): the mix cut-off loop never fires, lo_mix stays 1e9
                if (a.q.min_in > 0) lo_in = select_desc_wg(c.g_fi, c.g_fl, n, min(a.q.min_in - 1, rows - 1), tid, hist, sv, &s_x[1], s_oa);
                if (a.q.min_out > 0) lo_out = select_desc_wg(c.g_fo, c.g_fl, n, min(a.q.min_out - 1, rows - 1), tid, hist, sv, &s_x[1], s_oa);
            } else {
                lo_mix = 0.0;
            }
            if (stamp && tid == 0) stamp[3] = wall_clock64();
            // ---- C, S4 .. S7: in LDS — or over the scratch when the candidates do not fit
            const int last_tok = nrest > 0 ? q[nq - 1] : 0;
            if (tid == 0) {
                oid[0] = (nrest > 0 && last_tok != 0) ? last_tok : t.tok[root];   // :129
                orm[0] = 1ull; for (int x = 1; x < W; ++x) orm[x] = 0ull;
            }
            Cand<lds_ptr<int>, lds_ptr<double>, lds_ptr<unsigned char>> dl;
            {
                const lds_ptr<int> base = (lds_ptr<int>)pool;
                dl.par = base; dl.c0 = base + LA_WG_MCAP; dl.c1 = base + 2 * LA_WG_MCAP; dl.A = base + 3 * LA_WG_MCAP; dl.B = base + 4 * LA_WG_MCAP;
                dl.tok = base + 5 * LA_WG_MCAP; dl.fm = (lds_ptr<double>)(base + 6 * LA_WG_MCAP); dl.fl = (lds_ptr<unsigned char>)(dl.fm + LA_WG_MCAP);
                dl.dep = dl.fl + LA_WG_MCAP;
            }
            int M = compact_cands(c, dl, a.mcap, n, lo_in, lo_out, lo_mix);
            if (M <= a.onewave) {
                if (c.wv == 0) order_emit_chains<true>(c, dl, M, stamp);
            } else if (M <= a.mcap) {
                order_emit_chains<false>(c, dl, M, stamp);
            } else {
                Cand<int*, double*, unsigned char*> dg;
                dg.par = gi + 8 * (size_t)cap; dg.c0 = gi + 9 * (size_t)cap; dg.c1 = gi + 10 * (size_t)cap; dg.A = gi + 11 * (size_t)cap;
                dg.B = gi + 12 * (size_t)cap; dg.tok = gi + 13 * (size_t)cap; dg.fm = gv + 2 * (size_t)cap;
                dg.fl = (unsigned char*)(gi + 14 * (size_t)cap); dg.dep = (unsigned char*)(gi + 15 * (size_t)cap);
                M = compact_cands(c, dg, cap, n, lo_in, lo_out, lo_mix);
                order_emit<false>(c, dg, M, stamp);
            }
            __syncthreads();
            const int emitted = s_x[6];
            n_out = 1 + emitted; sz0 = s_x[7]; sz1 = s_x[8];
            if (stamp && tid == 0) { stamp[4] = wall_clock64(); stamp[5] |= (long long)n_out << 32; stamp[7] = ((long long)n << 32) | (long long)M; }
        }
        if (n_out >= branch_length) break;                                    // :433-434 (else a later suffix overwrites)
    }
    if (!have) {                                                              // :436-437
        if (nq > 0) one_row(q[nq - 1]);
        finish(nq > 0 ? 1 : 0, 0, 0, 2);
        return;
    }
    finish(n_out, sz0, sz1, 2);
}

int lk_trie_hier_get_wg(hipStream_t st, const TrieWgArgs& a, int B) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_trie_hier_get_wg, hipFuncAttributeMaxDynamicSharedMemorySize, LA_WG_POOL) != hipSuccess) return -1;
        attr_set = true;
    }
    TrieWgArgs x = a;
    x.q.dbg = g_la_dbg_times;
    x.lcap = a.lcap > 0 ? min(a.lcap, LA_WG_LCAP) : LA_WG_LCAP;          // 0 = the library's limits; smaller values (tests) send small sets
    x.mcap = a.mcap > 0 ? min(a.mcap, LA_WG_MCAP) : LA_WG_MCAP;          // down the global-scratch paths
    x.onewave = a.onewave > 0 ? min(a.onewave, x.mcap) : (a.onewave < 0 ? 0 : min(LA_WG_ONEWAVE, x.mcap));
    k_trie_hier_get_wg<<<B, WGT, LA_WG_POOL, st>>>(x);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
