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
// Sets that outgrow LDS (a level above LA_WG_LCAP entries, more than LA_WG_MCAP candidates) run the same code over the global scratch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "la_kernels.h"
#include "la_trie_dev.h"
extern long long* g_la_dbg_times;

#define WGT 256
#define NWV 4
#define LA_WG_LCAP 4096          // entries of one level whose run starts live in LDS
#define LA_WG_MCAP 3072          // candidate entries that live in LDS
#define LA_WG_SEL 4096           // values the radix select stages in LDS
#define LA_WG_MAXLV 128          // deepest level followed (a put inserts branch_length + 1 tokens)
#define LA_WG_POOL (LA_WG_MCAP * 34)
#define WBIG (1 << 20)
#define WTBIG 1e9
#define ORPHAN (-2)
#define F_FI 1
#define F_FO 2
#define F_OK 4
#define F_PRUNED 8
#define G_INF 1                  // g_fl: bit 0 = inF, bits 8.. = depth

static_assert(2 * LA_WG_LCAP * 8 <= LA_WG_POOL && LA_WG_SEL * 8 <= LA_WG_POOL, "the S2 / S3 buffers alias the candidate arrays");

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int x = __shfl_up(v, o, 64);
        if (lane >= o) v += x;
    }
    return v;
}

// a[i * stride] (i < cnt) <- base + exclusive prefix of the old values; -> their sum (uniform).  One barrier per 256 items; the caller
// puts a barrier behind the call before other threads' results are read.
__device__ int scan_counts(int* a, int stride, int cnt, int base, int tid, int* s_w) {
    const int lane = tid & 63, wv = tid >> 6;
    int carry = 0, it = 0;
    for (int c0 = 0; c0 < cnt; c0 += WGT, ++it) {
        const int i = c0 + tid;
        const int v = i < cnt ? a[(size_t)i * stride] : 0;
        const int inc = wave_incl_scan(v, lane);
        int* sw = s_w + (it & 1) * NWV;
        if (lane == 63) sw[wv] = inc;
        __syncthreads();
        int pre = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { const int x = sw[w]; pre += w < wv ? x : 0; tot += x; }
        if (i < cnt) a[(size_t)i * stride] = base + carry + pre + inc - v;
        carry += tot;
    }
    return carry;
}

// value at position r (0-based) of the DESCENDING sort of {inF(i) ? vals[i] : 0} (all >= 0), i < n: radix select on the bit patterns
__device__ double select_desc_wg(const double* vals, const int* fl, int n, int r, int tid, unsigned* hist, double* sv, int* s_sel) {
    const bool staged = n <= LA_WG_SEL;
    __syncthreads();                                                        // the pool's previous tenants are done
    if (staged) {
        for (int i = tid; i < n; i += WGT) sv[i] = (fl[i] & G_INF) ? vals[i] : 0.0;
    }
    unsigned long long prefix = 0ull, mask = 0ull;
    int rank = r;
    for (int byte = 7; byte >= 0; --byte) {
        hist[tid] = 0u;
        __syncthreads();
        const int sh = byte * 8;
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

// first index in [0, n) with a[i] >= key (a non-decreasing)
template <typename T>
__device__ __forceinline__ int lower_bound_i(const T* a, int n, int key) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
template <typename T>
__device__ __forceinline__ int upper_bound_i(const T* a, int n, int key) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)a[mid] <= key) lo = mid + 1; else hi = mid; }
    return lo;
}

__global__ __launch_bounds__(WGT) void k_trie_hier_get_wg(TrieWgArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pool[];
    __shared__ unsigned hist[256];
    __shared__ int s_w[2 * NWV];
    __shared__ int s_lv[LA_WG_MAXLV + 3];
    __shared__ int s_x[12];          // 0 found | 1, 2 select | 3 rows, 4 fi > 0, 5 fo > 0 | 6 emitted, 7 sizes[0], 8 sizes[1]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    TrieDev t = a.q.t;
    if (a.q.plane) t.fi += (size_t)a.q.plane[b] * (size_t)a.q.fi_stride;
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
    int* const g_par = gi; int* const g_tok = gi + cap; int* const g_fl = gi + 2 * (size_t)cap;
    double* const g_fi = gv; double* const g_fo = gv + cap;
    long long* const stamp = a.q.dbg ? a.q.dbg + (size_t)b * 8 : nullptr;
    if (stamp && tid == 0) { stamp[0] = wall_clock64(); for (int k = 1; k < 8; ++k) stamp[k] = 0; }

    auto finish = [&](int n, int s0, int s1, int nsizes) {
        if (tid == 0) { a.q.out_n[b] = n; a.q.out_sizes[b * 2] = s0; a.q.out_sizes[b * 2 + 1] = s1; a.q.out_nsizes[b] = nsizes; }
    };
    auto one_row = [&](int token) {
        if (tid == 0) { oid[0] = token; orm[0] = 1ull; for (int w = 1; w < W; ++w) orm[w] = 0ull; }
    };
    if (a.q.decoding_length <= 1 || branch_length == 0) {                     // :413-414
        if (nq > 0) one_row(q[nq - 1]);
        finish(nq > 0 ? 1 : 0, 0, 0, 0);
        return;
    }
    double w = 1e-4;
    if (mode == LA_MODE_INPUT) w = 0.0; else if (mode == LA_MODE_OUTPUT) w = 1.0;
    const double w1 = 1.0 - w;
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
            int2* const l_lv0 = (int2*)pool; int2* const l_lv1 = l_lv0 + LA_WG_LCAP;
            int2* const g_lv0 = (int2*)(gi + 4 * (size_t)cap); int2* const g_lv1 = (int2*)(gi + 6 * (size_t)cap);
            __syncthreads();                                                  // the pool may still be read by the previous suffix's emit
            if (tid == 0) { l_lv0[0] = make_int2(0, (int)((unsigned)t.cstart[cur] | 0x80000000u)); }
            if (tid < 12 && tid >= 3) s_x[tid] = 0;
            __syncthreads();
            int2* P = l_lv0;
            int np = 1, ps = -1, ebase = 0, total = cc_cur, depth = 1;
            int c_rows = 0, c_fi = 0, c_fo = 0;
            bool bad = false;
            while (total > 0) {
                if (depth > LA_WG_MAXLV || ebase + total > cap) { bad = true; break; }
                int2* const Q = total <= LA_WG_LCAP ? ((depth & 1) ? l_lv1 : l_lv0) : ((depth & 1) ? g_lv1 : g_lv0);
                for (int j = tid; j < total; j += WGT) {
                    const int e = ebase + j;
                    int lo = 0, hi = np;                                      // the parent: last p with P[p].x <= e
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (P[mid].x <= e) lo = mid; else hi = mid; }
                    const int2 pr = P[lo];
                    const int c = (int)((unsigned)pr.y & 0x7fffffffu) + (e - pr.x);
                    const double cfi = t.fi[c], cfo = t.fo[c];
                    const int ccs = t.cstart[c], ccc = t.ccount[c], ctok = t.tok[c];
                    const bool inF = (cfi > 0 || cfo > 0) && pr.y < 0;
                    const bool expand = ccc > 0 && (inF || depth < max_length);
                    g_par[e] = ps + lo; g_tok[e] = ctok; g_fi[e] = cfi; g_fo[e] = cfo; g_fl[e] = (depth << 8) | (inF ? G_INF : 0);
                    Q[j] = make_int2(expand ? ccc : 0, (int)((unsigned)ccs | (inF ? 0x80000000u : 0u)));
                    c_rows += inF; c_fi += inF && cfi > 0; c_fo += inF && cfo > 0;
                }
                __syncthreads();
                const int ntot = scan_counts(&Q[0].x, 2, total, ebase + total, tid, s_w);
                __syncthreads();
                P = Q; np = total; ps = ebase; ebase += total; total = ntot; ++depth;
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
            double* const sv = (double*)pool;
            if (mode == LA_MODE_INPUT) {
                lo_in = cnt_fi > max_size ? select_desc_wg(g_fi, g_fl, n, a.q.min_in <= 0 ? rows - 1 : min(a.q.min_in - 1, rows - 1), tid, hist, sv, &s_x[1]) : 0.0;
            } else if (mode == LA_MODE_OUTPUT) {
                lo_out = cnt_fo > max_size ? select_desc_wg(g_fo, g_fl, n, a.q.min_out <= 0 ? rows - 1 : min(a.q.min_out - 1, rows - 1), tid, hist, sv, &s_x[1]) : 0.0;
            } else if (rows > max_size) {
                // rows carry None as their index (:152): the mix cut-off loop never fires, lo_mix stays 1e9
                if (a.q.min_in > 0) lo_in = select_desc_wg(g_fi, g_fl, n, min(a.q.min_in - 1, rows - 1), tid, hist, sv, &s_x[1]);
                if (a.q.min_out > 0) lo_out = select_desc_wg(g_fo, g_fl, n, min(a.q.min_out - 1, rows - 1), tid, hist, sv, &s_x[1]);
            } else {
                lo_mix = 0.0;
            }
            if (stamp && tid == 0) stamp[3] = wall_clock64();
            // ---- C: candidates, compacted (stable) into LDS — or into the scratch when they do not fit
            int* d_par; int* d_c0; int* d_c1; int* d_A; int* d_B; int* d_tok; double* d_fm; unsigned char* d_fl; unsigned char* d_dep;
            int M = 0;
            for (int attempt = 0; attempt < 2; ++attempt) {
                int dcap;
                if (attempt == 0) {
                    int* base = (int*)pool;
                    d_par = base; d_c0 = base + LA_WG_MCAP; d_c1 = base + 2 * LA_WG_MCAP; d_A = base + 3 * LA_WG_MCAP; d_B = base + 4 * LA_WG_MCAP;
                    d_tok = base + 5 * LA_WG_MCAP; d_fm = (double*)(base + 6 * LA_WG_MCAP); d_fl = (unsigned char*)(d_fm + LA_WG_MCAP);
                    d_dep = d_fl + LA_WG_MCAP; dcap = LA_WG_MCAP;
                } else {
                    d_par = gi + 8 * (size_t)cap; d_c0 = gi + 9 * (size_t)cap; d_c1 = gi + 10 * (size_t)cap; d_A = gi + 11 * (size_t)cap;
                    d_B = gi + 12 * (size_t)cap; d_tok = gi + 13 * (size_t)cap; d_fm = gv + 2 * (size_t)cap;
                    d_fl = (unsigned char*)(gi + 14 * (size_t)cap); d_dep = (unsigned char*)(gi + 15 * (size_t)cap); dcap = cap;
                }
                __syncthreads();
                M = 0;
                int it = 0;
                for (int c0 = 0; c0 < n; c0 += WGT, ++it) {
                    const int e = c0 + tid;
                    bool cand = false;
                    double cfi = 0.0, cfo = 0.0, fm = 0.0;
                    int dep = 0;
                    if (e < n) {
                        dep = g_fl[e] >> 8;
                        cfi = g_fi[e]; cfo = g_fo[e];
                        fm = __dadd_rn(__dmul_rn(w1, cfi), __dmul_rn(w, cfo));                    // :254, no FMA
                        bool skip;
                        if (mode == LA_MODE_MIX) skip = cfi < lo_in && cfo < lo_out && fm < lo_mix;   // :265
                        else if (mode == LA_MODE_INPUT) skip = cfi < lo_in;
                        else skip = cfo < lo_out;
                        cand = dep <= max_length && !skip;
                    }
                    const unsigned long long bm = __ballot(cand);
                    int* sw = s_w + (it & 1) * NWV;
                    if (lane == 0) sw[wv] = __popcll(bm);
                    __syncthreads();
                    int pre = 0, tot = 0;
#pragma unroll
                    for (int x = 0; x < NWV; ++x) { const int v = sw[x]; pre += x < wv ? v : 0; tot += v; }
                    const int m = M + pre + __popcll(bm & ((1ull << lane) - 1ull));
                    if (cand && m < dcap) {
                        d_A[m] = e; d_B[m] = g_par[e]; d_tok[m] = g_tok[e]; d_fm[m] = fm;
                        d_fl[m] = (unsigned char)((cfi > 0 ? F_FI : 0) | (cfo > 0 ? F_FO : 0)); d_dep[m] = (unsigned char)dep;
                    }
                    M += tot;
                }
                if (M <= dcap) break;
            }
            __syncthreads();
            const int Dl = min(max_length, LA_WG_MAXLV);
            // parents, child runs and level starts in the compacted numbering (old index / old parent are non-decreasing)
            for (int m = tid; m < M; m += WGT) {
                const int op = d_B[m], me = d_A[m];
                int pp = -1;
                if (op >= 0) { pp = lower_bound_i(d_A, M, op); if (pp >= M || d_A[pp] != op) pp = ORPHAN; }
                d_par[m] = pp; d_c0[m] = lower_bound_i(d_B, M, me); d_c1[m] = upper_bound_i(d_B, M, me);
            }
            if (tid >= 1 && tid <= Dl + 1) s_lv[tid] = lower_bound_i(d_dep, M, tid);
            __syncthreads();
            int Dn = 0;                                                       // levels that hold candidates
            for (int d = 1; d <= Dl; ++d) { if (s_lv[d] < s_lv[d + 1]) Dn = d; else break; }
            // ---- S4: ok / rank / lower bound / prune, top-down
            for (int d = 1; d <= Dn; ++d) {
                const int ls = s_lv[d], le = s_lv[d + 1];
                for (int m = ls + tid; m < le; m += WGT) {
                    const int p = d_par[m];
                    const bool ok = p != ORPHAN && (p < 0 || ((d_fl[p] & (F_OK | F_PRUNED)) == F_OK));
                    if (ok) d_fl[m] |= F_OK;
                }
                __syncthreads();
                for (int m = ls + tid; m < le; m += WGT) {
                    if (!(d_fl[m] & F_OK)) continue;
                    const int p = d_par[m];
                    const int r0 = p < 0 ? 0 : d_c0[p], r1 = p < 0 ? s_lv[2] : d_c1[p];
                    const int lbp = p < 0 ? 0 : d_B[p];
                    const int room = max_size - lbp - 1;                      // rank >= room: lb >= max_size
                    const double fmm = d_fm[m];
                    int rank = 0;
                    for (int j = r0; j < r1 && rank < room; ++j) {
                        if (j == m || !(d_fl[j] & F_OK)) continue;
                        const double fj = d_fm[j];
                        rank += (fj > fmm) || (fj == fmm && j < m);
                    }
                    const int lb = lbp + 1 + rank;
                    d_B[m] = lb;
                    if (lb >= max_size) d_fl[m] |= F_PRUNED;
                }
                __syncthreads();
            }
            // ---- S5: subtree sizes, bottom-up
            for (int d = Dn; d >= 1; --d) {
                const int ls = s_lv[d], le = s_lv[d + 1];
                for (int m = ls + tid; m < le; m += WGT) {
                    const int f = d_fl[m];
                    if (!(f & F_OK)) continue;
                    int sz = WBIG;
                    if (!(f & F_PRUNED)) {
                        sz = 1;
                        const int r0 = d_c0[m], r1 = d_c1[m];
                        for (int j = r0; j < r1 && sz < WBIG; ++j) if (d_fl[j] & F_OK) sz += d_A[j];
                        sz = min(sz, WBIG);
                    }
                    d_A[m] = sz;
                }
                __syncthreads();
            }
            // ---- S6: preorder positions, top-down
            for (int d = 1; d <= Dn; ++d) {
                const int ls = s_lv[d], le = s_lv[d + 1];
                for (int m = ls + tid; m < le; m += WGT) {
                    if (!(d_fl[m] & F_OK)) continue;
                    const int p = d_par[m];
                    const int pp = p < 0 ? 0 : d_B[p];
                    int pos = WBIG;
                    if (pp < max_size - 1) {
                        const int r0 = p < 0 ? 0 : d_c0[p], r1 = p < 0 ? s_lv[2] : d_c1[p];
                        const int room = max_size - pp - 1;                   // before >= room: pos >= max_size
                        const double fmm = d_fm[m];
                        int before = 0;
                        for (int j = r0; j < r1 && before < room; ++j) {
                            if (j == m || !(d_fl[j] & F_OK)) continue;
                            const double fj = d_fm[j];
                            if ((fj > fmm) || (fj == fmm && j < m)) before += d_A[j];
                        }
                        pos = min(WBIG, pp + 1 + before);
                    }
                    d_B[m] = pos;
                }
                __syncthreads();
            }
            // ---- S7: emit
            const int last_tok = nrest > 0 ? q[nq - 1] : 0;
            if (tid == 0) {
                oid[0] = (nrest > 0 && last_tok != 0) ? last_tok : t.tok[root];   // :129
                orm[0] = 1ull; for (int x = 1; x < W; ++x) orm[x] = 0ull;
            }
            int e_cnt = 0, e_fi = 0, e_fo = 0;
            const int Mend = Dn > 0 ? s_lv[Dn + 1] : 0;
            for (int m = tid; m < Mend; m += WGT) {
                const int f = d_fl[m];
                if (!(f & F_OK)) continue;
                const int pos = d_B[m];
                if (pos >= max_size) continue;
                oid[pos] = d_tok[m];
                unsigned long long m0 = 1ull, m1 = 0ull, m2 = 0ull, m3 = 0ull;
                for (int x = m; x >= 0; x = d_par[x]) {
                    const int px = d_B[x];
                    const unsigned long long bit = 1ull << (px & 63);
                    const int wd = px >> 6;
                    m0 |= wd == 0 ? bit : 0ull; m1 |= wd == 1 ? bit : 0ull; m2 |= wd == 2 ? bit : 0ull; m3 |= wd == 3 ? bit : 0ull;
                }
                unsigned long long* o = orm + (size_t)pos * W;
                o[0] = m0;
                if (W > 1) o[1] = m1;
                if (W > 2) o[2] = m2;
                if (W > 3) o[3] = m3;
                ++e_cnt; e_fi += (f & F_FI) ? 1 : 0; e_fo += (f & F_FO) ? 1 : 0;
            }
            atomicAdd(&s_x[6], e_cnt); atomicAdd(&s_x[7], e_fi); atomicAdd(&s_x[8], e_fo);
            __syncthreads();
            n_out = 1 + s_x[6]; sz0 = s_x[7]; sz1 = s_x[8];
            if (stamp && tid == 0) { stamp[4] = wall_clock64(); stamp[6] = n_out; stamp[7] = ((long long)n << 32) | (long long)M; }
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
    k_trie_hier_get_wg<<<B, WGT, LA_WG_POOL, st>>>(x);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
