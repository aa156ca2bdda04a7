// la_trie_dev.hip — device-side trie retrieval: LookaheadCache.hier_get (lookahead_cache.py:408-439) with
// Tree.get/_match/_dfs_get_freqs/_ravel (:65-154, 224-293) as ONE WAVEFRONT PER QUERY over a mirrored arena
// (la_cache_export: breadth-first ids, children of a node = consecutive ids in dict insertion order).
//   prefix match      : 64 children compared per step, __ballot picks the hit            (:224-246)
//   live-subtree scan : wave-parallel frontier expansion, ballot prefix sums as queue     (:146-154)
//   cut-offs          : radix select of the k-th largest fi / fo (bit patterns of non-negative doubles)  (:78-125)
//   ordered DFS       : "next child in stable (fm desc, insertion asc) order" by a wave arg-max over the sibling
//                       range, explicit stack (depth <= branch_length), <= 64 emitted rows with 64-bit row masks  (:248-293)
// Bit-exact to the host trie / the reference: fm uses separately rounded fp64 multiplies and add (no FMA contraction).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "la_kernels.h"

#include "la_trie_dev.h"
#include "la_knobs.h"

#define TBIG 1e9
#define LA_SCAN_SMALL 6

__device__ __forceinline__ int wave_first(unsigned long long m) { return m ? __ffsll((long long)m) - 1 : -1; }

// child of node u with token t, or -1 (wave-uniform result)
__device__ int find_child(const TrieDev& t, int u, int token, int lane) {
    const int cs = t.cstart[u], cc = t.ccount[u];
    for (int base = 0; base < cc; base += 64) {
        const int i = base + lane;
        const bool hit = i < cc && t.tok[cs + i] == token;
        const unsigned long long m = __ballot(hit);
        if (m) return cs + base + wave_first(m);
    }
    return -1;
}

// value at position r (0-based) of the DESCENDING sort of vals[0..n) (all >= 0): radix select on the bit patterns.
// Round 3 (rocprof + phase stamps, profiles/r03_trie_device_profile.txt: 141 us median for ~1750 values): the values are staged ONCE
// in LDS (n <= LA_SEL_LDS; larger sets stream from the global scratch as before) instead of being re-read from global memory in
// each of the 8 byte passes, and the bin in which the rank falls is found by a wave-parallel suffix sum over the 256 bins
// (4 bins per lane, 6 shuffle steps) instead of a serial walk of up to 255 dependent LDS reads per pass.
#define LA_SEL_LDS 4096
__device__ double select_desc(const double* vals, int n, int r, int lane, unsigned* hist /*LDS[256]*/, double* sv /*LDS[LA_SEL_LDS]*/) {
    const bool staged = n <= LA_SEL_LDS;
    if (staged) {
        for (int i = lane; i < n; i += 64) sv[i] = vals[i];
        __syncthreads();
    }
    unsigned long long prefix = 0ull, mask = 0ull;
    int rank = r;
    for (int byte = 7; byte >= 0; --byte) {
        for (int i = lane; i < 256; i += 64) hist[i] = 0u;
        __syncthreads();
        const int sh = byte * 8;
        for (int i = lane; i < n; i += 64) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(staged ? sv[i] : vals[i]);
            if ((b & mask) == prefix) atomicAdd(&hist[(unsigned)((b >> sh) & 0xffull)], 1u);
        }
        __syncthreads();
        // lane l owns bins 255 - 4l .. 252 - 4l (descending): above = values in higher bins = exclusive prefix over lanes
        const int top = 255 - 4 * lane;
        const int h0 = (int)hist[top], h1 = (int)hist[top - 1], h2 = (int)hist[top - 2], h3 = (int)hist[top - 3];
        const int mine = h0 + h1 + h2 + h3;
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        const int above = incl - mine;
        const bool here = rank >= above && rank < incl;          // exactly one lane (rank < n = total)
        int bin = 0, nrank = 0;
        if (here) {
            int acc = above;
            bin = top;
            if (acc + h0 <= rank) { acc += h0; bin = top - 1;
                if (acc + h1 <= rank) { acc += h1; bin = top - 2;
                    if (acc + h2 <= rank) { acc += h2; bin = top - 3; } } }
            nrank = rank - acc;
        }
        const unsigned long long hm = __ballot(here);
        const int src = hm ? __ffsll((long long)hm) - 1 : 63;     // (empty set: n == 0 is never passed)
        bin = __shfl(bin, src, 64);
        rank = __shfl(nrank, src, 64);
        prefix |= (unsigned long long)bin << sh;
        mask |= 0xffull << sh;
        __syncthreads();
    }
    return __longlong_as_double((long long)prefix);
}

__global__ __launch_bounds__(64) void k_trie_hier_get(TrieQueryArgs a) {
    __shared__ unsigned hist[256];
    __shared__ __attribute__((aligned(16))) double sel_vals[LA_SEL_LDS];
    __shared__ int st_pid[72], st_depth[72], fr_cnt[72], fr_next[72];
    __shared__ int fr_child[72][64];
    __shared__ unsigned char fr_flag[72][64];
    __shared__ double tmp_fm[64];
    __shared__ int tmp_k[64], tmp_fl[64];
    const int b = blockIdx.x, lane = threadIdx.x;
    TrieDev t = a.t;
    if (a.plane) t.fi += (size_t)a.plane[b] * (size_t)a.fi_stride;
    const int branch_length = a.bl ? a.bl[b] : a.branch_length;
    const int* q = a.queries + b * 8;
    const int nq = a.nq[b];
    int* oid = a.out_ids + b * 64;
    unsigned long long* orm = a.out_rowmask + b * 64;
    int* queue = a.scratch_q + (size_t)b * t.n_nodes;
    double* vfi = a.scratch_v + (size_t)b * 2 * t.n_nodes;
    double* vfo = vfi + t.n_nodes;
    const int max_size = a.decoding_length, max_length = branch_length, mode = a.mode;
    long long* const stamp = a.dbg ? a.dbg + (size_t)b * 8 : nullptr;
    if (stamp && lane == 0) { stamp[0] = wall_clock64(); stamp[1] = stamp[2] = stamp[3] = stamp[4] = 0; stamp[5] = stamp[6] = 0; }

    auto finish = [&](int n, int s0, int s1, int nsizes) {
        if (lane == 0) { a.out_n[b] = n; a.out_sizes[b * 2] = s0; a.out_sizes[b * 2 + 1] = s1; a.out_nsizes[b] = nsizes; }
    };
    if (a.decoding_length <= 1 || branch_length == 0) {                       // :413-414
        if (nq > 0 && lane == 0) { oid[0] = q[nq - 1]; orm[0] = 1ull; }
        finish(nq > 0 ? 1 : 0, 0, 0, 0);
        return;
    }
    bool have = false;
    int n_out = 0, sz0 = 0, sz1 = 0;
    for (int i = 0; i < nq; ++i) {
        const int root = find_child(t, 0, q[i], lane);
        if (root < 0) continue;
        const int nrest = nq - (i + 1);
        bool is_stop = false;
        for (int k = 0; k < a.n_stop; ++k) is_stop |= (a.stop[k] == q[i]);
        if (is_stop && nrest == 0) continue;                                  // :422-423
        have = true;
        // ---- Tree._match
        int cur = root;
        for (int k = 0; k < nrest && cur >= 0; ++k) {
            const int ch = find_child(t, cur, q[i + 1 + k], lane);
            if (ch < 0) { cur = -1; break; }
            const double cfi = t.fi[ch], cfo = t.fo[ch];
            const bool live = mode == LA_MODE_INPUT ? cfi > 0 : mode == LA_MODE_OUTPUT ? cfo > 0 : (cfi > 0 || cfo > 0);
            cur = live ? ch : -1;
        }
        sz0 = sz1 = 0;
        if (stamp && lane == 0) stamp[1] = wall_clock64();
        if (cur < 0 || t.ccount[cur] == 0) {                                  // :70-72
            if (lane == 0) { oid[0] = nrest > 0 ? q[nq - 1] : t.tok[root]; orm[0] = 1ull; }
            n_out = 1;
        } else {
            // ---- _dfs_get_freqs: rows of live nodes reachable through live nodes
            int head = 0, tail = 0;
            {   // seed with the live children of cur
                const int cs = t.cstart[cur], cc = t.ccount[cur];
                for (int base = 0; base < cc; base += 64) {
                    const int c = cs + base + lane;
                    const bool lv = base + lane < cc && (t.fo[c] > 0 || t.fi[c] > 0);
                    const unsigned long long m = __ballot(lv);
                    if (lv) queue[tail + __popcll(m & ((1ull << lane) - 1ull))] = c;
                    tail += __popcll(m);
                }
                __threadfence_block();      // queue entries written by other lanes are read below
            }
            while (head < tail) {
                const int v = head + lane < tail ? queue[head + lane] : -1;
                int cs = 0, cc = 0;
                if (v >= 0) { vfi[head + lane] = t.fi[v]; vfo[head + lane] = t.fo[v]; cs = t.cstart[v]; cc = t.ccount[v]; }
                // children of the 64 nodes of this batch.  Low-degree nodes (the bulk of a trie): every lane walks its own node's
                // children, at most LA_SCAN_SMALL dependent steps; a high-degree node (the matched node and its first levels carry
                // hundreds of children) is expanded by the whole wave, 64 consecutive children per step (coalesced fi / fo reads) —
                // round 2 ran max(children) steps for the whole batch.  The queue order only feeds counts and the cut-off select.
                const int nproc = min(64, tail - head);
                head += nproc;
                const int ccs = cc <= LA_SCAN_SMALL ? cc : 0;
                int mx = ccs;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
                for (int k = 0; k < mx; ++k) {
                    const int c = cs + k;
                    const bool lv = k < ccs && (t.fo[c] > 0 || t.fi[c] > 0);
                    const unsigned long long m = __ballot(lv);
                    if (lv) queue[tail + __popcll(m & ((1ull << lane) - 1ull))] = c;
                    tail += __popcll(m);
                }
                for (unsigned long long big = __ballot(cc > LA_SCAN_SMALL); big; big &= big - 1ull) {
                    const int j = __ffsll((long long)big) - 1;
                    const int bcs = __shfl(cs, j, 64), bcc = __shfl(cc, j, 64);
                    for (int base = 0; base < bcc; base += 64) {
                        const int c = bcs + base + lane;
                        const bool lv = base + lane < bcc && (t.fo[c] > 0 || t.fi[c] > 0);
                        const unsigned long long m = __ballot(lv);
                        if (lv) queue[tail + __popcll(m & ((1ull << lane) - 1ull))] = c;
                        tail += __popcll(m);
                    }
                }
                __threadfence_block();
            }
            const int rows = tail;
            if (stamp && lane == 0) { stamp[2] = wall_clock64(); stamp[5] = rows; }
            __threadfence_block();
            __syncthreads();                                                  // vfi/vfo complete (single wave: ordering only)
            double w = 1e-4, lo_in = TBIG, lo_out = TBIG, lo_mix = TBIG;
            if (mode == LA_MODE_INPUT) {
                w = 0.0;
                int cnt = 0;
                for (int k = lane; k < rows; k += 64) cnt += vfi[k] > 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
                lo_in = cnt > max_size ? select_desc(vfi, rows, a.min_in <= 0 ? rows - 1 : min(a.min_in - 1, rows - 1), lane, hist, sel_vals) : 0.0;
            } else if (mode == LA_MODE_OUTPUT) {
                w = 1.0;
                int cnt = 0;
                for (int k = lane; k < rows; k += 64) cnt += vfo[k] > 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
                lo_out = cnt > max_size ? select_desc(vfo, rows, a.min_out <= 0 ? rows - 1 : min(a.min_out - 1, rows - 1), lane, hist, sel_vals) : 0.0;
            } else if (rows > max_size) {
                // rows carry None as their index (:152): the mix cut-off loop never fires, lo_mix stays 1e9
                if (a.min_in > 0) lo_in = select_desc(vfi, rows, min(a.min_in - 1, rows - 1), lane, hist, sel_vals);
                if (a.min_out > 0) lo_out = select_desc(vfo, rows, min(a.min_out - 1, rows - 1), lane, hist, sel_vals);
            } else {
                lo_mix = 0.0;
            }
            const double w1 = 1.0 - w;
            if (stamp && lane == 0) stamp[3] = wall_clock64();
            // ---- _ravel (lookahead_cache.py:248-293).  Round 3: a node's children are ORDERED ONCE, when the DFS first enters the
            // node — eligible children (the cut-off rule of :262-273) ranked by (fm desc, insertion position asc) into an LDS frame
            // of at most 64 entries (no node can contribute more rows than the budget left) — and then consumed from the frame.
            // Round 2 re-scanned ALL children of the node (two dependent global loads each) for every row it emitted or skipped:
            // 306 us median for ~35 rows below nodes with hundreds of children (profiles/r03_trie_device_profile.txt).
            const int last_tok = nrest > 0 ? q[nq - 1] : 0;
            if (lane == 0) { oid[0] = (nrest > 0 && last_tok != 0) ? last_tok : t.tok[root]; orm[0] = 1ull; }   // :129
            int n = 1, sp = 0;
            auto better = [](double afm, int ak, double bfm, int bk) { return afm > bfm || (afm == bfm && ak < bk); };
            // frame of node u at stack level lv: -> number of entries
            auto build_frame = [&](int lv, int u) -> int {
                const int cs = t.cstart[u], cc = t.ccount[u];
                double Lfm = -1.0; int Lk = 0x7fffffff, Lfl = 0; bool Lv = false;      // lane r holds the rank-r child found so far
                int nL = 0;
                for (int base = 0; base < cc; base += 64) {
                    const int k = base + lane;
                    const bool in = k < cc;
                    double cfi = 0.0, cfo = 0.0;
                    if (in) { cfi = t.fi[cs + k]; cfo = t.fo[cs + k]; }
                    const double fm = __dadd_rn(__dmul_rn(w1, cfi), __dmul_rn(w, cfo));      // :254, no FMA
                    bool skip;
                    if (mode == LA_MODE_MIX) skip = cfi < lo_in && cfo < lo_out && fm < lo_mix;            // :265
                    else if (mode == LA_MODE_INPUT) skip = cfi < lo_in;
                    else skip = cfo < lo_out;
                    bool Nv = in && !skip;
                    if (nL == 64) {                      // a full list: only children better than its last entry can enter
                        const double wfm = __shfl(Lfm, 63, 64); const int wk = __shfl(Lk, 63, 64);
                        Nv = Nv && better(fm, k, wfm, wk);
                    }
                    unsigned long long mN = __ballot(Nv);
                    if (mN == 0ull) continue;
                    const unsigned long long mL = __ballot(Lv);
                    int rL = lane, rN = 0;               // ranks in the union of the kept list and the new candidates
                    const int fl = (cfi > 0.0 ? 1 : 0) | (cfo > 0.0 ? 2 : 0);
                    for (unsigned long long m2 = mN; m2; m2 &= m2 - 1ull) {
                        const int j = __ffsll((long long)m2) - 1;
                        const double bfm = __shfl(fm, j, 64); const int bk = __shfl(k, j, 64);
                        if (Lv && better(bfm, bk, Lfm, Lk)) ++rL;
                        if (Nv && j != lane && better(bfm, bk, fm, k)) ++rN;
                    }
                    for (unsigned long long m2 = mL; m2; m2 &= m2 - 1ull) {
                        const int j = __ffsll((long long)m2) - 1;
                        const double bfm = __shfl(Lfm, j, 64); const int bk = __shfl(Lk, j, 64);
                        if (Nv && better(bfm, bk, fm, k)) ++rN;
                    }
                    if (Lv && rL < 64) { tmp_fm[rL] = Lfm; tmp_k[rL] = Lk; tmp_fl[rL] = Lfl; }
                    if (Nv && rN < 64) { tmp_fm[rN] = fm; tmp_k[rN] = k; tmp_fl[rN] = fl; }
                    nL = min(64, __popcll(mL) + __popcll(mN));
                    __syncthreads();
                    Lv = lane < nL;
                    if (Lv) { Lfm = tmp_fm[lane]; Lk = tmp_k[lane]; Lfl = tmp_fl[lane]; }
                    __syncthreads();
                }
                if (Lv) { fr_child[lv][lane] = cs + Lk; fr_flag[lv][lane] = (unsigned char)Lfl; }
                return nL;
            };
            {
                const int cnt = build_frame(0, cur);
                if (lane == 0) { fr_cnt[0] = cnt; fr_next[0] = 0; st_pid[0] = -1; st_depth[0] = max_length; }
            }
            __syncthreads();
            sp = 1;
            while (sp > 0 && n < max_size) {
                const int lv = sp - 1;
                const int nx = fr_next[lv];
                if (nx >= fr_cnt[lv]) { --sp; continue; }                     // children exhausted
                const int c = fr_child[lv][nx], fl = fr_flag[lv][nx], pid = st_pid[lv], depth = st_depth[lv];
                __syncthreads();                                              // everybody has read the cursor before lane 0 moves it
                if (lane == 0) fr_next[lv] = nx + 1;
                if (fl & 1) ++sz0;
                if (fl & 2) ++sz1;
                const int rid = n++;
                if (lane == 0) {
                    oid[rid] = t.tok[c];
                    orm[rid] = (pid > -1 ? orm[pid] : 1ull) | (1ull << rid);
                }
                if (t.ccount[c] > 0 && depth - 1 > 0 && n < max_size && sp < 72) {
                    const int cnt = build_frame(sp, c);
                    if (lane == 0) { fr_cnt[sp] = cnt; fr_next[sp] = 0; st_pid[sp] = rid; st_depth[sp] = depth - 1; }
                    ++sp;
                }
                __syncthreads();
            }
            n_out = n;
            if (stamp && lane == 0) { stamp[4] = wall_clock64(); stamp[6] = n; }
        }
        if (n_out >= branch_length) break;                                    // :433-434 (else a later suffix overwrites)
    }
    if (!have) {                                                              // :436-437
        if (nq > 0 && lane == 0) { oid[0] = q[nq - 1]; orm[0] = 1ull; }
        finish(nq > 0 ? 1 : 0, 0, 0, 2);
        return;
    }
    finish(n_out, sz0, sz1, 2);
}

int lk_trie_hier_get(hipStream_t st, const int* tok, const double* fo, const double* fi, const int* cstart, const int* ccount,
                     int n_nodes, const int* queries, const int* nq, int B, int decoding_length, int branch_length,
                     int min_in, int min_out, int mode, const int* stop, int n_stop, int* scratch_q, double* scratch_v,
                     int* out_ids, uint64_t* out_rowmask, int* out_n, int* out_sizes, int* out_nsizes) {
    TrieQueryArgs a{};
    a.plane = nullptr; a.fi_stride = 0; a.bl = nullptr;
    a.t = TrieDev{tok, fo, fi, cstart, ccount, n_nodes};
    a.queries = queries; a.nq = nq; a.decoding_length = decoding_length; a.branch_length = branch_length;
    a.min_in = min_in; a.min_out = min_out; a.mode = mode; a.stop = stop; a.n_stop = n_stop;
    a.scratch_q = scratch_q; a.scratch_v = scratch_v; a.out_ids = out_ids; a.out_rowmask = (unsigned long long*)out_rowmask;
    a.out_n = out_n; a.out_sizes = out_sizes; a.out_nsizes = out_nsizes; a.dbg = g_la_dbg_times;
    k_trie_hier_get<<<B, 64, 0, st>>>(a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}


// ---- one_get on the device (LookaheadCache.one_get, lookahead_cache.py:490-517, Tree.get_one_branch :171-222): one wavefront per query;
// prefix match as in hier_get, then per level the child of highest frequency (wave max + lowest lane = the reference's strict
// "freq > max_freq" scan in insertion order), a single chain of at most branch_length tokens with lower-triangular row masks.
__global__ __launch_bounds__(64) void k_trie_one_get(TrieQueryArgs a) {
    const int b = blockIdx.x, lane = threadIdx.x;
    TrieDev t = a.t;
    if (a.plane) t.fi += (size_t)a.plane[b] * (size_t)a.fi_stride;
    const int branch_length = a.bl ? a.bl[b] : a.branch_length;
    const int* q = a.queries + b * 8;
    const int nq = a.nq[b];
    int* oid = a.out_ids + b * 64;
    unsigned long long* orm = a.out_rowmask + b * 64;
    const int mode = a.mode;
    auto finish = [&](int n, int s0, int s1, int nsizes) {
        if (lane < n) orm[lane] = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
        if (lane == 0) { a.out_n[b] = n; a.out_sizes[b * 2] = s0; a.out_sizes[b * 2 + 1] = s1; a.out_nsizes[b] = nsizes; }
    };
    if (a.decoding_length <= 1 || branch_length == 0) {                       // :491-492
        if (nq > 0 && lane == 0) oid[0] = q[nq - 1];
        finish(nq > 0 ? 1 : 0, 0, 0, 0);
        return;
    }
    bool have = false;
    int n_out = 0, nsz = 2, len_out = 0;
    for (int i = 0; i < nq; ++i) {
        const int root = find_child(t, 0, q[i], lane);
        if (root < 0) continue;
        const int nrest = nq - (i + 1);
        bool is_stop = false;
        for (int k = 0; k < a.n_stop; ++k) is_stop |= (a.stop[k] == q[i]);
        if (is_stop && nrest == 0) continue;                                  // :500-501
        have = true;
        int cur = root;                                                       // Tree._match (:224-246)
        for (int k = 0; k < nrest && cur >= 0; ++k) {
            const int ch = find_child(t, cur, q[i + 1 + k], lane);
            if (ch < 0) { cur = -1; break; }
            const double cfi = t.fi[ch], cfo = t.fo[ch];
            const bool live = mode == LA_MODE_INPUT ? cfi > 0 : mode == LA_MODE_OUTPUT ? cfo > 0 : (cfi > 0 || cfo > 0);
            cur = live ? ch : -1;
        }
        if (cur < 0 || t.ccount[cur] == 0) {                                  // :175-177
            if (lane == 0) oid[0] = nrest > 0 ? q[nq - 1] : t.tok[root];
            n_out = 1; nsz = 2; len_out = 0;
        } else {
            const int last_tok = nrest > 0 ? q[nq - 1] : 0;
            if (lane == 0) oid[0] = (nrest > 0 && last_tok != 0) ? last_tok : t.tok[root];
            n_out = 1;
            int length = 0;
            while (length < branch_length) {
                const int cs = t.cstart[cur], cc = t.ccount[cur];
                if (cc == 0) break;
                double best = 0.0;                                            // max_freq = 0.0: a child needs freq > 0 (:186-201)
                int best_rec = -1;
                for (int base = 0; base < cc; base += 64) {
                    const int k = base + lane;
                    double v = -1.0;
                    if (k < cc) {
                        const double cfi = t.fi[cs + k], cfo = t.fo[cs + k];
                        if (mode == LA_MODE_MIX) {                            // :190-193 (names swapped there): freq = 10000 * freqs[-1] + freqs[idx]
                            if (cfi > 0 || cfo > 0) v = __dadd_rn(__dmul_rn(10000.0, cfo), cfi);
                        } else if (mode == LA_MODE_INPUT) { if (cfi > 0) v = cfi; }
                        else { if (cfo > 0) v = cfo; }
                    }
                    double wm = v;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) wm = fmax(wm, __shfl_xor(wm, o, 64));
                    if (wm > best) {                                          // strict: an equal later child never replaces an earlier one
                        const unsigned long long m = __ballot(v == wm);
                        best = wm; best_rec = cs + base + wave_first(m);
                    }
                }
                if (best_rec < 0) break;
                if (lane == 0) oid[n_out] = t.tok[best_rec];
                ++n_out; cur = best_rec; ++length;
            }
            nsz = 1; len_out = length;
        }
        if (n_out >= branch_length / 2) break;                                // :512
    }
    if (!have) {
        if (nq > 0 && lane == 0) oid[0] = q[nq - 1];
        finish(nq > 0 ? 1 : 0, 0, 0, 2);
        return;
    }
    finish(n_out, len_out, 0, nsz);
}

int lk_trie_one_get2(hipStream_t st, const int* tok, const double* fo, const double* fi, long fi_stride, const int* cstart,
                     const int* ccount, int n_nodes, const int* queries, const int* nq, const int* plane, const int* bl, int B,
                     int decoding_length, int branch_length, int mode, const int* stop, int n_stop, int* out_ids,
                     uint64_t* out_rowmask, int* out_n, int* out_sizes, int* out_nsizes) {
    TrieQueryArgs a{};
    a.t = TrieDev{tok, fo, fi, cstart, ccount, n_nodes};
    a.plane = plane; a.fi_stride = fi_stride; a.bl = bl;
    a.queries = queries; a.nq = nq; a.decoding_length = decoding_length; a.branch_length = branch_length;
    a.mode = mode; a.stop = stop; a.n_stop = n_stop;
    a.out_ids = out_ids; a.out_rowmask = (unsigned long long*)out_rowmask;
    a.out_n = out_n; a.out_sizes = out_sizes; a.out_nsizes = out_nsizes;
    k_trie_one_get<<<B, 64, 0, st>>>(a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// ---- incremental mirror: apply a patch (la_cache_mirror_patch) to the device image ---------------------------------------
__global__ void k_trie_patch(int* __restrict__ tok, double* __restrict__ fo, double* __restrict__ fi, long fi_stride,
                             int* __restrict__ cstart, int* __restrict__ ccount, int* __restrict__ ccap, const int* __restrict__ ipatch,
                             int n_i, const int* __restrict__ dkey, const double* __restrict__ dval, int n_d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_i) {
        const int arr = ipatch[3 * i], rec = ipatch[3 * i + 1], val = ipatch[3 * i + 2];
        if (arr == 3) { if (ccap) ccap[rec] = val; }              // block capacities: only the device-side update reads them
        else (arr == 0 ? tok : arr == 1 ? cstart : ccount)[rec] = val;
    } else if (i < n_i + n_d) {
        const int k = i - n_i;
        const int plane = dkey[2 * k], rec = dkey[2 * k + 1];
        if (plane == 0) fo[rec] = dval[k]; else fi[(size_t)(plane - 1) * (size_t)fi_stride + rec] = dval[k];
    }
}

int lk_trie_patch(hipStream_t st, int* tok, double* fo, double* fi, long fi_stride, int* cstart, int* ccount, int* ccap,
                  const int* ipatch, int n_i, const int* dkey, const double* dval, int n_d) {
    const int n = n_i + n_d;
    if (n <= 0) return 0;
    k_trie_patch<<<(n + 255) / 256, 256, 0, st>>>(tok, fo, fi, fi_stride, cstart, ccount, ccap, ipatch, n_i, dkey, dval, n_d);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int lk_trie_hier_get2(hipStream_t st, const int* tok, const double* fo, const double* fi, long fi_stride, const int* cstart,
                      const int* ccount, int n_nodes, const int* queries, const int* nq, const int* plane, const int* bl, int B,
                      int decoding_length, int branch_length, int min_in, int min_out, int mode, const int* stop, int n_stop,
                      int* scratch_q, double* scratch_v, int* out_ids, uint64_t* out_rowmask, int* out_n, int* out_sizes,
                      int* out_nsizes) {
    TrieQueryArgs a{};
    a.t = TrieDev{tok, fo, fi, cstart, ccount, n_nodes};
    a.plane = plane; a.fi_stride = fi_stride; a.bl = bl;
    a.queries = queries; a.nq = nq; a.decoding_length = decoding_length; a.branch_length = branch_length;
    a.min_in = min_in; a.min_out = min_out; a.mode = mode; a.stop = stop; a.n_stop = n_stop;
    a.scratch_q = scratch_q; a.scratch_v = scratch_v; a.out_ids = out_ids; a.out_rowmask = (unsigned long long*)out_rowmask;
    a.out_n = out_n; a.out_sizes = out_sizes; a.out_nsizes = out_nsizes; a.dbg = g_la_dbg_times;
    k_trie_hier_get<<<B, 64, 0, st>>>(a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}


// ---- device-side stream_put (LookaheadCache.stream_put, lookahead_cache.py:369-406, with Tree.put/_put/_pack :33-63) ------------
// The accepted tokens of a verify step never leave HBM on their way into the trie: the step's output block (or any int32 rows) is
// appended to the per-sequence hold-back buffers (_output_ids[idx]), every start offset that now has a full branch behind it is
// inserted — tree root by token, then branch_length nodes, output frequency + 1 each — and the buffers keep their last
// branch_length tokens.  The image grows by the SAME rule as the host mirror (Mirror::add_child in la_trie.cpp: a full child block
// moves to the arena end with twice the room, first block = 4 records), and the inserts run in the same order (puts in call order,
// start offsets ascending), so after the host replays the same puts on its trie the two images agree on every record id, block
// and capacity and on the frequencies of every live record (tests/test_gpu_trie.py; a dead copy left behind by a moved block may
// carry a count the parallel walk added just before the move — nothing reads it again) and no patch has to cross PCIe.
//   k_trie_put_walk : one wavefront per (put, start offset): walks the part of the branch that already exists (ballot child search,
//                     root through a token-indexed table) and adds its frequencies (fp64 atomics: integers, order-free);
//                     leaves {record where the path ended, levels done} per item.
//   k_trie_put_link : ONE wavefront, items in order: continues each unfinished branch from where its walk ended — a child created
//                     by an earlier item of the same call is found and incremented, anything else is appended (chain of fresh
//                     records: no search below a fresh node) — then rolls the hold-back buffers.  A block that moves re-bases the
//                     pending items' records (and the root table when it is the forest's root block).
// Capacity: an insert that would pass `cap` records sets meta[1] (sticky) and stops ALL further device inserts; the host (whose
// replay is the record of truth) sees its own record count pass the capacity and uploads a larger image.
// hold-back buffer of put k with the new tokens appended (-1 entries dropped, cut at the first eos: :350-352) -> LDS buf; returns ts
__device__ int put_window(const TriePutArgs& a, int k, int lane, int* buf, int* ol_out) {
    const int idx = a.put_idx[k];
    const int ol = min(max(a.olen[idx], 0), LA_TRIE_OBUF - LA_TRIE_ITEMS - 8);
    const int n_raw = min(max(a.src_cnt[k], 0), LA_TRIE_ITEMS);
    int t = -1;
    if (lane < n_raw) t = a.src_tok[(size_t)k * a.src_stride + lane];
    bool valid = lane < n_raw && t != -1;
    bool is_eos = false;
    for (int e = 0; e < a.n_eos; ++e) is_eos |= (valid && t == a.eos[e]);
    const unsigned long long em = __ballot(is_eos);
    if (em) valid = valid && lane < (__ffsll((long long)em) - 1);
    const unsigned long long vm = __ballot(valid);
    for (int i = lane; i < ol; i += 64) buf[i] = a.obuf[(size_t)idx * LA_TRIE_OBUF + i];
    if (valid) buf[ol + __popcll(vm & ((1ull << lane) - 1ull))] = t;
    __syncthreads();
    *ol_out = ol;
    return ol + __popcll(vm);
}

__device__ __forceinline__ int find_child_rw(const int* tok, const int* cstart, const int* ccount, int u, int token, int lane) {
    const int cs = cstart[u], cc = ccount[u];
    for (int base = 0; base < cc; base += 64) {
        const int i = base + lane;
        const bool hit = i < cc && tok[cs + i] == token;
        const unsigned long long m = __ballot(hit);
        if (m) return cs + base + wave_first(m);
    }
    return -1;
}

__device__ __forceinline__ int find_root(const TriePutArgs& a, int token, int lane) {
    if (token >= 0 && token < a.n_root_of) return a.root_of[token];
    return find_child_rw(a.tok, a.cstart, a.ccount, 0, token, lane);
}

__global__ __launch_bounds__(64) void k_trie_put_walk(TriePutArgs a) {
    __shared__ int buf[LA_TRIE_OBUF + 64];
    const int k = blockIdx.x, i = blockIdx.y, lane = threadIdx.x;
    int ol;
    const int ts = put_window(a, k, lane, buf, &ol);
    const int bl = a.branch_length;
    int* item = a.items + ((size_t)k * LA_TRIE_ITEMS + i) * 2;
    if (i >= ts - bl) { if (lane == 0) { item[0] = -1; item[1] = 0; } return; }
    const int root_tok = buf[i];
    bool is_stop = false;
    for (int s = 0; s < a.n_stop; ++s) is_stop |= (a.stop[s] == root_tok);                // :386-387
    if (is_stop || a.meta[1] != 0) { if (lane == 0) { item[0] = -1; item[1] = 0; } return; }
    int cur = find_root(a, root_tok, lane), lv = 0;
    if (cur >= 0) {
        lv = 1;
        for (; lv <= bl; ++lv) {
            const int ch = find_child_rw(a.tok, a.cstart, a.ccount, cur, buf[i + lv], lane);
            if (ch < 0) break;
            if (lane == 0) atomicAdd(&a.fo[ch], 1.0);                                     // :53 freqs[-1] += 1 (output mode)
            cur = ch;
        }
    } else {
        cur = 0;
    }
    if (lane == 0) { item[0] = lv > bl ? -1 : cur; item[1] = lv; }
}

__global__ __launch_bounds__(64) void k_trie_put_link(TriePutArgs a) {
    // First version: every level of every unfinished branch was a dependent global round trip (search, append, fence): 890 us for
    // the 8 x 7 branches of a Mistral bs=8 step (profiles/r03_trie_device_update.txt).  Now: (1) the walk left the node where each
    // branch ends and nothing but an EARLIER item of this call that appended under the same node can have changed that node, so
    // an item whose end node no earlier item touched appends without searching, from fields prefetched for all items at once;
    // (2) everything below the first appended node is a chain of fresh records whose ids follow from the record counter — the
    // whole chain is written in one wave-parallel step (lane j = chain node j, final field values, no read-modify-write).
    __shared__ int buf[LA_TRIE_OBUF + 64];
    __shared__ int it_u[LA_TRIE_PUTS * LA_TRIE_ITEMS], it_lv[LA_TRIE_PUTS * LA_TRIE_ITEMS];
    __shared__ int it_cs[LA_TRIE_PUTS * LA_TRIE_ITEMS], it_cc[LA_TRIE_PUTS * LA_TRIE_ITEMS], it_cp[LA_TRIE_PUTS * LA_TRIE_ITEMS];
    __shared__ unsigned char it_dirty[LA_TRIE_PUTS * LA_TRIE_ITEMS];
    const int lane = threadIdx.x;
    const int bl = a.branch_length, n_items = a.n_put * LA_TRIE_ITEMS;
    for (int q = lane; q < n_items; q += 64) {
        const int u = a.items[2 * q];
        it_u[q] = u; it_lv[q] = a.items[2 * q + 1]; it_dirty[q] = 0;
        if (u >= 0) { it_cs[q] = a.cstart[u]; it_cc[q] = a.ccount[u]; it_cp[q] = a.ccap[u]; }
    }
    __syncthreads();
    int n_rec = a.meta[0], ovf = a.meta[1], n_branch = 0, n_new = 0;
    for (int k = 0; k < a.n_put; ++k) {
        __syncthreads();
        int ol;
        const int ts = put_window(a, k, lane, buf, &ol);
        const int idx = a.put_idx[k];
        const int nit = min(max(ts - bl, 0), LA_TRIE_ITEMS);
        for (int i = 0; i < nit && !ovf; ++i) {
            const int q = k * LA_TRIE_ITEMS + i;
            int cur = it_u[q];
            if (cur < 0) continue;                                    // a stop-word root, or the whole branch existed already
            ++n_branch;
            int lv = it_lv[q];
            int cs = it_cs[q], cc = it_cc[q], cp = it_cp[q];
            if (it_dirty[q]) {
                // an earlier item of this call appended under this node: search again from here (its child may be ours)
                __threadfence_block();
                for (; lv <= bl; ++lv) {
                    const int t = buf[i + lv];
                    const int ch = lv == 0 ? find_root(a, t, lane) : find_child_rw(a.tok, a.cstart, a.ccount, cur, t, lane);
                    if (ch < 0) break;
                    if (lv >= 1 && lane == 0) atomicAdd(&a.fo[ch], 1.0);
                    cur = ch;
                }
                if (lv > bl) continue;
                cs = a.cstart[cur]; cc = a.ccount[cur]; cp = a.ccap[cur];
            }
            // ---- append: the first new node under `cur` (Mirror::add_child), then the chain below it
            const int m_chain = bl - lv;                              // fresh nodes below the first one
            int moved_from = -1, moved_cnt = 0, moved_to = 0;
            if (cc == cp) {                                           // block full: move it to the arena end with twice the room
                const int ncap = cp < 2 ? 4 : 2 * cp;
                if (n_rec + ncap + 4 * m_chain > a.cap) { ovf = 1; break; }
                __threadfence_block();                                // records of the old block written earlier in this launch
                const int nstart = n_rec, ostart = cs;
                for (int r = lane; r < ncap; r += 64) {
                    const int n = nstart + r, o = ostart + r;
                    const bool cpy = r < cc;
                    a.tok[n] = cpy ? a.tok[o] : -1; a.cstart[n] = cpy ? a.cstart[o] : 0; a.ccount[n] = cpy ? a.ccount[o] : 0;
                    a.ccap[n] = cpy ? a.ccap[o] : 0; a.fo[n] = cpy ? a.fo[o] : 0.0;
                    for (int p = 0; p < a.n_planes; ++p) a.fi[(size_t)p * a.fi_stride + n] = cpy ? a.fi[(size_t)p * a.fi_stride + o] : 0.0;
                    if (cpy && cur == 0) { const int t = a.tok[o]; if (t >= 0 && t < a.n_root_of) a.root_of[t] = n; }
                }
                moved_from = ostart; moved_cnt = cc; moved_to = nstart;
                n_rec += ncap;
                cs = nstart; cp = ncap;
                if (lane == 0) { a.cstart[cur] = nstart; a.ccap[cur] = ncap; }
            } else if (n_rec + 4 * m_chain > a.cap) { ovf = 1; break; }
            const int first = cs + cc;
            const int base = n_rec;                                   // chain node j = record base + 4 j (first slot of its own block)
            if (lane == 0) {
                const int t = buf[i + lv];
                a.ccount[cur] = cc + 1;
                a.tok[first] = t;
                a.cstart[first] = m_chain > 0 ? base : 0; a.ccount[first] = m_chain > 0 ? 1 : 0; a.ccap[first] = m_chain > 0 ? 4 : 0;
                a.fo[first] = lv >= 1 ? 1.0 : 0.0;                    // the tree root itself carries no frequency (:365-367)
                for (int p = 0; p < a.n_planes; ++p) a.fi[(size_t)p * a.fi_stride + first] = 0.0;
                if (cur == 0 && t >= 0 && t < a.n_root_of) a.root_of[t] = first;
            }
            for (int e = lane; e < 4 * m_chain; e += 64) {            // 4 records per chain block: slot 0 = the node, 1..3 spare
                const int j = e >> 2, slot = e & 3, rec = base + e;
                const bool node = slot == 0, last = j == m_chain - 1;
                a.tok[rec] = node ? buf[i + lv + 1 + j] : -1;
                a.cstart[rec] = node && !last ? base + 4 * (j + 1) : 0;
                a.ccount[rec] = node && !last ? 1 : 0;
                a.ccap[rec] = node && !last ? 4 : 0;
                a.fo[rec] = node ? 1.0 : 0.0;
                for (int p = 0; p < a.n_planes; ++p) a.fi[(size_t)p * a.fi_stride + rec] = 0.0;
            }
            n_rec += 4 * m_chain;
            n_new += 1 + m_chain;
            // pending items: records that sat in the moved block follow it; items that end at `cur` must search again
            for (int q2 = q + 1 + lane; q2 < n_items; q2 += 64) {
                int u2 = it_u[q2];
                if (u2 > 0 && u2 >= moved_from && u2 < moved_from + moved_cnt) { u2 = u2 - moved_from + moved_to; it_u[q2] = u2; }
                if (u2 == cur) it_dirty[q2] = 1;
            }
            __syncthreads();
        }
        // roll the hold-back buffer (:399-400)
        __syncthreads();
        if (ts > bl) {
            for (int j = lane; j < bl; j += 64) a.obuf[(size_t)idx * LA_TRIE_OBUF + j] = buf[ts - bl + j];
            if (lane == 0) a.olen[idx] = bl;
        } else {
            for (int j = ol + lane; j < ts; j += 64) a.obuf[(size_t)idx * LA_TRIE_OBUF + j] = buf[j];
            if (lane == 0) a.olen[idx] = ts;
        }
    }
    if (lane == 0) { a.meta[0] = n_rec; a.meta[1] = ovf; a.meta[2] += n_branch; a.meta[3] += n_new; }
}

__global__ void k_trie_root_index(const int* __restrict__ tok, const int* __restrict__ cstart, const int* __restrict__ ccount,
                                  int* __restrict__ root_of, int n_root_of) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int cs = cstart[0], cc = ccount[0];
    if (i < cc) { const int t = tok[cs + i]; if (t >= 0 && t < n_root_of) root_of[t] = cs + i; }
}

int lk_trie_root_index(hipStream_t st, const int* tok, const int* cstart, const int* ccount, int* root_of, int n_root_of, int n_roots_max) {
    hipError_t e = hipMemsetAsync(root_of, 0xff, (size_t)n_root_of * 4, st);
    if (e != hipSuccess) return (int)e;
    if (n_roots_max > 0) k_trie_root_index<<<(n_roots_max + 255) / 256, 256, 0, st>>>(tok, cstart, ccount, root_of, n_root_of);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int lk_trie_stream_put(hipStream_t st, const TriePutArgs& a) {
    k_trie_put_walk<<<dim3(a.n_put, LA_TRIE_ITEMS), 64, 0, st>>>(a);
    k_trie_put_link<<<1, 64, 0, st>>>(a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
