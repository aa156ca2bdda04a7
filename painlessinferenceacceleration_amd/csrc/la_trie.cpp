// la_trie.cpp — host trie cache behind the la_cache_* C ABI.
//
// Semantics follow lookahead/lookahead/common/lookahead_cache.py of the reference, restated on an
// index arena (no Python objects): every quirk that changes a returned draft is reproduced and
// tagged with the reference line it comes from.  Children keep dict insertion order through a
// sibling list; lookups go through one hash index keyed by (parent node, token).
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "../../include/lookahead_hip.h"

extern void la_set_error(const std::string& s);

namespace {

struct Node {
    int32_t token;
    int32_t parent;        // node index, -1 for a tree root
    int32_t first_child, last_child, next_sib, prev_sib;
    double fo;             // freqs[-1]
    std::vector<std::pair<int32_t, double>> fi;   // freqs[idx>=0], tiny
};

struct Tree {
    int32_t token;
    int32_t root;          // arena node whose children are Tree.nodes
    int64_t max_node, max_output_node;   // captured at creation (lookahead_cache.py:365)
    int64_t n_node, n_output_node;
    int64_t uid;
};

static inline uint64_t key_of(int32_t parent, int32_t token) {
    return ((uint64_t)(uint32_t)parent << 32) | (uint32_t)token;
}

// (parent node, token) -> child node: open addressing with linear probing over two flat arrays (values >= 0; -1 = empty slot),
// backward-shift deletion.  Round 3: std::unordered_map allocated one heap node per trie node — the largest single cost of an
// n-gram insert (scripts/host_trie_put_bench.py: 8 stream_puts of ~7 new tokens 380 -> 250 us on the build container).
struct FlatIndex {
    std::vector<uint64_t> keys;
    std::vector<int32_t> vals;
    size_t count = 0;
    int bits = 0;
    static inline size_t slot_of(uint64_t k, int b) { return (size_t)((k * 0x9E3779B97F4A7C15ull) >> (64 - b)); }
    void rehash(int nb) {
        std::vector<uint64_t> ok; std::vector<int32_t> ov;
        ok.swap(keys); ov.swap(vals);
        bits = nb;
        keys.assign((size_t)1 << bits, 0); vals.assign((size_t)1 << bits, -1);
        const size_t m = ((size_t)1 << bits) - 1;
        for (size_t i = 0; i < ov.size(); ++i)
            if (ov[i] >= 0) {
                size_t j = slot_of(ok[i], bits);
                while (vals[j] >= 0) j = (j + 1) & m;
                keys[j] = ok[i]; vals[j] = ov[i];
            }
    }
    int32_t find(uint64_t k) const {
        if (!bits) return -1;
        const size_t m = ((size_t)1 << bits) - 1;
        for (size_t i = slot_of(k, bits); vals[i] >= 0; i = (i + 1) & m)
            if (keys[i] == k) return vals[i];
        return -1;
    }
    void set(uint64_t k, int32_t v) {
        if (!bits) rehash(10);
        else if ((count + 1) * 4 > ((size_t)3 << bits)) rehash(bits + 1);       // load factor <= 0.75
        const size_t m = ((size_t)1 << bits) - 1;
        size_t i = slot_of(k, bits);
        for (; vals[i] >= 0; i = (i + 1) & m)
            if (keys[i] == k) { vals[i] = v; return; }
        keys[i] = k; vals[i] = v; ++count;
    }
    void erase(uint64_t k) {
        if (!bits) return;
        const size_t m = ((size_t)1 << bits) - 1;
        size_t i = slot_of(k, bits);
        for (; vals[i] >= 0; i = (i + 1) & m)
            if (keys[i] == k) break;
        if (vals[i] < 0) return;
        for (size_t j = i;;) {                              // close the gap: move back every entry the hole would cut off from its home slot
            j = (j + 1) & m;
            if (vals[j] < 0) break;
            const size_t home = slot_of(keys[j], bits);
            const bool between = i <= j ? (i < home && home <= j) : (i < home || home <= j);
            if (between) continue;
            keys[i] = keys[j]; vals[i] = vals[j];
            i = j;
        }
        vals[i] = -1;
        --count;
    }
    void clear() { keys.clear(); vals.clear(); count = 0; bits = 0; }
    void swap(FlatIndex& o) { keys.swap(o.keys); vals.swap(o.vals); std::swap(count, o.count); std::swap(bits, o.bits); }
};

}  // namespace

// ---- incremental device mirror (consumed by la_trie_hier_get_dev, csrc/la_trie_dev.hip) --------------------------------
// The host trie stays the single owner of all updates; the mirror is the SAME forest in the device layout — record 0 is a
// super-root whose children are the tree roots, the children of a record are consecutive records in insertion order — kept as
// host arrays plus a log of the words that changed since the last sync.  A node's child block grows like a vector (capacity
// doubling at the arena end; the old block becomes garbage), so an n-gram insert (<= 13 nodes) costs a handful of record
// writes, and a verify step's trie update reaches the device as a patch of a few hundred bytes instead of a re-export
// (212 ms for 157 k nodes, profiles/r01_trie_host_vs_device.txt).  Deletions (squeeze, fresh, load) mark the mirror stale:
// the next sync rebuilds and re-uploads it (they happen at request boundaries, lookahead_cache.py:572-576).
struct Mirror {
    std::vector<int32_t> plane_idx;                   // input-frequency slots mirrored as fi planes (e.g. 0..B-1)
    std::vector<int32_t> tok, cstart, ccount, ccap, host_of;
    std::vector<double> fo;
    std::vector<std::vector<double>> fi;              // [plane][record]
    std::vector<int32_t> dev_of;                      // host node id -> record (-1 = none)
    bool stale = true, full = true;
    std::unordered_map<uint64_t, size_t> ipos, dpos;  // (array, record) -> position in the log: last write wins
    std::vector<int32_t> ilog;                        // triples {array: 0 tok, 1 cstart, 2 ccount, 3 ccap; record; value}
    std::vector<int32_t> dkey;                        // pairs {plane: 0 fo, 1 + k fi plane k; record}
    std::vector<double> dval;

    int plane_of(int32_t idx) const {
        for (size_t k = 0; k < plane_idx.size(); ++k) if (plane_idx[k] == idx) return (int)k;
        return -1;
    }
    void clear_log() { ipos.clear(); dpos.clear(); ilog.clear(); dkey.clear(); dval.clear(); }
    void log_i(int arr, int32_t rec, int32_t val) {
        const uint64_t k = ((uint64_t)arr << 32) | (uint32_t)rec;
        auto it = ipos.find(k);
        if (it != ipos.end()) { ilog[it->second * 3 + 2] = val; return; }
        ipos[k] = ilog.size() / 3;
        ilog.push_back(arr); ilog.push_back(rec); ilog.push_back(val);
    }
    void log_d(int plane, int32_t rec, double val) {
        const uint64_t k = ((uint64_t)plane << 32) | (uint32_t)rec;
        auto it = dpos.find(k);
        if (it != dpos.end()) { dval[it->second] = val; return; }
        dpos[k] = dval.size();
        dkey.push_back(plane); dkey.push_back(rec); dval.push_back(val);
    }
    int32_t grow(int32_t n) {                          // n fresh records at the arena end
        const int32_t start = (int32_t)tok.size();
        tok.resize(start + n, -1); cstart.resize(start + n, 0); ccount.resize(start + n, 0); ccap.resize(start + n, 0);
        host_of.resize(start + n, -1); fo.resize(start + n, 0.0);
        for (auto& pl : fi) pl.resize(start + n, 0.0);
        return start;
    }
    void write_record(int32_t rec) {                   // log every word of a (new or moved) record
        log_i(0, rec, tok[rec]); log_i(1, rec, cstart[rec]); log_i(2, rec, ccount[rec]); log_i(3, rec, ccap[rec]);
        log_d(0, rec, fo[rec]);
        for (size_t k = 0; k < fi.size(); ++k) log_d(1 + (int)k, rec, fi[k][rec]);
    }
    // append a child record under `prec`; returns its record id
    int32_t add_child(int32_t prec, int32_t token, int32_t host_node) {
        if (ccount[prec] == ccap[prec]) {              // block full: move it to the arena end with twice the room
            const int32_t ncap = ccap[prec] < 2 ? 4 : 2 * ccap[prec];
            const int32_t nstart = grow(ncap), ostart = cstart[prec], cnt = ccount[prec];
            for (int32_t k = 0; k < cnt; ++k) {
                const int32_t o = ostart + k, n = nstart + k;
                tok[n] = tok[o]; cstart[n] = cstart[o]; ccount[n] = ccount[o]; ccap[n] = ccap[o]; host_of[n] = host_of[o];
                fo[n] = fo[o];
                for (auto& pl : fi) pl[n] = pl[o];
                if (host_of[n] >= 0) dev_of[host_of[n]] = n;
                host_of[o] = -1;
                write_record(n);
            }
            cstart[prec] = nstart; ccap[prec] = ncap;
            log_i(1, prec, nstart); log_i(3, prec, ncap);
        }
        const int32_t rec = cstart[prec] + ccount[prec];
        ccount[prec] += 1;
        log_i(2, prec, ccount[prec]);
        tok[rec] = token; cstart[rec] = 0; ccount[rec] = 0; ccap[rec] = 0; host_of[rec] = host_node; fo[rec] = 0.0;
        for (auto& pl : fi) pl[rec] = 0.0;
        if (host_node >= 0) {
            if ((size_t)host_node >= dev_of.size()) dev_of.resize((size_t)host_node + 1, -1);
            dev_of[host_node] = rec;
        }
        write_record(rec);
        return rec;
    }
};

struct la_cache {
    int64_t max_node, max_output_node;
    std::vector<int32_t> eos;            // empty <=> [None]
    std::unordered_set<int32_t> stop_words;
    std::vector<Node> nodes;
    std::vector<int32_t> free_nodes;
    FlatIndex child_index;
    std::unordered_map<int32_t, int32_t> mem;          // token -> tree slot
    std::vector<Tree> trees;
    std::vector<int32_t> free_trees;
    std::unordered_map<int64_t, int32_t> live_by_uid;  // uid -> tree slot
    std::unordered_set<int64_t> update_trees, update_input_trees;   // sets of Tree objects (by uid)
    std::unordered_map<int32_t, std::vector<int32_t>> output_ids;   // _output_ids[idx]
    int64_t next_uid = 1;
    int64_t live_nodes = 0;
    Mirror* mir = nullptr;                                        // optional device mirror (la_cache_mirror_*)
    ~la_cache() { delete mir; }
    bool mir_live() const { return mir && !mir->stale; }

    // ---- arena ----
    int32_t new_node(int32_t token, int32_t parent) {
        int32_t id;
        if (!free_nodes.empty()) { id = free_nodes.back(); free_nodes.pop_back(); }
        else { id = (int32_t)nodes.size(); nodes.emplace_back(); }
        Node& n = nodes[id];
        n.token = token; n.parent = parent;
        n.first_child = n.last_child = n.next_sib = n.prev_sib = -1;
        n.fo = 0.0; n.fi.clear();
        return id;
    }
    void link_child(int32_t parent, int32_t child) {
        Node& p = nodes[parent];
        Node& c = nodes[child];
        c.prev_sib = p.last_child; c.next_sib = -1;
        if (p.last_child >= 0) nodes[p.last_child].next_sib = child; else p.first_child = child;
        p.last_child = child;
        child_index.set(key_of(parent, c.token), child);
        ++live_nodes;
        if (mir_live()) {
            const int32_t prec = (size_t)parent < mir->dev_of.size() ? mir->dev_of[parent] : -1;
            if (prec < 0) mir->stale = true; else mir->add_child(prec, c.token, child);
        }
    }
    int32_t find_child(int32_t parent, int32_t token) const {
        return child_index.find(key_of(parent, token));
    }
    // delete `child` and its whole subtree (dict.pop of a Node drops everything below it)
    void drop_subtree(int32_t child) {
        if (mir) mir->stale = true;
        Node& c = nodes[child];
        Node& p = nodes[c.parent];
        if (c.prev_sib >= 0) nodes[c.prev_sib].next_sib = c.next_sib; else p.first_child = c.next_sib;
        if (c.next_sib >= 0) nodes[c.next_sib].prev_sib = c.prev_sib; else p.last_child = c.prev_sib;
        child_index.erase(key_of(c.parent, c.token));
        std::vector<int32_t> stack{child};
        bool top = true;
        while (!stack.empty()) {
            int32_t n = stack.back(); stack.pop_back();
            for (int32_t ch = nodes[n].first_child; ch >= 0; ch = nodes[ch].next_sib) stack.push_back(ch);
            if (!top) child_index.erase(key_of(nodes[n].parent, nodes[n].token));
            top = false;
            nodes[n].fi.clear(); nodes[n].fi.shrink_to_fit();
            nodes[n].first_child = nodes[n].last_child = -1;
            free_nodes.push_back(n);
            --live_nodes;
        }
    }
    static double get_fi(const Node& n, int32_t idx) {
        if (idx == -1) return n.fo;               // freqs is one dict: idx -1 aliases the output slot
        for (auto& p : n.fi) if (p.first == idx) return p.second;
        return 0.0;
    }
    static void add_freq(Node& n, int32_t idx, double f) {
        if (idx == -1) { n.fo += f; return; }
        for (auto& p : n.fi) if (p.first == idx) { p.second += f; return; }
        n.fi.emplace_back(idx, f);
    }
    static void set_fi(Node& n, int32_t idx, double f) {
        if (idx == -1) { n.fo = f; return; }
        for (auto& p : n.fi) if (p.first == idx) { p.second = f; return; }
        n.fi.emplace_back(idx, f);
    }

    // mirror: the freq slot `idx` of `node` changed on the host
    void mir_freq(int32_t node, int32_t idx) {
        if (!mir_live()) return;
        const int32_t rec = (size_t)node < mir->dev_of.size() ? mir->dev_of[node] : -1;
        if (rec < 0) { mir->stale = true; return; }
        if (idx == -1) { mir->fo[rec] = nodes[node].fo; mir->log_d(0, rec, nodes[node].fo); return; }
        const int pl = mir->plane_of(idx);
        if (pl < 0) return;                                          // this input slot is not mirrored
        const double v = get_fi(nodes[node], idx);
        mir->fi[pl][rec] = v; mir->log_d(1 + pl, rec, v);
    }

    // ---- Tree ----
    int32_t tree_get_or_create(int32_t token, bool* created) {
        auto it = mem.find(token);
        if (it != mem.end()) { *created = false; return it->second; }
        int32_t slot;
        if (!free_trees.empty()) { slot = free_trees.back(); free_trees.pop_back(); }
        else { slot = (int32_t)trees.size(); trees.emplace_back(); }
        Tree& t = trees[slot];
        t.token = token; t.root = new_node(token, -1);
        t.max_node = max_node; t.max_output_node = max_output_node;
        t.n_node = 0; t.n_output_node = 0; t.uid = next_uid++;
        mem[token] = slot;
        live_by_uid[t.uid] = slot;
        if (mir_live()) mir->add_child(0, token, t.root);            // tree roots are the super-root's children
        *created = true;
        return slot;
    }
    // Tree.put/_put/_pack (lookahead_cache.py:33-63)
    void tree_put(Tree& t, const int32_t* toks, int n, bool output_mode, int32_t idx) {
        if (output_mode) idx = -1;                                  // :35-36
        int32_t cur = t.root;
        for (int i = 0; i < n; ++i) {
            int32_t ch = find_child(cur, toks[i]);
            if (ch < 0) {                                           // :45-51 pack the rest as a chain
                for (int j = i; j < n; ++j) {
                    int32_t nn = new_node(toks[j], cur);
                    link_child(cur, nn);
                    add_freq(nodes[nn], idx, 1.0);
                    mir_freq(nn, idx);
                    cur = nn;
                }
                t.n_node += n - i;
                if (output_mode) t.n_output_node += n - i;
                return;
            }
            add_freq(nodes[ch], idx, 1.0);                          // :53
            mir_freq(ch, idx);
            cur = ch;
        }
    }

    struct Thresholds { double min_in, min_out, min_mix, output_weight; };

    // _dfs_get_freqs (:146-154): rows of live nodes reachable through live nodes
    void dfs_freqs(int32_t parent, int32_t idx, std::vector<double>& fis, std::vector<double>& fos) const {
        std::vector<int32_t> stack;
        // iterative pre-order; row order does not matter for the thresholds (only k-th largest values do)
        for (int32_t ch = nodes[parent].first_child; ch >= 0; ch = nodes[ch].next_sib) stack.push_back(ch);
        while (!stack.empty()) {
            int32_t n = stack.back(); stack.pop_back();
            const Node& nd = nodes[n];
            double fo = nd.fo, fi = get_fi(nd, idx);
            if (fo > 0 || fi > 0) {
                fis.push_back(fi); fos.push_back(fo);
                for (int32_t ch = nd.first_child; ch >= 0; ch = nodes[ch].next_sib) stack.push_back(ch);
            }
        }
    }
    // value at position (k-1) of the descending sort; Python's negative index for k==0 -> the minimum
    static double kth_desc(std::vector<double> v, int k) {
        size_t pos = (k <= 0) ? v.size() - 1 : (size_t)(k - 1);
        if (pos >= v.size()) pos = v.size() - 1;   // the reference would raise IndexError here
        std::nth_element(v.begin(), v.begin() + pos, v.end(), std::greater<double>());
        return v[pos];
    }

    // rows: ancestor masks, row i at rows + i * words (one flat scratch buffer per thread: a query allocates nothing)
    struct Out {
        int cap; int32_t* ids; int32_t* parent; uint64_t* rows; int words; int n;
        int32_t sizes[2];
    };
    struct Ent { int32_t node; double fm; };

    // _ravel (:248-293)
    void ravel(int32_t parent_node, int pid, int max_size, int max_length, const Thresholds& th, int mode,
               int32_t idx, Out& o) const {
        if (o.n >= max_size || max_length <= 0) return;
        // the children of this level live on a per-thread stack shared by the whole recursion: [base, end) is this call's range
        // (entries are read by value: deeper levels may grow the stack and move it)
        static thread_local std::vector<Ent> stk;
        const size_t base = stk.size();
        for (int32_t ch = nodes[parent_node].first_child; ch >= 0; ch = nodes[ch].next_sib) {
            const Node& nd = nodes[ch];
            double fm = (1.0 - th.output_weight) * get_fi(nd, idx) + th.output_weight * nd.fo;   // :254
            stk.push_back({ch, fm});
        }
        const size_t end = stk.size();
        std::stable_sort(stk.begin() + base, stk.begin() + end, [](const Ent& a, const Ent& b) { return a.fm > b.fm; });
        struct Pop { std::vector<Ent>& v; size_t n; ~Pop() { v.resize(n); } } pop{stk, base};
        for (size_t si = base; si < end; ++si) {
            const Ent e = stk[si];
            if (o.n >= max_size) return;                                                         // :260
            const Node& nd = nodes[e.node];
            double fi = get_fi(nd, idx), fo = nd.fo;
            if (mode == LA_MODE_MIX) {
                if (fi < th.min_in && fo < th.min_out && e.fm < th.min_mix) continue;            // :265
            } else if (mode == LA_MODE_INPUT) {
                if (fi < th.min_in) continue;
            } else {
                if (fo < th.min_out) continue;
            }
            if (fi > 0.0) o.sizes[0]++;
            if (fo > 0.0) o.sizes[1]++;
            int rid = o.n++;
            o.ids[rid] = nd.token;
            o.parent[rid] = pid > -1 ? pid : 0;
            uint64_t* row = o.rows + (size_t)rid * o.words;
            if (pid > -1) std::copy(o.rows + (size_t)pid * o.words, o.rows + (size_t)(pid + 1) * o.words, row);
            else { std::fill(row, row + o.words, 0); row[0] = 1; }
            row[rid >> 6] |= 1ull << (rid & 63);
            if (nd.first_child >= 0)
                ravel(e.node, rid, max_size, max_length - 1, th, mode, idx, o);
        }
    }

    // Tree._match (:224-246): returns the node whose children are `nodes`, or -1 for an empty dict
    int32_t match(const Tree& t, const int32_t* q, int nq, int mode, int32_t idx, bool* have_tok, int32_t* tok) const {
        *have_tok = false;
        if (nq == 0) return t.root;
        int32_t cur = t.root;
        for (int i = 0; i < nq; ++i) {
            *have_tok = true; *tok = q[i];
            if (cur < 0) return -1;                      // nodes = {} -> next get() is None -> break
            int32_t ch = find_child(cur, q[i]);
            if (ch < 0) return -1;
            const Node& nd = nodes[ch];
            bool live;
            if (mode == LA_MODE_INPUT) live = get_fi(nd, idx) > 0;
            else if (mode == LA_MODE_OUTPUT) live = nd.fo > 0;
            else live = get_fi(nd, idx) > 0 || nd.fo > 0;
            cur = live ? ch : -1;
        }
        return cur;
    }

    // Tree.get (:65-144).  Returns number of ids written.
    int tree_get(const Tree& t, const int32_t* q, int nq, int max_size, int max_length, int min_input_size,
                 int min_output_size, int mode, int32_t idx, Out& o) const {
        bool have_tok; int32_t tok = 0;
        int32_t at = match(t, q, nq, mode, idx, &have_tok, &tok);
        o.n = 0; o.sizes[0] = o.sizes[1] = 0;
        auto single = [&](int32_t token) {
            o.ids[0] = token; o.parent[0] = -1;
            std::fill(o.rows, o.rows + o.words, 0); o.rows[0] = 1;
            o.n = 1;
            return 1;
        };
        if (at < 0 || nodes[at].first_child < 0)                   // len(nodes) == 0  (:70-72)
            return single(nq > 0 ? q[nq - 1] : t.token);

        std::vector<double> fis, fos;
        dfs_freqs(at, idx, fis, fos);
        Thresholds th{1e9, 1e9, 1e9, 1e-4};
        const size_t rows = fis.size();
        if (mode == LA_MODE_INPUT) {
            th.output_weight = 0.0;
            size_t size = 0; for (double f : fis) if (f > 0) ++size;
            if ((int64_t)size > max_size) th.min_in = kth_desc(fis, min_input_size); else th.min_in = 0.0;
        } else if (mode == LA_MODE_OUTPUT) {
            th.output_weight = 1.0;
            size_t size = 0; for (double f : fos) if (f > 0) ++size;
            if ((int64_t)size > max_size) th.min_out = kth_desc(fos, min_output_size); else th.min_out = 0.0;
        } else {
            if ((int64_t)rows > max_size) {
                // every row's index slot is None (:152), so `indices` collapses to {None} and the mix loop
                // (:111-123) never assigns min_mix_freq: it stays 1e9.
                if (min_input_size > 0) th.min_in = kth_desc(fis, min_input_size);
                if (min_output_size > 0) th.min_out = kth_desc(fos, min_output_size);
            } else {
                th.min_mix = 0.0;                                                               // :125
            }
        }
        // ids[0] = match_token_id or self.token_id (:129): token 0 is falsy
        int32_t root_tok = (have_tok && tok != 0) ? tok : t.token;
        if (max_size <= 0) { o.n = 0; return 0; }
        single(root_tok);
        o.parent[0] = -1;
        ravel(at, -1, max_size, max_length, th, mode, idx, o);
        return o.n;
    }

    // squeeze (:295-318)
    void squeeze_rec(int32_t parent) {
        int32_t ch = nodes[parent].first_child;
        while (ch >= 0) {
            int32_t next = nodes[ch].next_sib;
            if (nodes[ch].fo > 1.0) {
                nodes[ch].fo *= 0.5;
                if (mir) mir->stale = true;
                if (nodes[ch].first_child >= 0) squeeze_rec(ch);
            } else {
                drop_subtree(ch);
            }
            ch = next;
        }
    }
    int64_t count_nodes(int32_t parent) const {
        int64_t n = 0;
        std::vector<int32_t> stack{parent};
        while (!stack.empty()) {
            int32_t p = stack.back(); stack.pop_back();
            for (int32_t ch = nodes[p].first_child; ch >= 0; ch = nodes[ch].next_sib) { ++n; stack.push_back(ch); }
        }
        return n;
    }
    void tree_squeeze(Tree& t) {
        if (t.n_node > t.max_node || t.n_output_node > t.max_output_node) {
            squeeze_rec(t.root);
            int64_t c = count_nodes(t.root);
            t.n_node = c; t.n_output_node = c;                     // :300-301
        }
    }
    // reset_input_freq (:320-333): stops descending where the slot is already 0
    void reset_rec(int32_t parent, int32_t idx) {
        std::vector<int32_t> stack{parent};
        while (!stack.empty()) {
            int32_t p = stack.back(); stack.pop_back();
            for (int32_t ch = nodes[p].first_child; ch >= 0; ch = nodes[ch].next_sib) {
                double f = get_fi(nodes[ch], idx);
                if (f == 0.0) continue;
                set_fi(nodes[ch], idx, 0.0);
                mir_freq(ch, idx);
                if (nodes[ch].first_child >= 0) stack.push_back(ch);
            }
        }
    }

    void reset_input_freqs(int32_t idx) {                          // :566-570
        for (int64_t uid : update_input_trees) {
            auto it = live_by_uid.find(uid);
            if (it != live_by_uid.end()) reset_rec(trees[it->second].root, idx);
        }
        update_input_trees.clear();
    }
    void squeeze_branch_counts() {                                 // :572-576
        if (update_trees.size() >= 1024) {
            for (int64_t uid : update_trees) {
                auto it = live_by_uid.find(uid);
                if (it != live_by_uid.end()) tree_squeeze(trees[it->second]);
            }
            update_trees.clear();
        }
    }
    void truncate_eos(const int32_t* toks, int& n) const {          // :350-352
        for (int32_t e : eos)
            for (int i = 0; i < n; ++i) if (toks[i] == e) { n = i; break; }
    }
};

extern "C" {

la_cache* la_cache_create(int max_node, int max_output_node) {
    la_cache* c = new la_cache();
    c->max_node = max_node; c->max_output_node = max_output_node;
    c->eos.push_back(2);
    return c;
}
void la_cache_destroy(la_cache* c) { delete c; }

int la_cache_set_limits(la_cache* c, int max_node, int max_output_node) {
    if (!c) return LA_E_ARG;
    c->max_node = max_node; c->max_output_node = max_output_node;
    return LA_OK;
}
int la_cache_set_eos(la_cache* c, const int32_t* eos_ids, int n) {
    if (!c || n < 0 || (n > 0 && !eos_ids)) return LA_E_ARG;
    c->eos.assign(eos_ids, eos_ids + n);
    return LA_OK;
}
int la_cache_set_stop_words(la_cache* c, const int32_t* ids, int n) {
    if (!c || n < 0 || (n > 0 && !ids)) return LA_E_ARG;
    c->stop_words.clear();
    for (int i = 0; i < n; ++i) c->stop_words.insert(ids[i]);
    return LA_OK;
}
int la_cache_fresh(la_cache* c) {
    if (!c) return LA_E_ARG;
    // self.mem = {} : trees die, the dirty sets keep their (now dead) members, stream buffers stay
    c->mem.clear(); c->live_by_uid.clear();
    c->nodes.clear(); c->free_nodes.clear(); c->child_index.clear();
    c->trees.clear(); c->free_trees.clear();
    c->live_nodes = 0;
    if (c->mir) c->mir->stale = true;
    return LA_OK;
}

int la_cache_put(la_cache* c, const int32_t* toks, int n, int branch_length, int final_, int mode, int idx) {
    if (!c || n < 0 || (n > 0 && !toks)) return LA_E_ARG;
    if (mode != LA_MODE_INPUT && mode != LA_MODE_OUTPUT) { la_set_error("put: mode must be input|output"); return LA_E_ARG; }
    c->truncate_eos(toks, n);
    if (n >= 2) {
        for (int i = 0; i < n - 1; ++i) {
            int len = std::min(branch_length, n - (i + 1));
            if (len < 0) len = 0;
            bool created;
            int32_t slot = c->tree_get_or_create(toks[i], &created);
            Tree& t = c->trees[slot];
            c->tree_put(t, toks + i + 1, len, mode == LA_MODE_OUTPUT, idx);
            if (!created) c->update_trees.insert(t.uid);            // :361-367: new trees are not marked
            if (mode == LA_MODE_INPUT) c->update_input_trees.insert(t.uid);
        }
    }
    if (final_) { c->reset_input_freqs(idx); c->squeeze_branch_counts(); }
    return LA_OK;
}

int la_cache_stream_put(la_cache* c, const int32_t* toks, int n, int branch_length, int final_, int idx) {
    if (!c || n < 0 || (n > 0 && !toks)) return LA_E_ARG;
    if (idx < 0) { la_set_error("stream_put: idx must be >= 0"); return LA_E_ARG; }
    c->truncate_eos(toks, n);
    std::vector<int32_t>& buf = c->output_ids[idx];
    buf.insert(buf.end(), toks, toks + n);
    const int ts = (int)buf.size();
    const int min_bl = final_ ? 1 : branch_length;
    if (ts > min_bl) {
        for (int i = 0; i < ts - min_bl; ++i) {
            int32_t tok = buf[i];
            if (c->stop_words.count(tok)) continue;
            int len = std::min(branch_length, ts - (i + 1));
            if (len < 0) len = 0;
            bool created;
            int32_t slot = c->tree_get_or_create(tok, &created);
            Tree& t = c->trees[slot];
            c->tree_put(t, buf.data() + i + 1, len, true, idx);
            c->update_trees.insert(t.uid);
        }
        if (!final_) {
            // output_ids[ts - branch_length:]  (a negative start clamps to 0)
            int start = ts - branch_length; if (start < 0) start = 0;
            std::vector<int32_t> rest(buf.begin() + start, buf.end());
            buf.swap(rest);
        }
    }
    if (final_) {
        buf.clear();
        c->reset_input_freqs(idx);
        c->squeeze_branch_counts();
    }
    return LA_OK;
}

// stream_put for several sequences in ONE call (a batch step's accepted tokens, pretrained_model_batch.py:1254-1259): put k appends
// toks[offsets[k] .. offsets[k + 1]) to slot idxs[k], in order — exactly n la_cache_stream_put calls without n trips through the binding
int la_cache_stream_put_many(la_cache* c, const int32_t* toks, const int32_t* offsets, const int32_t* idxs, int n, int branch_length,
                             int final_) {
    if (!c || n < 0 || (n > 0 && (!offsets || !idxs))) return LA_E_ARG;
    for (int k = 0; k < n; ++k) {
        const int len = offsets[k + 1] - offsets[k];
        if (len < 0 || (len > 0 && !toks)) return LA_E_ARG;
        const int rc = la_cache_stream_put(c, toks ? toks + offsets[k] : nullptr, len, branch_length, final_, idxs[k]);
        if (rc != LA_OK) return rc;
    }
    return LA_OK;
}

static void emit(const la_cache::Out& o, int n, uint64_t* out_rowmask, int64_t* out_mask) {
    // packed rows: W = ceil(decoding_length / 64) words per row (W = 1 for the 64-row device block: out_rowmask[i] as before;
    // wide trees, decoding_length up to 256: row i = words [i * W, (i + 1) * W))
    if (out_rowmask)
        for (int i = 0; i < n; ++i)
            for (int w = 0; w < o.words; ++w) out_rowmask[(size_t)i * o.words + w] = o.rows[(size_t)i * o.words + w];
    if (out_mask)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) out_mask[(size_t)i * n + j] = (o.rows[(size_t)i * o.words + (j >> 6)] >> (j & 63)) & 1ull;
}

int la_cache_hier_get(la_cache* c, const int32_t* q, int nq, int decoding_length, int branch_length,
                      int min_input_size, int min_output_size, int mode, int idx, int cap, int32_t* out_ids,
                      int32_t* out_parent, uint64_t* out_rowmask, int64_t* out_mask, int32_t out_sizes[2],
                      int32_t* out_nsizes, int32_t* out_n) {
    if (!c || nq < 0 || (nq > 0 && !q) || !out_ids || !out_parent || !out_sizes || !out_nsizes || !out_n || cap < 1)
        return LA_E_ARG;
    if (mode < 0 || mode > 2) { la_set_error("hier_get: bad mode"); return LA_E_ARG; }
    static thread_local std::vector<uint64_t> rows;                 // [decoding_length][words], reused by every query of the thread
    la_cache::Out o{cap, out_ids, out_parent, nullptr, 0, 0, {0, 0}};
    auto fallback_last = [&](int nsizes) {                          // token_ids[-1:], default_mask
        *out_nsizes = nsizes; out_sizes[0] = out_sizes[1] = 0;
        if (nq == 0) { *out_n = 0; return LA_OK; }
        out_ids[0] = q[nq - 1]; out_parent[0] = -1; *out_n = 1;
        // the same packed layout as emit(): row 0 = W words, W = ceil(decoding_length / 64) (W = 1 when decoding_length <= 64)
        if (out_rowmask) {
            const int W = decoding_length > 64 && decoding_length <= cap ? (decoding_length + 63) / 64 : 1;
            out_rowmask[0] = 1;
            for (int w = 1; w < W; ++w) out_rowmask[w] = 0;
        }
        if (out_mask) out_mask[0] = 1;
        return LA_OK;
    };
    if (decoding_length <= 1 || branch_length == 0) return fallback_last(0);    // :413-414
    if (decoding_length > cap) { la_set_error("hier_get: decoding_length exceeds output capacity"); return LA_E_RANGE; }
    o.words = (decoding_length + 63) / 64;
    rows.assign((size_t)decoding_length * o.words, 0);
    o.rows = rows.data();
    bool have = false;
    int n_out = 0;
    for (int i = 0; i < nq; ++i) {
        auto it = c->mem.find(q[i]);
        if (it == c->mem.end()) continue;
        const int rest = nq - (i + 1);
        if (c->stop_words.count(q[i]) && rest == 0) continue;        // :422-423
        n_out = c->tree_get(c->trees[it->second], q + i + 1, rest, decoding_length, branch_length,
                            min_input_size, min_output_size, mode, idx, o);
        have = true;                                                // later lookups overwrite earlier ones
        if (n_out >= branch_length) break;                          // :433-434
    }
    if (!have) return fallback_last(2);                             // :436-437, sizes stays [0,0]
    *out_n = n_out; *out_nsizes = 2;
    out_sizes[0] = o.sizes[0]; out_sizes[1] = o.sizes[1];
    emit(o, n_out, out_rowmask, out_mask);
    return LA_OK;
}

// par_get (:441-488): the hierarchical draft re-laid as independent root-to-leaf chains under a block mask.
//   rows are walked from last to first; a row's ancestor set (columns 1.. of its mask row, :454-455) is kept unless an already kept
//   set covers it (:457-462: only maximal paths survive); the kept paths, in draft order (:464), are laid out one after another,
//   each truncated to what is left of the budget true_decoding_length = len(hier ids) - 1 (:466-477); mask = lower-triangular ones
//   with masks[start : start + len, 1 : start] = 0 per chain (:479-486): row r of a chain starting at `start` sees the root and
//   rows start..r.  sizes = [rows behind the root] (:488).
int la_cache_par_get(la_cache* c, const int32_t* q, int nq, int decoding_length, int branch_length, int min_input_size,
                     int min_output_size, int mode, int idx, int cap, int32_t* out_ids, uint64_t* out_rowmask, int64_t* out_mask,
                     int32_t out_sizes[2], int32_t* out_nsizes, int32_t* out_n) {
    if (!c || nq < 0 || (nq > 0 && !q) || !out_ids || !out_sizes || !out_nsizes || !out_n || cap < 1) return LA_E_ARG;
    static thread_local std::vector<int32_t> hids, hpar;
    static thread_local std::vector<uint64_t> hrows;
    const int W = decoding_length > 64 && decoding_length <= cap ? (decoding_length + 63) / 64 : 1;
    hids.assign((size_t)cap, 0); hpar.assign((size_t)cap, 0); hrows.assign((size_t)cap * W, 0);
    int32_t hs[2] = {0, 0}, hns = 0, T = 0;
    const int rc = la_cache_hier_get(c, q, nq, decoding_length, branch_length, min_input_size, min_output_size, mode, idx, cap,
                                     hids.data(), hpar.data(), hrows.data(), nullptr, hs, &hns, &T);
    if (rc != LA_OK) return rc;
    out_sizes[0] = out_sizes[1] = 0; *out_nsizes = 1;
    if (T == 0) { *out_n = 0; return LA_OK; }            // (the reference raises IndexError on an empty query: nothing to lay out)
    const int budget = T - 1;
    // member set of row i = its mask row without column 0, as a bitset over columns 1..T-1 (bit j-1 <=> column j)
    auto members = [&](int row, uint64_t* m) {
        for (int w = 0; w < W; ++w) {
            uint64_t lo = hrows[(size_t)row * W + w] >> 1;
            if (w + 1 < W) lo |= hrows[(size_t)row * W + w + 1] << 63;
            m[w] = lo;
        }
    };
    std::vector<uint64_t> kept;                            // kept sets, W words each, in discovery order (last row first)
    std::vector<uint64_t> m((size_t)W);
    for (int row = budget; row >= 1; --row) {
        members(row, m.data());
        bool covered = false;
        for (size_t k = 0; k < kept.size() / (size_t)W && !covered; ++k) {
            bool sub = true;
            for (int w = 0; w < W; ++w) sub = sub && (m[w] & ~kept[k * W + w]) == 0ull;
            covered = sub;
        }
        if (!covered) kept.insert(kept.end(), m.begin(), m.end());
    }
    const int nk = (int)(kept.size() / (size_t)W);
    int used = 0, n = 1;
    out_ids[0] = hids[0];
    if (out_rowmask) { out_rowmask[0] = 1ull; for (int w = 1; w < W; ++w) out_rowmask[w] = 0ull; }
    for (int k = nk - 1; k >= 0 && used < budget; --k) {  // reversed: draft order
        const int start = n;
        for (int col = 0; col < budget && used < budget; ++col) {
            if (!((kept[(size_t)k * W + (col >> 6)] >> (col & 63)) & 1ull)) continue;
            out_ids[n] = hids[col + 1];
            if (out_rowmask) {
                uint64_t* r = out_rowmask + (size_t)n * W;
                for (int w = 0; w < W; ++w) r[w] = 0ull;
                r[0] = 1ull;
                for (int j = start; j <= n; ++j) r[j >> 6] |= 1ull << (j & 63);
            }
            ++n; ++used;
        }
    }
    if (out_mask)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) out_mask[(size_t)i * n + j] = 0;
    if (out_mask) {
        int i = 0;
        out_mask[0] = 1;
        // rebuild from the chain structure: a row sees the root and the rows of its own chain up to itself
        int start = 1;
        for (int k = nk - 1, u = 0; k >= 0 && u < budget; --k) {
            int len = 0;
            for (int col = 0; col < budget && u + len < budget; ++col)
                if ((kept[(size_t)k * W + (col >> 6)] >> (col & 63)) & 1ull) ++len;
            for (i = start; i < start + len; ++i) {
                out_mask[(size_t)i * n] = 1;
                for (int j = start; j <= i; ++j) out_mask[(size_t)i * n + j] = 1;
            }
            start += len; u += len;
        }
    }
    out_sizes[0] = n - 1;
    *out_n = n;
    return LA_OK;
}

// Tree.get_one_branch (:171-222) + one_get (:490-517)
int la_cache_one_get(la_cache* c, const int32_t* q, int nq, int decoding_length, int branch_length, int mode,
                     int idx, int cap, int32_t* out_ids, int32_t out_sizes[2], int32_t* out_nsizes, int32_t* out_n) {
    if (!c || nq < 0 || (nq > 0 && !q) || !out_ids || !out_sizes || !out_nsizes || !out_n || cap < 1) return LA_E_ARG;
    if (mode < 0 || mode > 2) return LA_E_ARG;
    out_sizes[0] = out_sizes[1] = 0;
    if (decoding_length <= 1 || branch_length == 0) {
        *out_nsizes = 0;
        if (nq == 0) { *out_n = 0; return LA_OK; }
        out_ids[0] = q[nq - 1]; *out_n = 1; return LA_OK;
    }
    if (branch_length + 1 > cap) return LA_E_RANGE;
    bool have = false; int n_out = 0; int nsz = 2;
    for (int i = 0; i < nq; ++i) {
        auto it = c->mem.find(q[i]);
        if (it == c->mem.end()) continue;
        const int rest = nq - (i + 1);
        if (c->stop_words.count(q[i]) && rest == 0) continue;
        const Tree& t = c->trees[it->second];
        bool have_tok; int32_t tok = 0;
        int32_t at = c->match(t, q + i + 1, rest, mode, idx, &have_tok, &tok);
        have = true;
        if (at < 0 || c->nodes[at].first_child < 0) {
            out_ids[0] = rest > 0 ? q[nq - 1] : t.token; n_out = 1; nsz = 2; out_sizes[0] = out_sizes[1] = 0;
        } else {
            out_ids[0] = (have_tok && tok != 0) ? tok : t.token;
            n_out = 1;
            int32_t cur = at; int length = 0;
            while (c->nodes[cur].first_child >= 0 && length < branch_length) {
                double max_freq = 0.0; int32_t max_node = -1;
                for (int32_t ch = c->nodes[cur].first_child; ch >= 0; ch = c->nodes[ch].next_sib) {
                    const Node& nd = c->nodes[ch];
                    double freq; bool live;
                    if (mode == LA_MODE_MIX) {
                        // :190-193 names are swapped in the reference: fo := freqs[idx], fi := freqs[-1]
                        double a = la_cache::get_fi(nd, idx), b = nd.fo;
                        live = a > 0 || b > 0; freq = 10000 * b + a;
                    } else if (mode == LA_MODE_INPUT) { freq = la_cache::get_fi(nd, idx); live = freq > 0; }
                    else { freq = nd.fo; live = freq > 0; }
                    if (live && freq > max_freq) { max_freq = freq; max_node = ch; }
                }
                if (max_node < 0) break;
                out_ids[n_out++] = c->nodes[max_node].token;
                cur = max_node; ++length;
            }
            nsz = 1; out_sizes[0] = length;
        }
        if (n_out >= branch_length / 2) break;                      // :512
    }
    if (!have) {
        *out_nsizes = 2; out_sizes[0] = out_sizes[1] = 0;
        if (nq == 0) { *out_n = 0; return LA_OK; }
        out_ids[0] = q[nq - 1]; *out_n = 1; return LA_OK;
    }
    *out_n = n_out; *out_nsizes = nsz;
    return LA_OK;
}

int la_cache_reset_input_freqs(la_cache* c, int idx) { if (!c) return LA_E_ARG; c->reset_input_freqs(idx); return LA_OK; }
int la_cache_squeeze(la_cache* c) { if (!c) return LA_E_ARG; c->squeeze_branch_counts(); return LA_OK; }

int la_cache_stats(la_cache* c, int64_t* n_trees, int64_t* n_nodes_live, int64_t* n_dirty, int64_t* n_dirty_in) {
    if (!c) return LA_E_ARG;
    if (n_trees) *n_trees = (int64_t)c->mem.size();
    if (n_nodes_live) *n_nodes_live = c->live_nodes;
    if (n_dirty) *n_dirty = (int64_t)c->update_trees.size();
    if (n_dirty_in) *n_dirty_in = (int64_t)c->update_input_trees.size();
    return LA_OK;
}
int la_cache_tree_counters(la_cache* c, int32_t token, int64_t* n_node, int64_t* n_output_node) {
    if (!c) return LA_E_ARG;
    auto it = c->mem.find(token);
    if (it == c->mem.end()) return LA_E_RANGE;
    if (n_node) *n_node = c->trees[it->second].n_node;
    if (n_output_node) *n_output_node = c->trees[it->second].n_output_node;
    return LA_OK;
}

// ---- device mirror export (for la_trie_hier_get_dev): live nodes renumbered breadth-first so that the children of
//      a node are consecutive ids in insertion order; node 0 is a super-root whose children are the tree roots.
int la_cache_export(la_cache* c, int idx, int32_t cap, int32_t* tok, double* fo, double* fi, int32_t* cstart,
                    int32_t* ccount, int32_t* n_nodes) {
    if (!c || !n_nodes) return LA_E_ARG;
    // count: super root + one root per tree + live arena nodes below them
    int64_t total = 1 + (int64_t)c->mem.size() + c->live_nodes;
    *n_nodes = (int32_t)total;
    if (cap == 0) return LA_OK;
    if (cap < total || !tok || !fo || !fi || !cstart || !ccount) return LA_E_RANGE;
    std::vector<int32_t> order;           // arena node id per exported id (>= 1)
    order.reserve((size_t)total);
    order.push_back(-1);                  // super root
    // tree roots in a deterministic order (by token); lookups are by token, so the order is free
    std::vector<std::pair<int32_t, int32_t>> roots;
    for (auto& kv : c->mem) roots.push_back({kv.first, c->trees[kv.second].root});
    std::sort(roots.begin(), roots.end());
    tok[0] = -1; fo[0] = 0; fi[0] = 0; cstart[0] = 1; ccount[0] = (int32_t)roots.size();
    for (auto& r : roots) order.push_back(r.second);
    for (size_t i = 1; i < order.size(); ++i) {
        const Node& nd = c->nodes[order[i]];
        tok[i] = nd.token; fo[i] = nd.fo; fi[i] = la_cache::get_fi(nd, idx);
        cstart[i] = (int32_t)order.size();
        int32_t cnt = 0;
        for (int32_t ch = nd.first_child; ch >= 0; ch = c->nodes[ch].next_sib) { order.push_back(ch); ++cnt; }
        ccount[i] = cnt;
        if ((int64_t)order.size() > total) return LA_E_STATE;
    }
    *n_nodes = (int32_t)order.size();
    return LA_OK;
}

// ---- incremental mirror: enable / state / full image / patch ------------------------------------------------------------
static void mirror_rebuild(la_cache* c) {
    Mirror& m = *c->mir;
    const size_t planes = m.plane_idx.size();
    m.tok.clear(); m.cstart.clear(); m.ccount.clear(); m.ccap.clear(); m.host_of.clear(); m.fo.clear();
    m.fi.assign(planes, std::vector<double>());
    m.dev_of.assign(c->nodes.size(), -1);
    m.clear_log();
    m.grow(1);                                           // super root
    std::vector<int32_t> roots;
    for (auto& kv : c->mem) roots.push_back(c->trees[kv.second].root);
    std::sort(roots.begin(), roots.end());               // creation order of the arena = a deterministic order
    // A rebuilt block gets room to grow (count / 4, at least one record): with exactly-full blocks the FIRST child appended under any
    // node after a rebuild moved that node's whole block (for the forest's root block: thousands of records per new tree) — 5 000
    // patch words per verify step and most of the device-side link kernel's time (profiles/r03_trie_device_update.txt).
    auto roomy = [](int32_t cnt) { return cnt + std::max<int32_t>(1, cnt / 4); };
    const int32_t n_roots = (int32_t)roots.size();
    const int32_t r0 = m.grow(n_roots ? roomy(n_roots) : 0);
    m.cstart[0] = r0; m.ccount[0] = n_roots; m.ccap[0] = n_roots ? roomy(n_roots) : 0;
    std::vector<int32_t> order;                          // records still to expand (breadth-first)
    for (size_t i = 0; i < roots.size(); ++i) {
        const int32_t rec = r0 + (int32_t)i;
        m.host_of[rec] = roots[i]; m.dev_of[roots[i]] = rec;
        order.push_back(rec);
    }
    for (size_t qi = 0; qi < order.size(); ++qi) {
        const int32_t rec = order[qi], node = m.host_of[rec];
        const Node& nd = c->nodes[node];
        m.tok[rec] = nd.token; m.fo[rec] = nd.fo;
        for (size_t k = 0; k < planes; ++k) m.fi[k][rec] = la_cache::get_fi(nd, m.plane_idx[k]);
        int32_t cnt = 0;
        for (int32_t ch = nd.first_child; ch >= 0; ch = c->nodes[ch].next_sib) ++cnt;
        if (cnt == 0) continue;
        const int32_t cs = m.grow(roomy(cnt));
        m.cstart[rec] = cs; m.ccount[rec] = cnt; m.ccap[rec] = roomy(cnt);
        int32_t k = 0;
        for (int32_t ch = nd.first_child; ch >= 0; ch = c->nodes[ch].next_sib, ++k) {
            m.host_of[cs + k] = ch; m.dev_of[ch] = cs + k;
            order.push_back(cs + k);
        }
    }
    m.stale = false;
    m.full = true;
}

int la_cache_mirror_enable(la_cache* c, const int32_t* idx_planes, int n_planes) {
    if (!c || n_planes < 0 || n_planes > 64 || (n_planes > 0 && !idx_planes)) return LA_E_ARG;
    delete c->mir;
    c->mir = new Mirror();
    c->mir->plane_idx.assign(idx_planes, idx_planes + n_planes);
    c->mir->stale = true;
    return LA_OK;
}

int la_cache_mirror_state(la_cache* c, int32_t* n_records, int32_t* full, int32_t* n_ipatch, int32_t* n_dpatch) {
    if (!c || !c->mir || !n_records || !full || !n_ipatch || !n_dpatch) return LA_E_ARG;
    if (c->mir->stale) mirror_rebuild(c);
    const Mirror& m = *c->mir;
    *n_records = (int32_t)m.tok.size();
    *full = m.full ? 1 : 0;
    *n_ipatch = (int32_t)(m.ilog.size() / 3);
    *n_dpatch = (int32_t)m.dval.size();
    return LA_OK;
}

// whole image into caller buffers of `cap` records (fi: [planes][cap]); clears the patch log
int la_cache_mirror_image(la_cache* c, int32_t cap, int32_t* tok, double* fo, double* fi, int32_t* cstart, int32_t* ccount) {
    if (!c || !c->mir || !tok || !fo || !cstart || !ccount) return LA_E_ARG;
    if (c->mir->stale) mirror_rebuild(c);
    Mirror& m = *c->mir;
    const size_t n = m.tok.size();
    if ((size_t)cap < n || (!m.fi.empty() && !fi)) return LA_E_RANGE;
    memcpy(tok, m.tok.data(), n * 4); memcpy(cstart, m.cstart.data(), n * 4); memcpy(ccount, m.ccount.data(), n * 4);
    memcpy(fo, m.fo.data(), n * 8);
    for (size_t k = 0; k < m.fi.size(); ++k) memcpy(fi + k * (size_t)cap, m.fi[k].data(), n * 8);
    m.clear_log();
    m.full = false;
    return LA_OK;
}

// block capacities of the image la_cache_mirror_image just produced (the device-side update, la_trie_stream_put_dev, grows child
// blocks by the same rule as Mirror::add_child and needs them); call right after la_cache_mirror_image
int la_cache_mirror_ccap(la_cache* c, int32_t cap, int32_t* ccap) {
    if (!c || !c->mir || !ccap) return LA_E_ARG;
    Mirror& m = *c->mir;
    if (m.stale) { la_set_error("mirror_ccap: the mirror is stale (take la_cache_mirror_image first)"); return LA_E_STATE; }
    if ((size_t)cap < m.ccap.size()) return LA_E_RANGE;
    memcpy(ccap, m.ccap.data(), m.ccap.size() * 4);
    return LA_OK;
}

// The device applied the SAME updates itself (la_trie_stream_put_dev, then the host replayed them with la_cache_stream_put in the
// same order): the words the replay logged are already in the device image — drop them.  LA_E_STATE when the replay left the
// mirror stale or a full image is due (then the device image is NOT the host's and the caller must upload one).
int la_cache_mirror_discard(la_cache* c, int32_t* n_records) {
    if (!c || !c->mir) return LA_E_ARG;
    Mirror& m = *c->mir;
    if (m.stale || m.full) { la_set_error("mirror_discard: a full image is due"); return LA_E_STATE; }
    m.clear_log();
    if (n_records) *n_records = (int32_t)m.tok.size();
    return LA_OK;
}

// _output_ids[idx] (lookahead_cache.py:369-375): the tokens stream_put holds back until a full branch follows them
int la_cache_stream_buffer(la_cache* c, int idx, int32_t cap, int32_t* out, int32_t* n) {
    if (!c || !n || cap < 0 || (cap > 0 && !out)) return LA_E_ARG;
    auto it = c->output_ids.find(idx);
    const size_t have = it == c->output_ids.end() ? 0 : it->second.size();
    *n = (int32_t)have;
    if (have > (size_t)cap) return cap == 0 ? LA_OK : LA_E_RANGE;
    if (have) memcpy(out, it->second.data(), have * 4);
    return LA_OK;
}

// the words that changed since the last sync: ipatch int32[n_i][3] = {array (0 tok, 1 cstart, 2 ccount, 3 ccap), record, value},
// dkey int32[n_d][2] = {plane (0 fo, 1 + k fi plane k), record}, dval double[n_d]; every (array, record) appears once
int la_cache_mirror_patch(la_cache* c, int32_t* ipatch, int32_t* dkey, double* dval) {
    if (!c || !c->mir) return LA_E_ARG;
    Mirror& m = *c->mir;
    if (m.stale || m.full) { la_set_error("mirror_patch: a full image is due (see la_cache_mirror_state)"); return LA_E_STATE; }
    if (!m.ilog.empty()) { if (!ipatch) return LA_E_ARG; memcpy(ipatch, m.ilog.data(), m.ilog.size() * 4); }
    if (!m.dval.empty()) {
        if (!dkey || !dval) return LA_E_ARG;
        memcpy(dkey, m.dkey.data(), m.dkey.size() * 4); memcpy(dval, m.dval.data(), m.dval.size() * 8);
    }
    m.clear_log();
    return LA_OK;
}

// bat_get() without the padded canvas (lookahead_cache.py:519-561): the per-sample drafts of a batch in ONE call — same
// budget rule (decoding_length // bs per sample, min_output_size = max(per // 2, 1), idx = indices[b]); hier: ids + 64-bit
// row masks, one: ids + chain masks.  Rows b of the outputs hold out_n[b] <= cap entries.
int la_cache_bat_get_packed(la_cache* c, const int32_t* queries, const int32_t* nq, int q_stride, int bs, int decoding_length,
                            int branch_length, int mode, const int32_t* indices, int one_branch, int cap, int32_t* out_ids,
                            uint64_t* out_rowmask, int32_t* out_n, int32_t* out_sizes, int32_t* out_nsizes) {
    if (!c || !queries || !nq || !indices || !out_ids || !out_rowmask || !out_n || !out_sizes || !out_nsizes || bs < 1 ||
        cap < 1 || cap > 64 || q_stride < 1) return LA_E_ARG;
    const int per = decoding_length / bs;
    const int min_out = per / 2 > 1 ? per / 2 : 1;
    std::vector<int32_t> parent((size_t)cap);
    for (int b = 0; b < bs; ++b) {
        int32_t* ids = out_ids + (size_t)b * cap;
        uint64_t* rm = out_rowmask + (size_t)b * cap;
        int rc;
        if (!one_branch) {
            rc = la_cache_hier_get(c, queries + (size_t)b * q_stride, nq[b], per, branch_length, 0, min_out, mode, indices[b], cap,
                                   ids, parent.data(), rm, nullptr, out_sizes + 2 * b, out_nsizes + b, out_n + b);
            if (rc == LA_OK && out_n[b] == 1) rm[0] = 1ull;
        } else {
            rc = la_cache_one_get(c, queries + (size_t)b * q_stride, nq[b], per, branch_length, mode, indices[b], cap, ids,
                                  out_sizes + 2 * b, out_nsizes + b, out_n + b);
            for (int i = 0; rc == LA_OK && i < out_n[b]; ++i) rm[i] = i == 63 ? ~0ull : ((2ull << i) - 1ull);
        }
        if (rc != LA_OK) return rc;
    }
    return LA_OK;
}

// ---- persistence: "LATRIE01" | n_trees | per tree {token, max_node, max_output_node, n_node, n_output_node,
//      n_rec} | per record (pre-order, insertion order) {token, depth, fo, n_fi, (idx, f)*}
int la_cache_save(la_cache* c, const char* path) {
    if (!c || !path) return LA_E_ARG;
    // written next to the target and renamed over it only when complete: a crash or a full disk mid-save leaves the
    // previous snapshot intact
    const std::string tmp_path = std::string(path) + ".tmp";
    FILE* f = fopen(tmp_path.c_str(), "wb");
    if (!f) { la_set_error(std::string("save: cannot open ") + tmp_path); return LA_E_IO; }
    auto w = [&](const void* p, size_t n) { return fwrite(p, 1, n, f) == n; };
    bool ok = w("LATRIE01", 8);
    // keep dict order of mem irrelevant: trees are looked up by token only
    int64_t nt = (int64_t)c->mem.size();
    ok = ok && w(&nt, 8);
    for (auto& kv : c->mem) {
        const Tree& t = c->trees[kv.second];
        std::vector<std::pair<int32_t, int32_t>> order;   // (node, depth) pre-order
        std::vector<std::pair<int32_t, int32_t>> stack;
        std::vector<int32_t> kids;
        for (int32_t ch = c->nodes[t.root].first_child; ch >= 0; ch = c->nodes[ch].next_sib) kids.push_back(ch);
        for (auto it = kids.rbegin(); it != kids.rend(); ++it) stack.push_back({*it, 1});
        while (!stack.empty()) {
            auto [n, d] = stack.back(); stack.pop_back();
            order.push_back({n, d});
            kids.clear();
            for (int32_t ch = c->nodes[n].first_child; ch >= 0; ch = c->nodes[ch].next_sib) kids.push_back(ch);
            for (auto it = kids.rbegin(); it != kids.rend(); ++it) stack.push_back({*it, d + 1});
        }
        int64_t hdr[6] = {t.token, t.max_node, t.max_output_node, t.n_node, t.n_output_node, (int64_t)order.size()};
        ok = ok && w(hdr, sizeof(hdr));
        for (auto& [n, d] : order) {
            const Node& nd = c->nodes[n];
            int32_t rec[3] = {nd.token, d, (int32_t)nd.fi.size()};
            ok = ok && w(rec, sizeof(rec)) && w(&nd.fo, 8);
            for (auto& p : nd.fi) { ok = ok && w(&p.first, 4) && w(&p.second, 8); }
        }
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) { remove(tmp_path.c_str()); la_set_error("save: short write"); return LA_E_IO; }
    if (rename(tmp_path.c_str(), path) != 0) { remove(tmp_path.c_str()); la_set_error(std::string("save: cannot rename to ") + path); return LA_E_IO; }
    return LA_OK;
}

int la_cache_load(la_cache* live, const char* path) {
    if (!live || !path) return LA_E_ARG;
    FILE* f = fopen(path, "rb");
    if (!f) { la_set_error(std::string("load: cannot open ") + path); return LA_E_IO; }
    auto r = [&](void* p, size_t n) { return fread(p, 1, n, f) == n; };
    char magic[8];
    if (!r(magic, 8) || memcmp(magic, "LATRIE01", 8) != 0) { fclose(f); la_set_error("load: bad magic"); return LA_E_IO; }
    // The snapshot is parsed into a scratch cache and swapped in only when the whole file was valid: a truncated or corrupt
    // file leaves the live forest untouched (the reference's load_mem keeps self.mem when unpickling fails,
    // lookahead_cache.py:583-587).
    la_cache scratch;
    la_cache* c = &scratch;
    c->max_node = live->max_node; c->max_output_node = live->max_output_node; c->next_uid = live->next_uid;
    int64_t nt = 0;
    bool ok = r(&nt, 8);
    for (int64_t ti = 0; ok && ti < nt; ++ti) {
        int64_t hdr[6];
        ok = r(hdr, sizeof(hdr));
        if (!ok) break;
        bool created;
        int32_t slot = c->tree_get_or_create((int32_t)hdr[0], &created);
        Tree& t = c->trees[slot];
        t.max_node = hdr[1]; t.max_output_node = hdr[2]; t.n_node = hdr[3]; t.n_output_node = hdr[4];
        std::vector<int32_t> path_nodes{t.root};
        for (int64_t i = 0; ok && i < hdr[5]; ++i) {
            int32_t rec[3]; double fo;
            ok = r(rec, sizeof(rec)) && r(&fo, 8);
            if (!ok || rec[1] < 1 || rec[1] > (int32_t)path_nodes.size() || rec[2] < 0) { ok = false; break; }
            path_nodes.resize(rec[1]);
            int32_t nn = c->new_node(rec[0], path_nodes.back());
            c->link_child(path_nodes.back(), nn);
            c->nodes[nn].fo = fo;
            for (int k = 0; ok && k < rec[2]; ++k) {
                int32_t id; double v;
                ok = r(&id, 4) && r(&v, 8);
                if (ok) c->nodes[nn].fi.emplace_back(id, v);
            }
            path_nodes.push_back(nn);
        }
    }
    fclose(f);
    if (!ok) { la_set_error("load: truncated or corrupt snapshot"); return LA_E_IO; }
    // load_mem replaces self.mem only: dirty sets, stream buffers, eos / stop words of the live cache stay
    live->mem.swap(c->mem); live->live_by_uid.swap(c->live_by_uid);
    live->nodes.swap(c->nodes); live->free_nodes.swap(c->free_nodes); live->child_index.swap(c->child_index);
    live->trees.swap(c->trees); live->free_trees.swap(c->free_trees);
    live->live_nodes = c->live_nodes; live->next_uid = c->next_uid;
    if (live->mir) live->mir->stale = true;
    return LA_OK;
}

}  // extern "C"
