// la_abi.cpp — C-ABI glue: error string, argument validation, thin wrappers over the kernel
// launchers for the single-kernel entry points declared in include/lookahead_hip.h.
#include <hip/hip_runtime.h>
#include <string>
#include "la_common.h"
#include "la_kernels.h"
#include "la_mblock.h"
#include "la_trie_dev.h"

static thread_local std::string g_err;
void la_set_error(const std::string& s) { g_err = s; }

#define WRAP(call) do { int e_ = (call); if (e_ != 0) { \
    la_set_error(std::string(#call) + ": " + (e_ > 0 ? hipGetErrorString((hipError_t)e_) : "bad argument")); \
    return e_ > 0 ? LA_E_HIP : LA_E_ARG; } return LA_OK; } while (0)

extern "C" {

int la_abi_version(void) { return LA_ABI_VERSION; }
int la_abi_dtype(void) { return LA_DTYPE; }
const char* la_last_error(void) { return g_err.c_str(); }
extern int g_la_stop_layers, g_la_graph_epoch;
int la_mb_gemm(void* stream, int kind, const void* wp, const void* xp, int N, int K, int nblk, int n_wg, int ksplit,
               float* slabs, int slab_rows, void* act_xp, void* logits, float* cand_val, int32_t* cand_idx, const int32_t* pos,
               const void* rcos, const void* rsin, void* qf, void* kfresh, void* vfresh, int nh, int nkv) {
    if (!wp || !xp) return LA_E_ARG;
    MbGemm g{}; g.wp = wp; g.xp = xp; g.N = N; g.K = K; g.nblk = nblk; g.n_wg = n_wg; g.ksplit = ksplit;
    g.slabs = slabs; g.slab_rows = slab_rows; g.act_xp = act_xp; g.logits = logits; g.cand_val = cand_val; g.cand_idx = cand_idx;
    g.pos = pos; g.rcos = rcos; g.rsin = rsin; g.qf = qf; g.kfresh = kfresh; g.vfresh = vfresh; g.nh = nh; g.nkv = nkv;
    WRAP(lk_mb_gemm((hipStream_t)stream, kind, g));
}
// The product header keeps ONE debug key: 13 = depth probe of the parity tests (the step runs the first n layers, then the final
// norm + lm_head).  The measurement knobs / A/B switches are la_lab_* (la_lab.cpp, include/lookahead_hip_lab.h).
int la_debug_set(int key, int value) {
    if (key == 13 && value >= 0) { g_la_stop_layers = value; ++g_la_graph_epoch; return LA_OK; }
    return LA_E_ARG;
}
int la_debug_get(int key) { return key == 13 ? g_la_stop_layers : LA_E_ARG; }

int la_build_tree_inputs(void* stream, const int32_t* d_in, int32_t* d_state, int32_t* d_pos, uint64_t* d_rowmask,
                         int32_t* d_ids) {
    if (!d_in || !d_state || !d_pos || !d_rowmask || !d_ids) return LA_E_ARG;
    WRAP(lk_build_tree_inputs((hipStream_t)stream, d_in, d_state, d_pos, d_rowmask, d_ids));
}
int la_accept_scan(void* stream, const int32_t* d_ids, const uint64_t* d_rowmask, int32_t* d_state) {
    if (!d_ids || !d_rowmask || !d_state) return LA_E_ARG;
    WRAP(lk_accept_scan((hipStream_t)stream, d_ids, d_rowmask, d_state));
}
int la_kv_commit(void* stream, const void* kf, const void* vf, void* km, void* vm, const int32_t* d_state,
                 int n_layers, int n_kv_heads, int max_keys) {
    if (!kf || !vf || !km || !vm || !d_state || n_layers <= 0 || n_kv_heads <= 0 || max_keys % 32) return LA_E_ARG;
    WRAP(lk_kv_commit((hipStream_t)stream, kf, vf, km, vm, d_state, n_layers, n_kv_heads, max_keys));
}
int la_pack_weight(void* stream, const void* w, const void* w2, int N, int K, int interleave2, void* out) {
    if (!w || !out || N % 32 || K % 16 || (interleave2 && !w2)) return LA_E_ARG;
    WRAP(lk_pack_weight((hipStream_t)stream, w, w2, N, K, interleave2, out));
}
int la_pack_x(void* stream, const void* x, int K, void* out) {
    if (!x || !out || K % 16) return LA_E_ARG;
    WRAP(lk_pack_x((hipStream_t)stream, x, K, out));
}
int la_gemm64_slab(void* stream, const void* wp, const void* xp, int N, int K, int rb, int ksplit, float* slabs) {
    if (!wp || !xp || !slabs || N % 32 || K % 16 || ksplit < 1 || ksplit > 16 || ((rb & 0xff) != 1 && (rb & 0xff) != 2)) return LA_E_ARG;
    WRAP(lk_gemm64_slab((hipStream_t)stream, wp, xp, N, K, rb, ksplit, slabs));
}
int la_gemm64_swiglu(void* stream, const void* wp, const void* xp, int F, int K, void* act, int variant) {
    if (!wp || !xp || !act || F % 32 || K % 16) return LA_E_ARG;
    WRAP(lk_gemm64_swiglu((hipStream_t)stream, wp, xp, F, K, act, variant));
}
int la_gemm64_qkv(void* stream, const void* wp, const void* xp, int nh, int nkv, int K, const int32_t* pos, const void* rcos,
                  const void* rsin, void* qf, void* kf, void* vf, int variant) {
    if (!wp || !xp || !pos || !rcos || !rsin || !qf || !kf || !vf || nh <= 0 || nkv <= 0 || K % 16) return LA_E_ARG;
    WRAP(lk_gemm64_qkv((hipStream_t)stream, wp, xp, nh, nkv, K, pos, rcos, rsin, qf, kf, vf, variant));
}
int la_qkv_row_perm(int nh, int nkv, int32_t* perm) {
    if (!perm || nh <= 0 || nkv <= 0) return LA_E_ARG;
    lk_qkv_row_perm(nh, nkv, perm);
    return LA_OK;
}
int la_head_lane_map(int head_dim, int32_t* lane_src) {
    if (!lane_src || head_dim < 2 || head_dim > 128 || (head_dim & 1)) return LA_E_ARG;
    const int half = head_dim / 2;
    for (int j = 0; j < 128; ++j) {
        const int d = j & 63;
        lane_src[j] = d < half ? (j < 64 ? d : half + d) : -1;
    }
    return LA_OK;
}
int la_rowplan(int kind, int n_rows, int n_wg, int32_t* out) {
    if (kind < 0 || kind > 2 || n_rows <= 0 || n_wg <= 0) return LA_E_ARG;
    int n = lk_rowplan(kind, n_rows, n_wg, out);
    return n < 0 ? LA_E_RANGE : n;
}
int64_t la_planned_elems(int kind, int n_rows, int K, int n_wg) {
    if (kind < 0 || kind > 2 || K % 16 || n_wg <= 0 || lk_rowplan(kind, n_rows, n_wg, nullptr) < 0) return LA_E_RANGE;
    return (int64_t)lk_planned_elems(kind, n_rows, K, n_wg);
}
int la_pack_planned(void* stream, const void* w, const void* w2, const int32_t* d_plan, int kind, int n_rows, int K, int n_wg,
                    void* out) {
    if (!w || !d_plan || !out || kind < 0 || kind > 2 || (kind == 1 && !w2) || K % 16 || n_wg <= 0) return LA_E_ARG;
    if (lk_rowplan(kind, n_rows, n_wg, nullptr) < 0) return LA_E_RANGE;
    WRAP(lk_pack_planned((hipStream_t)stream, w, w2, d_plan, kind, n_rows, K, n_wg, out));
}
int la_gemm64r_swiglu(void* stream, const void* wp, const void* xp, int F, int K, int n_wg, void* act) {
    if (!wp || !xp || !act || K % 16 || n_wg <= 0) return LA_E_ARG;
    if (lk_gemm64r_init() != 0) return LA_E_HIP;
    WRAP(lk_gemm64r_swiglu((hipStream_t)stream, wp, xp, F, K, n_wg, act));
}
int la_gemm64r_logits(void* stream, const void* wp, const void* xp, int V, int K, int n_wg, void* logits, float* cv, int32_t* ci) {
    if (!wp || !xp || !cv || !ci || K % 16 || n_wg <= 0) return LA_E_ARG;
    if (lk_gemm64r_init() != 0) return LA_E_HIP;
    WRAP(lk_gemm64r_logits((hipStream_t)stream, wp, xp, V, K, n_wg, logits, cv, ci));
}
int la_gemm64r_qkv(void* stream, const void* wp, const void* xp, int nh, int nkv, int K, int n_wg, const int32_t* pos,
                   const void* rcos, const void* rsin, void* qf, void* kf, void* vf) {
    if (!wp || !xp || !pos || !rcos || !rsin || !qf || !kf || !vf || nh <= 0 || nkv <= 0 || K % 16 || n_wg <= 0) return LA_E_ARG;
    if (lk_gemm64r_init() != 0) return LA_E_HIP;
    WRAP(lk_gemm64r_qkv((hipStream_t)stream, wp, xp, nh, nkv, K, n_wg, pos, rcos, rsin, qf, kf, vf));
}
int la_gemm64_logits(void* stream, const void* wp, const void* xp, int V, int K, int rb, void* logits, float* cv,
                     int32_t* ci) {
    if (!wp || !xp || !cv || !ci || V % 32 || K % 16 || ((rb & 0xff) != 1 && (rb & 0xff) != 2)) return LA_E_ARG;
    WRAP(lk_gemm64_logits((hipStream_t)stream, wp, xp, V, K, rb, logits, cv, ci));
}
int la_argmax_finalize(void* stream, const float* cv, const int32_t* ci, int n_tiles, int32_t* d_state) {
    if (!cv || !ci || !d_state || n_tiles <= 0) return LA_E_ARG;
    WRAP(lk_argmax_finalize((hipStream_t)stream, cv, ci, n_tiles, d_state + LA_ST_ARGMAX));
}
int la_embed_norm(void* stream, const void* embed, const int32_t* ids, const void* nw, int hidden, float eps, void* h,
                  void* xp) {
    if (!embed || !ids || !nw || !h || !xp) return LA_E_ARG;
    WRAP(lk_embed_norm((hipStream_t)stream, embed, ids, nw, hidden, eps, h, xp));
}
int la_resid_norm(void* stream, void* h, const float* slabs, int n_slabs, const void* nw, int hidden, float eps,
                  void* xp) {
    if (!h || !nw || !xp || n_slabs < 0 || (n_slabs > 0 && !slabs)) return LA_E_ARG;
    WRAP(lk_resid_norm((hipStream_t)stream, h, slabs, n_slabs, nw, hidden, eps, xp));
}
int la_qkv_post(void* stream, const float* slabs, int n_slabs, int nh, int nkv, const int32_t* pos, const void* rcos,
                const void* rsin, void* qf, void* kf, void* vf) {
    if (!slabs || n_slabs < 1 || nh <= 0 || nkv <= 0 || !pos || !rcos || !rsin || !qf || !kf || !vf) return LA_E_ARG;
    WRAP(lk_qkv_post((hipStream_t)stream, slabs, n_slabs, nh, nkv, pos, rcos, rsin, qf, kf, vf));
}
int la_tree_attn(void* stream, const void* qf, const void* km, const void* vm, const void* kf, const void* vf,
                 const uint64_t* rowmask, const int32_t* d_state, int nh, int nkv, int max_keys, int nsplit,
                 float* opart, float* mpart, float* lpart, void* attn_xp) {
    if (!qf || !km || !vm || !kf || !vf || !rowmask || !d_state || !opart || !mpart || !lpart || !attn_xp ||
        nh <= 0 || nkv <= 0 || nh % nkv || max_keys % 32 || nsplit < 1 || nsplit > 64) return LA_E_ARG;
    WRAP(lk_tree_attn((hipStream_t)stream, qf, km, vm, kf, vf, rowmask, d_state, nh, nkv, max_keys, nsplit, opart,
                      mpart, lpart, attn_xp));
}

int la_resid_norm_router(void* stream, void* h, const float* slabs, int n_slabs, const void* nw, int hidden, float eps,
                         void* xp, const void* wr, int n_experts, int top_k, float* route_w, const int32_t* n_rows) {
    if (!h || !slabs || !nw || !xp || !wr || !route_w || !n_rows || n_experts < 1 || n_experts > LA_MOE_MAX_E ||
        top_k < 1 || top_k > n_experts) return LA_E_ARG;
    WRAP(lk_resid_norm_router((hipStream_t)stream, h, slabs, n_slabs, nw, hidden, eps, xp, wr, n_experts, top_k, route_w, n_rows, 1));
}
int la_moe_accum(void* stream, const float* slabs, int n_slabs, const float* route_w, int expert, int hidden, void* acc,
                 int first) {
    if (!slabs || !route_w || !acc || expert < 0 || expert >= LA_MOE_MAX_E) return LA_E_ARG;
    WRAP(lk_moe_accum((hipStream_t)stream, slabs, n_slabs, route_w + expert, hidden, acc, first));
}
int la_resid_norm_addend(void* stream, void* h, const void* addend, const void* nw, int hidden, float eps, void* xp) {
    if (!h || !addend || !nw || !xp) return LA_E_ARG;
    WRAP(lk_resid_norm_addend((hipStream_t)stream, h, addend, nw, hidden, eps, xp, 1));
}

int la_build_batch_inputs(void* stream, const int32_t* d_in, int32_t* d_bstate, int32_t* d_pos, uint64_t* d_rowmask,
                          int32_t* d_ids) {
    if (!d_in || !d_bstate || !d_pos || !d_rowmask || !d_ids) return LA_E_ARG;
    WRAP(lk_build_tree_inputs_b((hipStream_t)stream, d_in, d_bstate, d_pos, d_rowmask, d_ids));
}
int la_accept_scan_batch(void* stream, const int32_t* d_in, const int32_t* d_ids, const uint64_t* d_rowmask,
                         int32_t* d_bstate, int n_slots, int slot_keys) {
    if (!d_in || !d_ids || !d_rowmask || !d_bstate || n_slots < 1 || n_slots > LA_MAX_SEQ || slot_keys % 32) return LA_E_ARG;
    WRAP(lk_accept_scan_b((hipStream_t)stream, d_in, d_ids, d_rowmask, d_bstate, n_slots, slot_keys));
}
int la_kv_commit_batch(void* stream, const void* kf, const void* vf, void* km, void* vm, const int32_t* d_bstate,
                       int n_layers, int nkv, int total_keys) {
    if (!kf || !vf || !km || !vm || !d_bstate || n_layers <= 0 || nkv <= 0 || total_keys % 32) return LA_E_ARG;
    WRAP(lk_kv_commit_b((hipStream_t)stream, kf, vf, km, vm, d_bstate, n_layers, nkv, total_keys));
}
int la_tree_attn_batch(void* stream, const void* qf, const void* km, const void* vm, const void* kf, const void* vf,
                       const uint64_t* rowmask, const int32_t* d_bstate, int nh, int nkv, int slot_keys, int n_slots,
                       int nsplit, float* opart, float* mpart, float* lpart, void* attn_xp) {
    if (!qf || !km || !vm || !kf || !vf || !rowmask || !d_bstate || !opart || !mpart || !lpart || !attn_xp ||
        nh <= 0 || nkv <= 0 || nh % nkv || slot_keys % 32 || nsplit < 1 || nsplit > 64 || n_slots < 1 ||
        n_slots > LA_MAX_SEQ) return LA_E_ARG;
    WRAP(lk_tree_attn_b((hipStream_t)stream, qf, km, vm, kf, vf, rowmask, d_bstate, nh, nkv, slot_keys, n_slots, nsplit,
                        opart, mpart, lpart, attn_xp));
}

int la_trie_hier_get_dev(void* stream, const int32_t* d_tok, const double* d_fo, const double* d_fi, const int32_t* d_cstart,
                         const int32_t* d_ccount, int n_nodes, const int32_t* d_queries, const int32_t* d_nq, int B,
                         int decoding_length, int branch_length, int min_input_size, int min_output_size, int mode,
                         const int32_t* d_stop, int n_stop, int32_t* d_scratch_q, double* d_scratch_v, int32_t* d_out_ids,
                         uint64_t* d_out_rowmask, int32_t* d_out_n, int32_t* d_out_sizes, int32_t* d_out_nsizes) {
    if (!d_tok || !d_fo || !d_fi || !d_cstart || !d_ccount || n_nodes < 1 || !d_queries || !d_nq || B < 1 ||
        !d_scratch_q || !d_scratch_v || !d_out_ids || !d_out_rowmask || !d_out_n || !d_out_sizes || !d_out_nsizes ||
        mode < 0 || mode > 2 || (n_stop > 0 && !d_stop)) return LA_E_ARG;
    if (decoding_length > LA_TREE_MAX) { la_set_error("device hier_get handles decoding_length <= 64"); return LA_E_RANGE; }
    WRAP(lk_trie_hier_get((hipStream_t)stream, d_tok, d_fo, d_fi, d_cstart, d_ccount, n_nodes, d_queries, d_nq, B,
                          decoding_length, branch_length, min_input_size, min_output_size, mode, d_stop, n_stop,
                          d_scratch_q, d_scratch_v, d_out_ids, d_out_rowmask, d_out_n, d_out_sizes, d_out_nsizes));
}

int la_trie_patch_dev(void* stream, int32_t* d_tok, double* d_fo, double* d_fi, int64_t fi_stride, int32_t* d_cstart,
                      int32_t* d_ccount, int32_t* d_ccap, const int32_t* d_ipatch, int n_i, const int32_t* d_dkey, const double* d_dval,
                      int n_d) {
    if (!d_tok || !d_fo || !d_cstart || !d_ccount || n_i < 0 || n_d < 0 || (n_i > 0 && !d_ipatch) || (n_d > 0 && (!d_dkey || !d_dval)))
        return LA_E_ARG;
    WRAP(lk_trie_patch((hipStream_t)stream, d_tok, d_fo, d_fi, (long)fi_stride, d_cstart, d_ccount, d_ccap, d_ipatch, n_i, d_dkey,
                       d_dval, n_d));
}

int la_trie_one_get_dev2(void* stream, const int32_t* d_tok, const double* d_fo, const double* d_fi, int64_t fi_stride,
                         const int32_t* d_cstart, const int32_t* d_ccount, int32_t n_records, const int32_t* d_queries,
                         const int32_t* d_nq, const int32_t* d_plane, const int32_t* d_branch_length, int B, int decoding_length,
                         int branch_length, int mode, const int32_t* d_stop, int n_stop, int32_t* d_out_ids, uint64_t* d_out_rowmask,
                         int32_t* d_out_n, int32_t* d_out_sizes, int32_t* d_out_nsizes) {
    if (!d_tok || !d_fo || !d_fi || !d_cstart || !d_ccount || n_records < 1 || !d_queries || !d_nq || B < 1 || !d_out_ids ||
        !d_out_rowmask || !d_out_n || !d_out_sizes || !d_out_nsizes || mode < 0 || mode > 2 || (n_stop > 0 && !d_stop)) return LA_E_ARG;
    if (branch_length + 1 > LA_TREE_MAX) { la_set_error("device one_get: branch_length + 1 <= 64"); return LA_E_RANGE; }
    WRAP(lk_trie_one_get2((hipStream_t)stream, d_tok, d_fo, d_fi, (long)fi_stride, d_cstart, d_ccount, n_records, d_queries, d_nq,
                          d_plane, d_branch_length, B, decoding_length, branch_length, mode, d_stop, n_stop, d_out_ids,
                          d_out_rowmask, d_out_n, d_out_sizes, d_out_nsizes));
}

static int trie_image_ok(const la_trie_image* g) {
    return g && g->tok && g->fo && g->cstart && g->ccount && g->ccap && g->meta && g->cap > 0 && g->n_planes >= 0 &&
           (g->n_planes == 0 || g->fi) && g->n_root_of >= 0 && (g->n_root_of == 0 || g->root_of);
}

int la_trie_root_index_dev(void* stream, const la_trie_image* img, int n_roots_max) {
    if (!trie_image_ok(img) || n_roots_max < 0) return LA_E_ARG;
    if (img->n_root_of == 0) return LA_OK;
    WRAP(lk_trie_root_index((hipStream_t)stream, img->tok, img->cstart, img->ccount, img->root_of, img->n_root_of, n_roots_max));
}

int la_trie_stream_put_dev(void* stream, const la_trie_image* img, int32_t* d_obuf, int32_t* d_olen, const int32_t* d_src_tok,
                           int src_stride, const int32_t* d_src_cnt, const int32_t* d_put_idx, int n_put, int branch_length,
                           const int32_t* d_stop, int n_stop, const int32_t* d_eos, int n_eos, int32_t* d_items) {
    if (!trie_image_ok(img) || !d_obuf || !d_olen || !d_src_tok || !d_src_cnt || !d_put_idx || !d_items || src_stride < 1 ||
        n_put < 0 || n_stop < 0 || n_eos < 0 || (n_stop > 0 && !d_stop) || (n_eos > 0 && !d_eos)) return LA_E_ARG;
    if (n_put > LA_TRIE_PUTS || branch_length < 1 || branch_length > 64) {
        la_set_error("device stream_put: at most 64 puts per call, branch_length 1..64"); return LA_E_RANGE;
    }
    if (n_put == 0) return LA_OK;
    TriePutArgs a{};
    a.tok = img->tok; a.fo = img->fo; a.fi = img->fi; a.fi_stride = (long)img->fi_stride; a.n_planes = img->n_planes;
    a.cstart = img->cstart; a.ccount = img->ccount; a.ccap = img->ccap; a.meta = img->meta; a.cap = img->cap;
    a.root_of = img->root_of; a.n_root_of = img->n_root_of;
    a.obuf = d_obuf; a.olen = d_olen; a.src_tok = d_src_tok; a.src_stride = src_stride; a.src_cnt = d_src_cnt;
    a.put_idx = d_put_idx; a.n_put = n_put; a.branch_length = branch_length;
    a.stop = d_stop; a.n_stop = n_stop; a.eos = d_eos; a.n_eos = n_eos; a.items = d_items;
    WRAP(lk_trie_stream_put((hipStream_t)stream, a));
}

int la_trie_hier_get_dev2(void* stream, const int32_t* d_tok, const double* d_fo, const double* d_fi, int64_t fi_stride,
                          const int32_t* d_cstart, const int32_t* d_ccount, int32_t n_records, const int32_t* d_queries,
                          const int32_t* d_nq, const int32_t* d_plane, const int32_t* d_branch_length, int B, int decoding_length,
                          int branch_length, int min_in, int min_out, int mode, const int32_t* d_stop, int n_stop,
                          int32_t* d_scratch_q, double* d_scratch_v, int32_t* d_out_ids, uint64_t* d_out_rowmask, int32_t* d_out_n,
                          int32_t* d_out_sizes, int32_t* d_out_nsizes) {
    if (!d_tok || !d_fo || !d_fi || !d_cstart || !d_ccount || n_records < 1 || !d_queries || !d_nq || B < 1 ||
        !d_scratch_q || !d_scratch_v || !d_out_ids || !d_out_rowmask || !d_out_n || !d_out_sizes || !d_out_nsizes ||
        mode < 0 || mode > 2 || (n_stop > 0 && !d_stop)) return LA_E_ARG;
    if (decoding_length > LA_TREE_MAX) { la_set_error("device hier_get handles decoding_length <= 64"); return LA_E_RANGE; }
    WRAP(lk_trie_hier_get2((hipStream_t)stream, d_tok, d_fo, d_fi, (long)fi_stride, d_cstart, d_ccount, n_records, d_queries, d_nq,
                           d_plane, d_branch_length, B, decoding_length, branch_length, min_in, min_out, mode, d_stop, n_stop,
                           d_scratch_q, d_scratch_v, d_out_ids, d_out_rowmask, d_out_n, d_out_sizes, d_out_nsizes));
}

int la_trie_hier_get_wg(void* stream, const la_trie_query* q) {
    if (!q || !q->tok || !q->fo || !q->fi || !q->cstart || !q->ccount || q->n_records < 1 || !q->queries || !q->nq || q->B < 1 ||
        !q->scratch_i || !q->scratch_v || !q->out_ids || !q->out_rowmask || !q->out_n || !q->out_sizes || !q->out_nsizes ||
        q->mode < 0 || q->mode > 2 || (q->n_stop > 0 && !q->stop) || (q->root_of && q->n_root_of < 1)) return LA_E_ARG;
    if (q->decoding_length > LA_TREE_WIDE_MAX || q->mask_words < 1 || q->mask_words > 4 || q->row_stride < 1 ||
        q->decoding_length > q->row_stride || q->decoding_length > 64 * q->mask_words) {
        la_set_error("la_trie_hier_get_wg: decoding_length <= min(256, row_stride, 64 * mask_words), mask_words <= 4");
        return LA_E_RANGE;
    }
    TrieWgArgs a{};
    a.q.t = TrieDev{q->tok, q->fo, q->fi, q->cstart, q->ccount, q->n_records};
    a.q.plane = q->plane; a.q.fi_stride = (long)q->fi_stride; a.q.bl = q->branch_lengths;
    a.q.queries = q->queries; a.q.nq = q->nq; a.q.decoding_length = q->decoding_length; a.q.branch_length = q->branch_length;
    a.q.min_in = q->min_in; a.q.min_out = q->min_out; a.q.mode = q->mode; a.q.stop = q->stop; a.q.n_stop = q->n_stop;
    a.q.out_ids = q->out_ids; a.q.out_rowmask = (unsigned long long*)q->out_rowmask; a.q.out_n = q->out_n; a.q.out_sizes = q->out_sizes;
    a.q.out_nsizes = q->out_nsizes;
    a.root_of = q->root_of; a.n_root_of = q->n_root_of; a.scr_i = q->scratch_i; a.scr_v = q->scratch_v;
    a.row_stride = q->row_stride; a.mask_words = q->mask_words;
    a.lcap = q->lds_level_cap; a.mcap = q->lds_cand_cap; a.onewave = q->one_wave_cap;
    WRAP(lk_trie_hier_get_wg((hipStream_t)stream, a, q->B));
}

}  // extern "C"
