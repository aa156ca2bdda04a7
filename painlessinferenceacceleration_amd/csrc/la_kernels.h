// la_kernels.h — host-callable launchers of the gfx950 kernels in la_kernels.hip (internal; the
// public surface is include/lookahead_hip.h).  All return 0 or a hipError_t value.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lookahead_hip.h"

// residual + RMSNorm row stage fused into the consuming balanced GEMM (producer workgroups 0..63, in-kernel hand-over)
struct FusedNorm {
    const float* slabs; int n_slabs;     // split-K partial sums of the preceding projection
    void* h; const void* nw;             // residual stream (updated in place), norm weight
    int hidden; float eps; int cast_first;
    int* counter;                        // zeroed device int, one per fused launch and step
    int write_through;                   // 1: producers publish with sc1 (write-through) stores + drained flag instead of a release fence
};

// Idle-window weight prefetch.  The row kernels (k_row_norm) and the attention combine stream almost nothing: HBM is idle while
// they run, and the GEMM that follows starts cold.  Extra workgroups appended to those launches (block ids past the kernel's own)
// read the first k-tiles every workgroup of the NEXT GEMM will stream — default cache policy, so the lines stay in the L2 of the
// XCD whose CUs will ask for them (block b of a grid runs on XCD b % 8; the extra ids keep that residue) — and drop the data.
// A descriptor names those bytes: consumer workgroup (bx, ks) / wave w / row-block rb starts at
// base + bx*A + ks*A2 + boff[rb] + w*C[rb] and the first L[rb] bytes of that run are fetched.  Bit-identical by construction
// (nothing is written); la_debug_set keys 7 (KiB per consumer workgroup, 0 = off) and 8 (start delay) drive it.
struct PfDesc {
    const char* base;         // null: no prefetch workgroups are appended
    int n_consumers;          // workgroups of the consuming launch (= appended workgroups)
    int nbx;                  // consumer gridDim.x (ks = b / nbx)
    unsigned A, A2;           // bytes per bx / per ks step
    unsigned boff[4], C[4], L[4];
    int RB, NW;
    int delay;                // s_sleep(32) rounds before the first load (lets the kernel's own loads go first)
    unsigned magic;           // value the xor of the fetched data is compared with (never equal in practice): keeps the loads alive
    int* sink;
};
void lk_pf_planned(PfDesc* d, const void* wp, int kind, int n_rows, int K, int n_wg, int kib, int delay, int* sink);
void lk_pf_classic(PfDesc* d, const void* wp, int N, int K, int rbv, int ksplit, int kib, int delay, int* sink);
int lk_pf_only(hipStream_t st, const PfDesc* pf);     // the prefetch workgroups of a descriptor as their own launch (forked graph branch)

int lk_pack_weight(hipStream_t st, const void* w, const void* w2, int N, int K, int interleave2, void* out);
int lk_pack_x(hipStream_t st, const void* x, int K, void* out);
int lk_gemm64_slab(hipStream_t st, const void* wp, const void* xp, int N, int K, int rb, int ksplit, float* slabs,
                   const float* route_col = nullptr);
int lk_gemm64_swiglu(hipStream_t st, const void* wp, const void* xp, int F, int K, void* act_xp, int variant,
                     const float* route_col = nullptr);
int lk_gemm64_logits(hipStream_t st, const void* wp, const void* xp, int V, int K, int rb, void* logits, float* cv, int* ci);
int lk_logits_cand_slots(int V, int rbv);
int lk_gemm64_qkv(hipStream_t st, const void* wp, const void* xp, int nh, int nkv, int K, const int* pos, const void* rcos,
                  const void* rsin, void* qf, void* kfresh, void* vfresh, int variant);
void lk_qkv_row_perm(int nh, int nkv, int* perm);
int lk_gemm64r_init();
int lk_step_head(hipStream_t st, const int* in, int* state, int* pos, uint64_t* rowmask, int* ids, const void* embed, const void* nw,
                 int hidden, float eps, void* h, void* xp, int cast_first, const PfDesc* pf, uint64_t* gran = nullptr, int n_gran = 0);
int lk_resid_norm4(hipStream_t st, void* h, const float* slabs, int n_slabs, const void* nw, int hidden, float eps, void* xp,
                   int cast_first, uint64_t* gran);
int lk_step_tail(hipStream_t st, const float* cv, const int* ci, int n_tiles, const int* ids, const uint64_t* rowmask, int* state,
                 int* host_out);
int lk_gateup_down(hipStream_t st, const void* wgu, const void* xp, int F, int K, int n_wg, void* act_xp, const void* wdown, int N,
                   int ksplit, float* slabs, int* counter, int dd, const FusedNorm* fn = nullptr);
int lk_rowplan(int kind, int n_rows, int n_wg, int* out);
long lk_planned_elems(int kind, int n_rows, int K, int n_wg);
int lk_pack_planned(hipStream_t st, const void* w, const void* w2, const int* d_plan, int kind, int n_rows, int K, int n_wg, void* out);
int lk_gemm64r_swiglu(hipStream_t st, const void* wp, const void* xp, int F, int K, int n_wg, void* act_xp,
                      const float* route_col = nullptr, const FusedNorm* fn = nullptr, const PfDesc* pf = nullptr);
int lk_gemm64r_logits(hipStream_t st, const void* wp, const void* xp, int V, int K, int n_wg, void* logits, float* cv, int* ci);
int lk_gemm64r_qkv(hipStream_t st, const void* wp, const void* xp, int nh, int nkv, int K, int n_wg, const int* pos,
                   const void* rcos, const void* rsin, void* qf, void* kfresh, void* vfresh, const FusedNorm* fn = nullptr);
int lk_argmax_finalize(hipStream_t st, const float* cv, const int* ci, int n_tiles, int* out_rows);
int lk_embed_norm(hipStream_t st, const void* embed, const int* ids, const void* nw, int hidden, float eps, void* h, void* xp,
                  int cast_first = 0, const PfDesc* pf = nullptr);
int lk_resid_norm(hipStream_t st, void* h, const float* slabs, int n_slabs, const void* nw, int hidden, float eps, void* xp,
                  int cast_first = 0, const PfDesc* pf = nullptr);
int lk_resid_norm_router(hipStream_t st, void* h, const float* slabs, int n_slabs, const void* nw, int hidden, float eps,
                         void* xp, const void* wrouter, int n_experts, int top_k, float* route_w, const int* n_rows,
                         int cast_first = 0);
int lk_resid_norm_addend(hipStream_t st, void* h, const void* addend, const void* nw, int hidden, float eps, void* xp,
                         int cast_first = 0);
int lk_moe_accum(hipStream_t st, const float* slabs, int n_slabs, const float* route_col, int hidden, void* acc, int first);
int lk_build_tree_inputs(hipStream_t st, const int* in, int* state, int* pos, uint64_t* rowmask, int* ids);
int lk_qkv_post(hipStream_t st, const float* slabs, int n_slabs, int nh, int nkv, const int* pos, const void* rcos,
                const void* rsin, void* qf, void* kfresh, void* vfresh);
int lk_tree_attn(hipStream_t st, const void* qf, const void* kmain, const void* vmain, const void* kfresh,
                 const void* vfresh, const uint64_t* rowmask, const int* state, int nh, int nkv, int max_keys,
                 int nsplit, float* opart, float* mpart, float* lpart, void* attn_xp, int window = 0, int ring_keys = 0,
                 const PfDesc* pf = nullptr, int form = -1, const PfDesc* ride = nullptr, int head_dim = 128);      // form: 0 = key splits + combine, -1 = the default (one launch, la_attn1.hip) unless a lab knob says otherwise
int lk_attn1_init();
// 0 if the softmax scale of this build (attn_scale, la_common.h) is EXACT for head_dim: the fp16 build divides (always exact); the bf16 build
// multiplies by fp32(1 / sqrt(head_dim)), which this routine checks against the correctly rounded quotient for every finite bf16 value
int lk_qk_scale_check(int head_dim);
// lab (round 6, knob 33): o_proj that merges the key-split attention partials while it builds its x operand (la_oproj_merge.hip)
int lk_oproj_merge(hipStream_t st, const void* wp, int N, int K, int ksplit, int nsplit, const float* opart, const float* mpart,
                   const float* lpart, float* slabs);
// single-sequence step, ONE launch (la_attn1.hip): no key-split partials, no combine kernel
int lk_tree_attn1(hipStream_t st, const void* qf, const void* kmain, const void* vmain, const void* kfresh, const void* vfresh,
                  const uint64_t* rowmask, const int* state, int nh, int nkv, int max_keys, void* attn_xp, int window, int ring_keys,
                  const PfDesc* pf = nullptr, int head_dim = 128);      // pf: weight-prefetch riders for the next launch (o_proj) on the CUs this one leaves idle
int lk_tree_attn_b(hipStream_t st, const void* qf, const void* kmain, const void* vmain, const void* kfresh,
                   const void* vfresh, const uint64_t* rowmask, const int* bstate, int nh, int nkv, int slot_keys,
                   int n_slots, int nsplit, float* opart, float* mpart, float* lpart, void* attn_xp, int window = 0, int ring_keys = 0,
                   const PfDesc* pf = nullptr, int head_dim = 128);
int lk_build_tree_inputs_b(hipStream_t st, const int* in, int* bstate, int* pos, uint64_t* rowmask, int* ids);
int lk_accept_scan_b(hipStream_t st, const int* in, const int* ids, const uint64_t* rowmask, int* bstate, int n_slots,
                     int slot_keys, int ring = 0);
int lk_kv_commit_b(hipStream_t st, const void* kfresh, const void* vfresh, void* kmain, void* vmain, const int* bstate,
                   int n_layers, int nkv, int total_keys);
int lk_gemm64r_swiglu_ex(hipStream_t st, const void* wp0, long w_stride, const void* xp, int F, int K, int n_wg, void* act0,
                         long act_stride, const float* route_w, int E);
int lk_gemm64_swiglu_ex(hipStream_t st, const void* wp0, long w_stride, const void* xp, int F, int K, void* act0, long act_stride,
                        const float* route_w, int E);
int lk_gemm64_slab_ex(hipStream_t st, const void* wp0, long w_stride, const void* xp0, long x_stride, int N, int K, int rbv, int ksplit,
                      float* slabs0, long slab_stride, const float* route_w, int E);
int lk_moe_accum_all(hipStream_t st, const float* slabs0, long slab_stride, int n_slabs, const float* route_w, int E, int hidden, void* acc);
int lk_publish(hipStream_t st, int* state, int* host_out);
int lk_accept_scan(hipStream_t st, const int* ids, const uint64_t* rowmask, int* state);
int lk_kv_commit(hipStream_t st, const void* kfresh, const void* vfresh, void* kmain, void* vmain, const int* state,
                 int n_layers, int nkv, int max_keys, int ring = 0);
int lk_trie_hier_get(hipStream_t st, const int* tok, const double* fo, const double* fi, const int* cstart, const int* ccount,
                     int n_nodes, const int* queries, const int* nq, int B, int decoding_length, int branch_length,
                     int min_in, int min_out, int mode, const int* stop, int n_stop, int* scratch_q, double* scratch_v,
                     int* out_ids, uint64_t* out_rowmask, int* out_n, int* out_sizes, int* out_nsizes);
int lk_trie_patch(hipStream_t st, int* tok, double* fo, double* fi, long fi_stride, int* cstart, int* ccount, int* ccap,
                  const int* ipatch, int n_i, const int* dkey, const double* dval, int n_d);
int lk_trie_one_get2(hipStream_t st, const int* tok, const double* fo, const double* fi, long fi_stride, const int* cstart,
                     const int* ccount, int n_nodes, const int* queries, const int* nq, const int* plane, const int* bl, int B,
                     int decoding_length, int branch_length, int mode, const int* stop, int n_stop, int* out_ids,
                     uint64_t* out_rowmask, int* out_n, int* out_sizes, int* out_nsizes);
// device-side stream_put (la_trie_dev.hip)
#define LA_TRIE_OBUF 128
#define LA_TRIE_ITEMS 40          // start offsets one put can complete = tokens it appends (<= LA_MOUT_TOKS)
#define LA_TRIE_PUTS 64
struct TriePutArgs {
    int* tok; double* fo; double* fi; long fi_stride; int n_planes;
    int* cstart; int* ccount; int* ccap;
    int* meta;                    // [0] records in use  [1] overflow (sticky)  [2] branches inserted  [3] records appended
    int cap;
    int* root_of; int n_root_of;  // token -> record of its tree root (-1: none); tokens >= n_root_of fall back to the ballot search
    int* obuf; int* olen;         // [n_idx][LA_TRIE_OBUF], [n_idx]
    const int* src_tok; int src_stride; const int* src_cnt;     // put k appends src_tok[k * src_stride ..][0 .. src_cnt[k])
    const int* put_idx; int n_put, branch_length;
    const int* stop; int n_stop; const int* eos; int n_eos;
    int* items;                   // [n_put][LA_TRIE_ITEMS][2]
};
int lk_trie_root_index(hipStream_t st, const int* tok, const int* cstart, const int* ccount, int* root_of, int n_root_of, int n_roots_max);
int lk_trie_stream_put(hipStream_t st, const TriePutArgs& a);
int lk_trie_hier_get2(hipStream_t st, const int* tok, const double* fo, const double* fi, long fi_stride, const int* cstart,
                      const int* ccount, int n_nodes, const int* queries, const int* nq, const int* plane, const int* bl, int B,
                      int decoding_length, int branch_length, int min_in, int min_out, int mode, const int* stop, int n_stop,
                      int* scratch_q, double* scratch_v, int* out_ids, uint64_t* out_rowmask, int* out_n, int* out_sizes,
                      int* out_nsizes);
