// la_mblock.h — multi-block verify step (B sequences x 64 tree rows, or one prompt as a chain of 64-row blocks):
// launchers of the kernels in la_mblock.hip (internal; the public surface is include/lookahead_hip.h).
//
// Layouts (extends la_common.h): every activation matrix of a multi-block step is the concatenation of `nblk` 64-row
// images of the single-block layout — XP [blk][k-tile][token block][lane][8], residual stream row-major [blk*64 + t][hidden],
// Q fragments [blk][head][2][8][512], fresh K/V tiles per layer [blk][kv head][2][4096], split-K slabs [ks][blk*64 + t][N].
// Weights are the SAME packed images the single-block kernels stream (la_pack_weight / la_pack_planned): no second copy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lookahead_hip.h"

// device block-meta words (one record of LA_MB_META ints per block, written by k_build_inputs_mb)
#define LA_MB_META      8
#define LA_MBM_SLOT     0    // sequence slot (KV region) of the block
#define LA_MBM_T        1    // valid rows
#define LA_MBM_MODE     2    // 0 = verify tree, 1 = prefill chain (commit all rows), 2 = forward only, 3 = later piece of a wide tree
#define LA_MBM_LIMIT    3    // max tokens to emit
#define LA_MBM_NKEYS    4    // committed keys of the slot when the step started
#define LA_MBM_BASE     5    // rows of earlier blocks of the same slot in this step (prefill chains)
#define LA_MBM_FIRST    6    // index of the first block of the same slot in this step

int lk_mb_init();
int lk_mb_fill_from_trie(hipStream_t st, const int* t_ids, const uint64_t* t_rm, const int* t_n, const int* slots, const int* limits,
                         const int* last, int nblk, int* d_in);
int lk_mb_build_inputs(hipStream_t st, const int* d_in, const int* d_bstate, int nblk, int* d_meta, int* d_pos,
                       uint64_t* d_rowmask, int* d_ids);
int lk_mb_embed_norm(hipStream_t st, const void* embed, const int* ids, const void* nw, int hidden, float eps, void* h, void* xp,
                     int M, int cast_first);
int lk_mb_resid_norm(hipStream_t st, void* h, const float* slabs, int n_slabs, int slab_rows, const void* nw, int hidden, float eps,
                     void* xp, int M, int cast_first);
// kind: 0 = o/down split-K slabs (classic packed image), 1 = gate/up + SwiGLU (planned), 2 = QKV + RoPE (planned), 3 = lm_head (planned)
struct MbGemm {
    const void* wp; const void* xp;
    int N, K, nblk, n_wg, ksplit;
    float* slabs; int slab_rows;      // slab stride in rows per K split (>= rows written: whole passes of 4 blocks)
    void* act_xp;
    void* logits; float* cand_val; int* cand_idx;
    const int* pos; const void* rcos; const void* rsin; void* qf; void* kfresh; void* vfresh; int nh, nkv;
    const float* route_col;            // MoE: this expert's routing weights (stride LA_MOE_MAX_E floats per row); null = dense
    const int* nblk_dev;               // gathered MoE: the expert's block count on the device (nblk = the upper bound); null = nblk
    // gathered MoE, ALL experts of a stage in one launch (ex_n > 1, needs nblk_dev): expert e reads wp + e * ex_w_stride, xp + e *
    // ex_x_stride (bf16 elements), writes act_xp + e * ex_o_stride (bf16 elements) / slabs + e * ex_o_stride (floats), count nblk_dev[e]
    int ex_n; long ex_w_stride, ex_x_stride, ex_o_stride;
};
int lk_mb_gemm(hipStream_t st, int kind, const MbGemm& g);
int lk_mb_resid_norm_router(hipStream_t st, void* h, const float* slabs, int n_slabs, int slab_rows, const void* nw, int hidden, float eps,
                            void* xp, int M, int cast_first, const void* wrouter, int n_experts, int top_k, float* route_w, const int* meta);
int lk_mb_resid_norm_addend(hipStream_t st, void* h, const void* addend, const void* nw, int hidden, float eps, void* xp, int M, int cast_first);
int lk_mb_moe_accum_norm(hipStream_t st, const float* slabs0, long slab_stride, int n_slabs, int slab_rows, const float* route_w, int E,
                         int hidden, int M, const int* pos, void* h, const void* nw, float eps, void* xp, int cast_first);
int lk_mb_moe_accum(hipStream_t st, const float* slabs0, long slab_stride, int n_slabs, int slab_rows, const float* route_w, int E, int hidden,
                    void* acc, int M, const int* pos);
// gathered MoE: perm [E][LA_MB_MAX*64], pos [M][LA_MOE_MAX_E], cnt_nb [2][LA_MOE_MAX_E] = {rows, 64-row blocks} per expert
int lk_mb_moe_plan(hipStream_t st, const float* route_w, int M, int E, int* perm, int* pos, int* cnt_nb);
int lk_mb_moe_gather(hipStream_t st, const void* xp, const int* perm, const int* cnt_nb, int hidden, int nblk, int E, void* xg, long xg_stride);
int lk_mb_moe_plan_gather(hipStream_t st, const float* route_w, int M, const void* xp, int hidden, int nblk, int E, void* xg, long xg_stride,
                          int* perm, int* pos, int* cnt_nb);
int lk_mb_cand_slots(int n_wg);
int lk_mb_logits_wgs(int V, int n_wg);
int lk_mb_argmax(hipStream_t st, const float* cv, const int* ci, int n_tiles, int nblk, int* out_rows);
int lk_mb_tree_attn(hipStream_t st, const void* qf, const void* kmain, const void* vmain, const void* kfresh, const void* vfresh,
                    const uint64_t* rowmask, const int* meta, int nblk, int nh, int nkv, int slot_keys, int n_slots, int nsplit,
                    float* opart, float* mpart, float* lpart, void* attn_xp, int window, int ring, const uint64_t* xmask, int head_dim = 128);
int lk_mb_accept_scan(hipStream_t st, const int* meta, const int* ids, const uint64_t* rowmask, const uint64_t* xmask, const int* argmax, int nblk,
                      int slot_keys, int ring, int* bstate, int* d_out);
int lk_mb_kv_commit(hipStream_t st, const void* kfresh, const void* vfresh, void* kmain, void* vmain, const int* d_out, int nblk,
                    int n_layers, int nkv, int total_keys);
