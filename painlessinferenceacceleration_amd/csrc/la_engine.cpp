// la_engine.cpp — the whole verify step of a Llama-family model as one hipGraph on one stream.
// Replaces LlamaForCausalLM.forward under the rank-4 mask hook (modeling_llama.py:544-677, 710-794),
// the accept scan (pretrained_model.py:764-892) and the KV compaction (:894-907).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <chrono>
#include <atomic>
#include "la_kernels.h"
#include "la_mblock.h"
#include "la_knobs.h"
extern int g_la_graph_epoch;
int g_la_stop_layers = 0;     // la_debug_set key 13 (parity tests): the single-sequence step runs only the first n layers, then the final norm + lm_head

extern void la_set_error(const std::string& s);

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    la_set_error(std::string(#x) + ": " + hipGetErrorString(e_)); return LA_E_HIP; } } while (0)
#define KCHK(x) do { int e_ = (x); if (e_ != 0) { \
    la_set_error(std::string(#x) + ": launch error " + std::to_string(e_)); return LA_E_HIP; } } while (0)

struct la_llama {
    la_llama_config cfg;
    std::vector<la_llama_layer_weights> layers;
    std::vector<const void*> ex_gateup, ex_down;     // [n_layers][n_experts] copies of the caller's pointer arrays
    uint16_t* moe_acc;
    float* route_w;
    // merged expert launches (all experts of a layer in one grid) when every layer's expert images are equally spaced
    bool ex_merged;
    std::vector<long> ex_gu_stride, ex_dn_stride;     // per layer, in bf16 elements
    uint16_t* act_ex;       // [E][64][ffn] packed SwiGLU outputs
    float* slabs_ex;        // [E][down_ks][64][hidden]
    uint64_t* norm_gran;  // [2 * n_layers][64 rows][4] {tag, partial sum of squares} granules of the k_row_norm4 launches of one step
    int* fuse_cnt;      // [2 * n_layers] hand-over counters of the fused norm->GEMM launches (zeroed every step)
    int fuse;           // bit 0: post-attention norm -> gate/up GEMM, bit 1: input norm of layer l>0 -> QKV GEMM
    la_llama_weights w;
    // derived
    int qkv_n, o_k, nsplit;
    int qkv_rb, qkv_ks, o_rb, o_ks, down_rb, down_ks, lm_rb, gu_variant;
    bool qkv_fused;
    // device buffers (carved from the caller's workspace)
    char* ws;
    uint16_t *kmain, *vmain, *kfresh, *vfresh, *qf, *h, *xp, *attn_xp, *act_xp, *logits;
    float *slabs, *opart, *mpart, *lpart, *cand_val;
    int *cand_idx, *state, *in, *pos, *ids, *bstate, *bin;
    uint64_t* rowmask;
    size_t kv_layer_elems, fresh_layer_elems;
    int n_slots, total_keys;
    // multi-block step (cfg.max_blocks > 1): activations of up to max_blocks x 64 rows
    int mb_max, mb_nsplit;
    uint16_t *mb_h, *mb_xp, *mb_attn_xp, *mb_act, *mb_logits, *mb_qf, *mb_kfresh, *mb_vfresh;
    float *mb_slabs, *mb_opart, *mb_mpart, *mb_lpart, *mb_cand_val;
    int *mb_cand_idx, *mb_meta, *mb_pos, *mb_ids, *mb_in, *mb_out;
    uint16_t *mb_moe_acc, *mb_act_ex;     // MoE: accumulated expert outputs [M][hidden], per-expert SwiGLU outputs [E][M x ffn]
    float *mb_route_w, *mb_slabs_ex;      // routing weights [M][LA_MOE_MAX_E], per-expert down-projection slabs [E][ks][M][hidden]
    // gathered MoE (nblk >= 2): rows per expert [E][512], position of a row in each expert's list [M][8], {rows, blocks} per expert,
    // per-expert packed inputs [E][M x hidden]
    int *mb_moe_perm, *mb_moe_pos, *mb_moe_cnt;
    uint16_t* mb_xg;
    uint64_t* mb_rowmask;
    size_t mb_fresh_layer;
    hipGraphExec_t mgraphs[2 * (LA_MB_MAX + 1)];     // [wide][nblk]: passes with wide-tree pieces use the masked attention instantiation
    bool mready[2 * (LA_MB_MAX + 1)];
    int mepoch[2 * (LA_MB_MAX + 1)];     // g_la_graph_epoch at capture time, per multi-block graph
    hipGraphExec_t graph_exec, bgraph_exec;      // bgraph_exec: scratch slot used while capturing a batch variant
    hipGraphExec_t graph_long = nullptr;         // the single-sequence step with the key-split attention (contexts past attn_thr)
    bool graph_long_ready = false;
    int attn_thr = 0;                            // committed keys + 64 up to which the single-launch attention is used (see resolve_cfg)
    int ex_down_ks = 4;                          // K splits of the gathered experts' down projection (multi-block MoE step)
    hipGraphExec_t bgraphs[4];                  // batch step captured per attention key-split count {8, 4, 2, 1}
    bool bready[4];
    int bepoch[4];
    const int32_t* zc_in;      // pinned host blocks the captured single-sequence graph reads / writes (zero-copy)
    int32_t* zc_out;
    int seq_expected;          // value host_out[LA_ST_SEQ] takes when the last launched step has been published
    bool graph_ready, bgraph_ready;
    int graph_epoch = 0;       // g_la_graph_epoch at the time the single-sequence step graph was captured
    hipStream_t graph_stream;
    // forked weight prefetch (round 6): a low-priority side stream that becomes a parallel branch of the captured single-sequence
    // step (k_pf_only launches under the latency-bound kernels), and the two events that fork / join it
    hipStream_t pf_stream = nullptr;
    hipEvent_t pf_fork = nullptr, pf_join = nullptr;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Carver {
    char* base; size_t off;
    template <typename T> T* take(size_t n) {
        off = align_up(off, 256);
        T* p = base ? (T*)(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

static void resolve_cfg(la_llama* m) {
    const la_llama_config& c = m->cfg;
    // every head occupies a 128-feature LANE of the Q / K / V fragment layouts and of o_proj's K dimension, whatever cfg.head_dim is: a
    // model with head_dim < 128 arrives with its projections padded to the lane (zero rows / columns at pack time, la_llama_config), the
    // kernels see 128 everywhere and only the softmax scale (la_qk_scale) knows the real head_dim
    m->qkv_n = (c.n_heads + 2 * c.n_kv_heads) * 128;
    m->o_k = c.n_heads * 128;
    // key splits of the tree attention: heads x splits workgroups should fit ONE wave over the CUs (32 heads -> 8, 40 -> 6, 64 -> 4)
    {
        const int cus = c.balanced_wg[1] > 0 ? c.balanced_wg[1] : 256;
        const int k = c.n_heads > 0 ? cus / c.n_heads : 8;
        const int auto_split = k >= 8 ? 8 : k >= 6 ? 6 : k >= 4 ? 4 : k >= 1 ? k : 1;       // 5 and 7 have no combine instantiation
        m->nsplit = c.attn_split > 0 ? c.attn_split : auto_split;
    }
    auto pick = [](int v, int d) { return v > 0 ? v : d; };
    // defaults from scripts/gpu_tune.py on MI355X (Llama-2-7B shapes): per-CU balanced grids (multiples of 256
    // workgroups) beat everything else; see DESIGN.md section 4
    m->qkv_rb = pick(c.gemm_cfg[0], 2);
    m->qkv_ks = pick(c.gemm_cfg[1], 1);
    // K splits of the slab GEMMs (o_proj, down_proj: N = hidden): as many as keep hidden / 64 row-blocks x splits within ONE wave of
    // workgroups over the CUs — 4 at hidden 4096 (64 x 4 = 256), 3 at 5120 (80 x 3 = 240; 4 splits = 320 workgroups run as two waves
    // on 256 CUs), 2 at 8192.  Dense models only: the MoE accumulate kernels are instantiated for 4 slabs.
    int auto_ks = 4;
    if (c.n_experts == 0 && c.hidden >= 64) {
        const int cus = c.balanced_wg[1] > 0 ? c.balanced_wg[1] : 256;
        const int k = cus / (c.hidden / 64);
        auto_ks = k >= 4 ? 4 : k >= 1 ? k : 1;                   // never more than the 4 the small shapes were tuned and tested with
    }
    m->o_rb = pick(c.gemm_cfg[2], 2 | (3 << 8));      // o_proj: 8 waves x 8 tile-sets (the whole K slice in flight at once): 10.8 -> 9.4 us
    m->o_ks = pick(c.gemm_cfg[3], auto_ks);
    m->down_rb = pick(c.gemm_cfg[4], 2);
    m->down_ks = pick(c.gemm_cfg[5], auto_ks);
    // a K split of fewer than 4 k-tiles is outside what the slab kernels were exercised with (observed: a device fault on the tiny
    // test shape, K = 256, with 8 splits = 2 tiles per split; 4 splits = 4 tiles run in every test): step an oversized request down
    // through the supported counts
    auto fit_ks = [](int ks, int k16) {
        static const int ok[] = {8, 6, 4, 3, 2, 1};
        for (int v : ok) if (v <= ks && (v == 1 || k16 / v >= 4)) return v;
        return 1;
    };
    m->o_ks = fit_ks(m->o_ks, m->o_k / 16);
    m->down_ks = fit_ks(m->down_ks, c.ffn / 16);
    m->lm_rb = pick(c.gemm_cfg[6], 2);
    m->gu_variant = c.gemm_cfg[7];
    // qkv_ks == -1 in the config selects the unfused path (plain [Wq;Wk;Wv] packing + k_qkv_post)
    m->qkv_fused = c.gemm_cfg[1] >= 0;
    if (!m->qkv_fused) m->qkv_ks = 1;
    // In-kernel norm fusion (opt-in: measured SLOWER than separate kernels on MI355X) needs the
    // balanced (one workgroup per CU) GEMM, 4 split-K slabs in front of it and a dense MLP.
    m->fuse = 0;
    if (c.fuse > 0 && c.n_experts == 0) {
        const int want = c.fuse;
        if ((want & 1) && c.balanced_wg[1] >= LA_TREE_MAX && m->o_ks == 4) m->fuse |= 1;
        if ((want & 2) && c.balanced_wg[0] >= LA_TREE_MAX && m->down_ks == 4) m->fuse |= 2;
        // bit 4 (value 16, round 4): the fused producers publish write-through (sc1 stores + drained flag) instead of plain
        // stores + release fence — the cheap publish form of MI355X_MICROARCH.md (publish-large: 3.0 vs 8.2 us)
        if ((want & 16) && (m->fuse & 3)) m->fuse |= 16;
    }
    // bit 2 (value 4), bit 3 (value 8 = 8 tile-sets in flight in the down role): gate/up and down_proj as ONE role-fused launch
    // (k_gateup_down): needs the balanced gate/up image, the 64-row classic down image and a dense MLP
    if ((c.fuse & 4) && c.n_experts == 0 && c.balanced_wg[1] > 0 && (c.hidden % 64) == 0 && (c.ffn % 16) == 0) m->fuse |= (c.fuse & 12);
    // single-launch attention while the K/V one XCD's heads read fits its 4 MiB L2 with room to spare: kv heads per XCD x 512 B per key
    // (measured at 4 kv heads per XCD: ahead of the key-split pair up to ~1500 keys, behind from ~2000, profiles/r04_attention_one_launch.txt)
    {
        const int kvx = c.n_kv_heads >= 8 ? (c.n_kv_heads + 7) / 8 : 1;
        m->attn_thr = (int)(2800000 / ((long)kvx * 512));
    }
    // gathered multi-block MoE: half the K splits of the dense down projection (Mixtral-8x7B bs=4: 21.5-21.65 -> 20.9-21.05 ms per step
    // at 2 splits, 21.1 at 1; profiles/r04_moe_down_ks.txt)
    m->ex_down_ks = m->down_ks >= 2 ? m->down_ks / 2 : 1;
    if (m->qkv_n % 64) m->qkv_rb = (m->qkv_rb & ~0xff) | 1;
    if (c.hidden % 64) { m->o_rb = (m->o_rb & ~0xff) | 1; m->down_rb = (m->down_rb & ~0xff) | 1; }
    if (c.vocab % 64) m->lm_rb = (m->lm_rb & ~0xff) | 1;
}

static size_t carve(la_llama* m, char* base) {
    const la_llama_config& c = m->cfg;
    Carver cv{base, 0};
    m->n_slots = c.n_slots > 1 ? c.n_slots : 1;
    m->total_keys = m->n_slots * c.max_keys;
    const size_t KB = (size_t)m->total_keys / 32;
    m->kv_layer_elems = (size_t)c.n_kv_heads * KB * 4096;
    m->fresh_layer_elems = (size_t)c.n_kv_heads * 2 * 4096;
    m->kmain = cv.take<uint16_t>(m->kv_layer_elems * c.n_layers);
    m->vmain = cv.take<uint16_t>(m->kv_layer_elems * c.n_layers);
    m->kfresh = cv.take<uint16_t>(m->fresh_layer_elems * c.n_layers);
    m->vfresh = cv.take<uint16_t>(m->fresh_layer_elems * c.n_layers);
    m->qf = cv.take<uint16_t>((size_t)c.n_heads * 2 * 4096);
    m->h = cv.take<uint16_t>((size_t)64 * c.hidden);
    m->xp = cv.take<uint16_t>((size_t)64 * c.hidden);
    m->attn_xp = cv.take<uint16_t>((size_t)64 * m->o_k);
    m->act_xp = cv.take<uint16_t>((size_t)64 * c.ffn);
    m->logits = cv.take<uint16_t>((size_t)64 * c.vocab);
    size_t slab_n = (size_t)m->qkv_n * m->qkv_ks;
    if ((size_t)c.hidden * m->o_ks > slab_n) slab_n = (size_t)c.hidden * m->o_ks;
    if ((size_t)c.hidden * m->down_ks > slab_n) slab_n = (size_t)c.hidden * m->down_ks;
    m->slabs = cv.take<float>(slab_n * 64);
    m->opart = cv.take<float>((size_t)c.n_heads * m->nsplit * 64 * 128);
    m->mpart = cv.take<float>((size_t)c.n_heads * m->nsplit * 64);
    m->lpart = cv.take<float>((size_t)c.n_heads * m->nsplit * 64);
    size_t cand = (size_t)(c.vocab / 32) * 4;
    if ((size_t)c.balanced_wg[2] * 8 > cand) cand = (size_t)c.balanced_wg[2] * 8;
    m->cand_val = cv.take<float>(cand * 64);
    m->cand_idx = cv.take<int>(cand * 64);
    m->state = cv.take<int>(LA_ST_WORDS);
    m->in = cv.take<int>(LA_IN_WORDS);
    m->pos = cv.take<int>(64);
    m->ids = cv.take<int>(64);
    m->rowmask = cv.take<uint64_t>(64);
    m->bstate = cv.take<int>(LA_BST_WORDS);
    m->bin = cv.take<int>(LA_BIN_WORDS);
    m->moe_acc = cv.take<uint16_t>(c.n_experts > 0 ? (size_t)64 * c.hidden : 8);
    m->act_ex = cv.take<uint16_t>(c.n_experts > 0 ? (size_t)c.n_experts * 64 * c.ffn : 8);
    m->slabs_ex = cv.take<float>(c.n_experts > 0 ? (size_t)c.n_experts * m->down_ks * 64 * c.hidden : 8);
    m->fuse_cnt = cv.take<int>((size_t)3 * c.n_layers + 8);
    m->norm_gran = cv.take<uint64_t>((size_t)2 * c.n_layers * 256);
    m->route_w = cv.take<float>((size_t)(c.n_experts > 0 ? c.n_layers : 1) * 64 * LA_MOE_MAX_E);   // kept per layer (parity tests)
    m->mb_max = c.max_blocks > 1 ? c.max_blocks : 0;
    if (m->mb_max) {
        const size_t MB = (size_t)LA_MB_MAX, R = MB * 64;            // sized for whole passes of 4 blocks
        m->mb_nsplit = 8;                                           // splits x blocks <= 8 (see mb_split): partial buffers hold 8 units
        m->mb_h = cv.take<uint16_t>(R * c.hidden);
        m->mb_xp = cv.take<uint16_t>(R * c.hidden);
        m->mb_attn_xp = cv.take<uint16_t>(R * m->o_k);
        m->mb_act = cv.take<uint16_t>(R * c.ffn);
        m->mb_logits = cv.take<uint16_t>(R * c.vocab);
        const int ksm = m->o_ks > m->down_ks ? m->o_ks : m->down_ks;
        m->mb_slabs = cv.take<float>((size_t)ksm * R * c.hidden);
        m->mb_qf = cv.take<uint16_t>(MB * c.n_heads * 8192);
        m->mb_fresh_layer = MB * c.n_kv_heads * 8192;
        m->mb_kfresh = cv.take<uint16_t>(m->mb_fresh_layer * c.n_layers);
        m->mb_vfresh = cv.take<uint16_t>(m->mb_fresh_layer * c.n_layers);
        m->mb_opart = cv.take<float>((size_t)8 * c.n_heads * 64 * 128);
        m->mb_mpart = cv.take<float>((size_t)8 * c.n_heads * 64);
        m->mb_lpart = cv.take<float>((size_t)8 * c.n_heads * 64);
        const size_t cslots = (size_t)lk_mb_cand_slots(lk_mb_logits_wgs(c.vocab, c.balanced_wg[2]));
        m->mb_cand_val = cv.take<float>(MB * cslots * 64);
        m->mb_cand_idx = cv.take<int>(MB * cslots * 64);
        m->mb_meta = cv.take<int>(MB * LA_MB_META);
        m->mb_pos = cv.take<int>(R);
        m->mb_ids = cv.take<int>(R);
        m->mb_rowmask = cv.take<uint64_t>(R);
        m->mb_in = cv.take<int>(LA_MIN_WORDS);
        m->mb_out = cv.take<int>(LA_MOUT_WORDS);
        const size_t E = (size_t)c.n_experts;
        m->mb_moe_acc = cv.take<uint16_t>(E ? R * c.hidden : 8);
        m->mb_act_ex = cv.take<uint16_t>(E ? E * R * c.ffn : 8);
        m->mb_route_w = cv.take<float>(E ? (size_t)c.n_layers * R * LA_MOE_MAX_E : 8);    // [layer][M][8]: kept per layer (parity tests force the oracle's routing with them)
        m->mb_slabs_ex = cv.take<float>(E ? E * m->down_ks * R * c.hidden : 8);
        m->mb_moe_perm = cv.take<int>(E ? E * LA_MB_MAX * 64 : 8);
        m->mb_moe_pos = cv.take<int>(E ? R * LA_MOE_MAX_E : 8);
        m->mb_moe_cnt = cv.take<int>(2 * LA_MOE_MAX_E);
        m->mb_xg = cv.take<uint16_t>(E ? E * R * c.hidden : 8);
    }
    return align_up(cv.off, 256);
}

static int validate(const la_llama_config* c) {
    if (!c) return LA_E_ARG;
    if (c->head_dim < 8 || c->head_dim > 128 || (c->head_dim & 1)) { la_set_error("head_dim must be even, 8 .. 128 (heads narrower than 128 run in padded 128-feature lanes)"); return LA_E_ARG; }
    if (lk_qk_scale_check(c->head_dim) != 0) { la_set_error("softmax scale: x * fp32(1 / sqrt(head_dim)) is not the rounded quotient for every bf16 x at this head_dim"); return LA_E_ARG; }
    if (c->n_layers <= 0 || c->hidden % 32 || c->hidden > 8192 || c->ffn % 32 || c->vocab % 32 ||
        c->n_heads % c->n_kv_heads || c->max_keys % 32 || c->max_keys < 96 || c->n_slots < 0 || c->n_slots > LA_MAX_SEQ ||
        c->max_blocks < 0 || c->max_blocks > LA_MB_MAX || c->n_experts < 0 || c->n_experts > LA_MOE_MAX_E || (c->n_experts > 0 && (c->top_k < 1 || c->top_k > c->n_experts))) {
        la_set_error("unsupported llama config (need hidden%32==0<=8192, ffn%32==0, vocab%32==0, max_keys%32==0)");
        return LA_E_ARG;
    }
    if (c->kv_ring) {
        const int blocks = c->max_blocks > 1 ? c->max_blocks : 1;
        if (c->sliding_window <= 0 || c->max_keys < c->sliding_window + 64 * blocks + 32) {
            la_set_error("kv_ring needs sliding_window > 0 and max_keys >= sliding_window + 64 * max(max_blocks, 1) + 32");
            return LA_E_ARG;
        }
    }
    return LA_OK;
}

extern "C" int64_t la_llama_workspace_bytes(const la_llama_config* cfg) {
    if (validate(cfg) != LA_OK) return LA_E_ARG;
    la_llama tmp{};
    tmp.cfg = *cfg;
    resolve_cfg(&tmp);
    return (int64_t)carve(&tmp, nullptr);
}

extern "C" la_llama* la_llama_create(const la_llama_config* cfg, const la_llama_weights* w, void* ws, int64_t ws_bytes) {
    if (validate(cfg) != LA_OK || !w || !ws || !w->layers) { if (!w || !ws) la_set_error("null weights/workspace"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        la_set_error("no HIP device: liblookahead_hip has no CPU fallback");
        return nullptr;
    }
    if (lk_gemm64r_init() != 0) {      // kernels with > 64 KiB of dynamic LDS (never inside a stream capture)
        la_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        return nullptr;
    }
    la_llama* m = new la_llama();
    m->cfg = *cfg;
    m->w = *w;
    m->layers.assign(w->layers, w->layers + cfg->n_layers);
    m->w.layers = m->layers.data();
    if (cfg->n_experts > 0) {
        for (int l = 0; l < cfg->n_layers; ++l) {
            const la_llama_layer_weights& L = w->layers[l];
            if (!L.router || !L.ex_gateup || !L.ex_down) { la_set_error("MoE layer without router/expert weights"); delete m; return nullptr; }
            for (int e = 0; e < cfg->n_experts; ++e) { m->ex_gateup.push_back(L.ex_gateup[e]); m->ex_down.push_back(L.ex_down[e]); }
        }
        // equally spaced expert images (the Python engine packs them into one buffer per layer) -> one launch per stage
        m->ex_merged = cfg->top_k <= 4;          // k_moe_accum_all gathers at most 4 routed experts per row
        for (int l = 0; l < cfg->n_layers && m->ex_merged; ++l) {
            const char* g0 = (const char*)m->ex_gateup[(size_t)l * cfg->n_experts];
            const char* d0 = (const char*)m->ex_down[(size_t)l * cfg->n_experts];
            long gs = cfg->n_experts > 1 ? (long)((const char*)m->ex_gateup[(size_t)l * cfg->n_experts + 1] - g0) : 0;
            long ds = cfg->n_experts > 1 ? (long)((const char*)m->ex_down[(size_t)l * cfg->n_experts + 1] - d0) : 0;
            for (int e = 0; e < cfg->n_experts; ++e) {
                if ((const char*)m->ex_gateup[(size_t)l * cfg->n_experts + e] != g0 + (long)e * gs) m->ex_merged = false;
                if ((const char*)m->ex_down[(size_t)l * cfg->n_experts + e] != d0 + (long)e * ds) m->ex_merged = false;
            }
            if (gs < 0 || ds < 0 || (gs & 15) || (ds & 15)) m->ex_merged = false;
            m->ex_gu_stride.push_back(gs / 2);
            m->ex_dn_stride.push_back(ds / 2);
        }
    }
    resolve_cfg(m);
    size_t need = carve(m, (char*)ws);
    if ((int64_t)need > ws_bytes) { la_set_error("workspace too small"); delete m; return nullptr; }
    m->ws = (char*)ws;
    m->graph_ready = m->bgraph_ready = false;
    m->graph_exec = m->bgraph_exec = nullptr;
    m->graph_stream = nullptr;
    m->zc_in = nullptr; m->zc_out = nullptr; m->seq_expected = 0;
    if (cfg->n_experts == 0) m->ex_merged = false;
    for (int i = 0; i < 4; ++i) { m->bgraphs[i] = nullptr; m->bready[i] = false; }
    for (int i = 0; i < 2 * (LA_MB_MAX + 1); ++i) { m->mgraphs[i] = nullptr; m->mready[i] = false; }
    if (m->mb_max && lk_mb_init() != 0) { la_set_error("hipFuncSetAttribute failed for the multi-block kernels"); delete m; return nullptr; }
    return m;
}

extern "C" void la_llama_destroy(la_llama* m) {
    if (!m) return;
    if (m->graph_exec) (void)hipGraphExecDestroy(m->graph_exec);
    if (m->graph_long) (void)hipGraphExecDestroy(m->graph_long);
    if (m->bgraph_exec) (void)hipGraphExecDestroy(m->bgraph_exec);
    for (int i = 0; i < 4; ++i) if (m->bready[i]) (void)hipGraphExecDestroy(m->bgraphs[i]);
    for (int i = 0; i < 2 * (LA_MB_MAX + 1); ++i) if (m->mready[i]) (void)hipGraphExecDestroy(m->mgraphs[i]);
    if (m->pf_fork) (void)hipEventDestroy(m->pf_fork);
    if (m->pf_join) (void)hipEventDestroy(m->pf_join);
    if (m->pf_stream) (void)hipStreamDestroy(m->pf_stream);
    delete m;
}

extern "C" int la_llama_reset(la_llama* m, void* stream) {
    if (!m) return LA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(m->state, 0, LA_ST_WORDS * sizeof(int), st));
    int mk = m->cfg.max_keys;
    HIPCHK(hipMemcpyAsync(m->state + LA_ST_MAXKEYS, &mk, sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(m->bstate, 0, LA_BST_WORDS * sizeof(int), st));
    HIPCHK(hipStreamSynchronize(st));
    m->seq_expected = 0;
    if (m->zc_out) m->zc_out[LA_ST_SEQ] = 0;
    return LA_OK;
}

extern "C" int la_llama_reset_slot(la_llama* m, void* stream, int slot) {
    if (!m || slot >= m->n_slots) return LA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (slot < 0) HIPCHK(hipMemsetAsync(m->bstate, 0, LA_BST_WORDS * sizeof(int), st));
    else HIPCHK(hipMemsetAsync(m->bstate + LA_BST_NKEYS + slot, 0, sizeof(int), st));
    HIPCHK(hipStreamSynchronize(st));
    return LA_OK;
}

// kernel classes for la_llama_profile
enum { KC_QKV = 0, KC_O, KC_GATEUP, KC_DOWN, KC_LMHEAD, KC_ATTN, KC_OTHER, KC_N };

struct Prof {
    bool on = false;
    std::vector<hipEvent_t> ev;
    std::vector<int> cls;
    size_t used = 0;
    hipStream_t st;
    void mark(int c) {
        if (!on) return;
        if (used >= ev.size()) { hipEvent_t e; (void)hipEventCreate(&e); ev.push_back(e); }
        (void)hipEventRecord(ev[used++], st);
        cls.push_back(c);
    }
};

// enqueue every kernel of one block on `st` (used eagerly, under graph capture, and by the profiler)
static int enqueue_step(la_llama* m, hipStream_t st, Prof* pf, bool batch = false, const int32_t* zc_in = nullptr,
                        int32_t* zc_out = nullptr, int bsplit = 0, bool long_ctx = false, bool fork = false) {
    const la_llama_config& c = m->cfg;
    // Forked weight prefetch: `fork_pf(d)` makes the side stream wait for everything queued on `st` so far and launches the
    // prefetch workgroups of descriptor d there — under stream capture a parallel branch of the graph that starts when the step
    // reaches this point and is joined at the end of the step only (nothing waits for a prefetch: it writes nothing).
    bool forked = false;
    auto fork_pf = [&](const PfDesc& d, bool wait_here) -> int {
        if (!fork || !m->pf_stream || !d.base) return LA_OK;
        if (wait_here) {
            HIPCHK(hipEventRecord(m->pf_fork, st));
            HIPCHK(hipStreamWaitEvent(m->pf_stream, m->pf_fork, 0));
        }
#if LA_LAB
        KCHK(lk_pf_only(m->pf_stream, &d));
#endif
        forked = true;
        return LA_OK;
    };
    auto join_pf = [&]() -> int {
        if (!forked) return LA_OK;
        HIPCHK(hipEventRecord(m->pf_join, m->pf_stream));
        HIPCHK(hipStreamWaitEvent(st, m->pf_join, 0));
        forked = false;
        return LA_OK;
    };
    const int* fk = g_la_fork_pf;
    const int ring = c.kv_ring ? c.max_keys : 0;          // sliding-window ring: position p of a sequence lives in row p mod max_keys
    auto P = [&](int cls) { if (pf) pf->mark(cls); };
    P(KC_OTHER);
    if (batch) KCHK(lk_build_tree_inputs_b(st, m->bin, m->bstate, m->pos, m->rowmask, m->ids));
    else if (g_la_split_head_tail) KCHK(lk_build_tree_inputs(st, zc_in ? (const int*)zc_in : m->in, m->state, m->pos, m->rowmask, m->ids));
    const int cf = c.norm_cast_first;
    if (m->fuse) HIPCHK(hipMemsetAsync(m->fuse_cnt, 0, sizeof(int) * 3 * c.n_layers, st));
    // Idle-window weight prefetch (la_kernels.h, PfDesc): the row kernels and the attention combine carry extra workgroups that
    // pull the first KiB every workgroup of the NEXT GEMM will stream into the L2 of its XCD.  g_la_pf_kib = 0 switches it off.
    const int pf_kib = m->fuse ? 0 : g_la_pf_kib, pf_dly = g_la_pf_delay;
    // four workgroups per norm row (k_row_norm4): single-sequence step with the fused head kernel (it zeroes the granule words); the
    // idle-window prefetch workgroups ride on k_row_norm, so the knob (key 7) keeps the one-workgroup form
    [[maybe_unused]] const bool norm4 = g_la_norm4 && !batch && !g_la_split_head_tail && pf_kib == 0 && (c.hidden & 15) == 0;
    auto pf_qkv = [&](int l, PfDesc* d) {
        *d = PfDesc{};
        if (pf_kib > 0 && l < c.n_layers && c.balanced_wg[0] > 0)
            lk_pf_planned(d, m->layers[l].wqkv, 2, (c.n_heads + 2 * c.n_kv_heads) * 128, c.hidden, c.balanced_wg[0], pf_kib, pf_dly, nullptr);
    };
    PfDesc pd{};
    pf_qkv(0, &pd);
    // single-sequence step: input expansion + embedding row kernel in ONE launch (k_step_head); la_debug_set key 14 = 1 restores
    // the separate kernels of rounds 1-2 (A/B measurements)
    if (!batch && !g_la_split_head_tail)
        KCHK(lk_step_head(st, zc_in ? (const int*)zc_in : m->in, m->state, m->pos, m->rowmask, m->ids, m->w.embed, m->layers[0].norm1,
                          c.hidden, c.rms_eps, m->h, m->xp, cf, &pd, m->norm_gran, 2 * c.n_layers * 256));
    else
        KCHK(lk_embed_norm(st, m->w.embed, m->ids, m->layers[0].norm1, c.hidden, c.rms_eps, m->h, m->xp, cf, &pd));
    // depth probe (la_debug_set key 13): the first nl layers, then the FINAL norm + lm_head — h / logits after nl layers for the
    // per-depth parity test (tests/test_gpu_e2e.py::test_full_size_llama7b_32_layers_vs_oracle); 0 = the whole model
    const int nl = (!batch && g_la_stop_layers > 0 && g_la_stop_layers < c.n_layers) ? g_la_stop_layers : c.n_layers;
    for (int l = 0; l < nl; ++l) {
        const la_llama_layer_weights& L = m->layers[l];
        uint16_t* kf = m->kfresh + (size_t)l * m->fresh_layer_elems;
        uint16_t* vf = m->vfresh + (size_t)l * m->fresh_layer_elems;
        P(KC_QKV);
        if (c.balanced_wg[0] > 0) {
            // layers > 0: the input norm (residual + down-projection slabs of the previous layer) runs inside this launch
            FusedNorm fq{m->slabs, m->down_ks, m->h, L.norm1, c.hidden, c.rms_eps, cf, m->fuse_cnt + 2 * l, (m->fuse & 16) ? 1 : 0};
            KCHK(lk_gemm64r_qkv(st, L.wqkv, m->xp, c.n_heads, c.n_kv_heads, c.hidden, c.balanced_wg[0], m->pos,
                                m->w.rope_cos, m->w.rope_sin, m->qf, kf, vf, ((m->fuse & 2) && l > 0) ? &fq : nullptr));
        } else if (m->qkv_fused) {
            KCHK(lk_gemm64_qkv(st, L.wqkv, m->xp, c.n_heads, c.n_kv_heads, c.hidden, m->pos, m->w.rope_cos, m->w.rope_sin,
                               m->qf, kf, vf, m->qkv_rb >> 8));
        } else {
            KCHK(lk_gemm64_slab(st, L.wqkv, m->xp, m->qkv_n, c.hidden, m->qkv_rb, m->qkv_ks, m->slabs));
            P(KC_OTHER);
            KCHK(lk_qkv_post(st, m->slabs, m->qkv_ks, c.n_heads, c.n_kv_heads, m->pos, m->w.rope_cos, m->w.rope_sin,
                             m->qf, kf, vf));
        }
        P(KC_ATTN);
        if (fork && c.n_experts == 0) {
            PfDesc fd{};
            bool first = true;
            if (fk[0] > 0) { lk_pf_classic(&fd, L.wo, c.hidden, m->o_k, m->o_rb, m->o_ks, fk[0], 0, nullptr); if (fd.base) { int rc = fork_pf(fd, first); if (rc != LA_OK) return rc; first = false; } }
            if (fk[1] > 0 && c.balanced_wg[1] > 0) { lk_pf_planned(&fd, L.wgateup, 1, c.ffn, c.hidden, c.balanced_wg[1], fk[1], 0, nullptr); if (fd.base) { int rc = fork_pf(fd, first); if (rc != LA_OK) return rc; first = false; } }
        }
        pd = PfDesc{};
        if (pf_kib > 0) lk_pf_classic(&pd, L.wo, c.hidden, m->o_k, m->o_rb, m->o_ks, pf_kib, pf_dly, nullptr);
        if (batch)
            KCHK(lk_tree_attn_b(st, m->qf, m->kmain + (size_t)l * m->kv_layer_elems, m->vmain + (size_t)l * m->kv_layer_elems,
                                kf, vf, m->rowmask, m->bstate, c.n_heads, c.n_kv_heads, c.max_keys, m->n_slots, bsplit > 0 ? bsplit : m->nsplit,
                                m->opart, m->mpart, m->lpart, m->attn_xp, c.sliding_window, ring, &pd, c.head_dim));
#if LA_LAB
        else if (g_la_attn_merge_ns > 0 && !long_ctx && c.n_experts == 0 && c.sliding_window <= 0 && (m->o_k / 16) / m->o_ks == 64 &&
                 g_la_attn_merge_ns <= m->nsplit && c.hidden % 64 == 0) {
            // lab knob 33 (review item 1b): key-split attention over NS splits, NO combine launch — o_proj merges the partials on load
            KCHK(lk_tree_attn(st, m->qf, m->kmain + (size_t)l * m->kv_layer_elems, m->vmain + (size_t)l * m->kv_layer_elems,
                              kf, vf, m->rowmask, m->state, c.n_heads, c.n_kv_heads, m->total_keys, g_la_attn_merge_ns,
                              m->opart, m->mpart, m->lpart, nullptr, c.sliding_window, ring, nullptr, 0, nullptr, c.head_dim));
            P(KC_O);
            KCHK(lk_oproj_merge(st, L.wo, c.hidden, m->o_k, m->o_ks, g_la_attn_merge_ns, m->opart, m->mpart, m->lpart, m->slabs));
            goto after_oproj;
        }
#endif
        else {
            // riders (la_lab_set key 31): the single-launch attention occupies nh * 4 CUs; the others pull o_proj's first KiB into L2
            PfDesc rd{};
            if (g_la_attn_ride_kib > 0 && !long_ctx) lk_pf_classic(&rd, L.wo, c.hidden, m->o_k, m->o_rb, m->o_ks, g_la_attn_ride_kib, g_la_attn_ride_delay, nullptr);
            KCHK(lk_tree_attn(st, m->qf, m->kmain + (size_t)l * m->kv_layer_elems, m->vmain + (size_t)l * m->kv_layer_elems,
                              kf, vf, m->rowmask, m->state, c.n_heads, c.n_kv_heads, m->total_keys, m->nsplit,
                              m->opart, m->mpart, m->lpart, m->attn_xp, c.sliding_window, ring, &pd, long_ctx ? 0 : -1, &rd, c.head_dim));
        }
        P(KC_O);
        if (g_la_oproj_probe && !batch)            // timing probe (lab knob 34): another o_proj geometry; numerics are NOT preserved
            KCHK(lk_gemm64_slab(st, L.wo, m->attn_xp, c.hidden, m->o_k, (g_la_oproj_probe & 16) ? (1 | (1 << 8)) : m->o_rb,
                                (g_la_oproj_probe & 15) ? (g_la_oproj_probe & 15) : m->o_ks, m->slabs));
        else
        KCHK(lk_gemm64_slab(st, L.wo, m->attn_xp, c.hidden, m->o_k, m->o_rb, m->o_ks, m->slabs));
#if LA_LAB
    after_oproj:
#endif
        P(KC_OTHER);
        if (fork && c.n_experts == 0) {
            PfDesc fd{};
            bool first = true;
            if (fk[2] > 0 && c.balanced_wg[1] > 0) { lk_pf_planned(&fd, L.wgateup, 1, c.ffn, c.hidden, c.balanced_wg[1], fk[2], 0, nullptr); if (fd.base) { int rc = fork_pf(fd, first); if (rc != LA_OK) return rc; first = false; } }
            if (fk[4] > 0) { lk_pf_classic(&fd, L.wdown, c.hidden, c.ffn, m->down_rb, m->down_ks, fk[4], 0, nullptr); if (fd.base) { int rc = fork_pf(fd, first); if (rc != LA_OK) return rc; first = false; } }
        }
        const void* nw = (l + 1 < nl) ? m->layers[l + 1].norm1 : m->w.final_norm;
        if (c.n_experts > 0) {
            // sparse MoE MLP: router fused into the norm, then per expert {gate/up+SwiGLU, down, weighted accumulate};
            // an expert no row routes to costs three empty launches and no weight traffic
            float* rw = m->route_w + (size_t)l * 64 * LA_MOE_MAX_E;
            KCHK(lk_resid_norm_router(st, m->h, m->slabs, m->o_ks, L.norm2, c.hidden, c.rms_eps, m->xp, L.router, c.n_experts,
                                      c.top_k, rw, batch ? m->bin + LA_BIN_T : m->state + LA_ST_T, cf));
            if (m->ex_merged) {
                // one launch per stage for ALL experts: workgroups of experts nobody routes to return at once, the others
                // keep the chip full across expert boundaries (no per-expert ramp / drain, no empty launches)
                const void* wgu0 = m->ex_gateup[(size_t)l * c.n_experts];
                const void* wdn0 = m->ex_down[(size_t)l * c.n_experts];
                const long act_stride = (long)64 * c.ffn, slab_stride = (long)m->down_ks * 64 * c.hidden;
                P(KC_GATEUP);
                if (c.balanced_wg[1] > 0)
                    KCHK(lk_gemm64r_swiglu_ex(st, wgu0, m->ex_gu_stride[l], m->xp, c.ffn, c.hidden, c.balanced_wg[1], m->act_ex,
                                              act_stride, rw, c.n_experts));
                else
                    KCHK(lk_gemm64_swiglu_ex(st, wgu0, m->ex_gu_stride[l], m->xp, c.ffn, c.hidden, m->act_ex, act_stride, rw,
                                             c.n_experts));
                P(KC_DOWN);
                KCHK(lk_gemm64_slab_ex(st, wdn0, m->ex_dn_stride[l], m->act_ex, act_stride, c.hidden, c.ffn, m->down_rb, m->down_ks,
                                       m->slabs_ex, slab_stride, rw, c.n_experts));
                P(KC_OTHER);
                KCHK(lk_moe_accum_all(st, m->slabs_ex, slab_stride, m->down_ks, rw, c.n_experts, c.hidden, m->moe_acc));
            } else
            for (int e = 0; e < c.n_experts; ++e) {
                const float* col = rw + e;
                const void* wgu = m->ex_gateup[(size_t)l * c.n_experts + e];
                const void* wdn = m->ex_down[(size_t)l * c.n_experts + e];
                P(KC_GATEUP);
                if (c.balanced_wg[1] > 0) KCHK(lk_gemm64r_swiglu(st, wgu, m->xp, c.ffn, c.hidden, c.balanced_wg[1], m->act_xp, col));
                else KCHK(lk_gemm64_swiglu(st, wgu, m->xp, c.ffn, c.hidden, m->act_xp, m->gu_variant, col));
                P(KC_DOWN);
                KCHK(lk_gemm64_slab(st, wdn, m->act_xp, c.hidden, c.ffn, m->down_rb, m->down_ks, m->slabs, col));
                P(KC_OTHER);
                KCHK(lk_moe_accum(st, m->slabs, m->down_ks, col, c.hidden, m->moe_acc, e == 0));
            }
            KCHK(lk_resid_norm_addend(st, m->h, m->moe_acc, nw, c.hidden, c.rms_eps, m->xp, cf));
            continue;
        }
        if (m->fuse & 1) {
            FusedNorm fg{m->slabs, m->o_ks, m->h, L.norm2, c.hidden, c.rms_eps, cf, m->fuse_cnt + 2 * l + 1, (m->fuse & 16) ? 1 : 0};
            P(KC_GATEUP);
            if ((m->fuse & 4) && (m->down_rb & 0xff) == 2) {
                // bits 0 + 2: norm -> gate/up + SwiGLU -> down_proj as ONE launch (k_gateup_down<.., NSF = 4>): the MLP half of the layer
                // with two in-launch hand-overs instead of two kernel boundaries
                KCHK(lk_gateup_down(st, L.wgateup, m->xp, c.ffn, c.hidden, c.balanced_wg[1], m->act_xp, L.wdown, c.hidden, m->down_ks,
                                    m->slabs, m->fuse_cnt + 2 * c.n_layers + l, (m->fuse & 8) ? 8 : 4, &fg));
                P(KC_OTHER);
                goto after_down;
            }
            KCHK(lk_gemm64r_swiglu(st, L.wgateup, m->xp, c.ffn, c.hidden, c.balanced_wg[1], m->act_xp, nullptr, &fg));
        } else {
            pd = PfDesc{};
            if (pf_kib > 0 && c.balanced_wg[1] > 0) lk_pf_planned(&pd, L.wgateup, 1, c.ffn, c.hidden, c.balanced_wg[1], pf_kib, pf_dly, nullptr);
            if ((g_la_oproj_probe & 32) && !batch) {}          // timing probe: the post-attention norm launch is skipped (x is stale)
#if LA_LAB
            else if (norm4) KCHK(lk_resid_norm4(st, m->h, m->slabs, m->o_ks, L.norm2, c.hidden, c.rms_eps, m->xp, cf, m->norm_gran + (size_t)(2 * l) * 256));
#endif
            else KCHK(lk_resid_norm(st, m->h, m->slabs, m->o_ks, L.norm2, c.hidden, c.rms_eps, m->xp, cf, &pd));
            P(KC_GATEUP);
            if ((m->fuse & 4) && (m->down_rb & 0xff) == 2) {
                // gate/up + down_proj in one launch: the down role's workgroups start as gate/up's exit, with their first weight
                // tile-sets in flight while they wait for act (k_gateup_down)
                KCHK(lk_gateup_down(st, L.wgateup, m->xp, c.ffn, c.hidden, c.balanced_wg[1], m->act_xp, L.wdown, c.hidden, m->down_ks,
                                    m->slabs, m->fuse_cnt + 2 * c.n_layers + l, (m->fuse & 8) ? 8 : 4));
                P(KC_OTHER);
                goto after_down;
            }
            if (c.balanced_wg[1] > 0) {
                // down_proj follows at once: its first k-tiles are pulled into L2 from the tail of the gate/up launch
                pd = PfDesc{};
                if (g_la_pf_tail_kib > 0 && !m->fuse) lk_pf_classic(&pd, L.wdown, c.hidden, c.ffn, m->down_rb, m->down_ks, g_la_pf_tail_kib, 0, nullptr);
                if (pd.n_consumers > c.balanced_wg[1]) pd = PfDesc{};         // one prefetching workgroup per consumer workgroup
                KCHK(lk_gemm64r_swiglu(st, L.wgateup, m->xp, c.ffn, c.hidden, c.balanced_wg[1], m->act_xp, nullptr, nullptr, &pd));
            } else KCHK(lk_gemm64_swiglu(st, L.wgateup, m->xp, c.ffn, c.hidden, m->act_xp, m->gu_variant));
        }
        P(KC_DOWN);
        KCHK(lk_gemm64_slab(st, L.wdown, m->act_xp, c.hidden, c.ffn, m->down_rb, m->down_ks, m->slabs));
        P(KC_OTHER);
        if (fork && fk[3] > 0) {
            PfDesc fd{};
            if (l + 1 < nl) { if (c.balanced_wg[0] > 0) lk_pf_planned(&fd, m->layers[l + 1].wqkv, 2, (c.n_heads + 2 * c.n_kv_heads) * 128, c.hidden, c.balanced_wg[0], fk[3], 0, nullptr); }
            else if (c.balanced_wg[2] > 0) lk_pf_planned(&fd, m->w.lm_head, 0, c.vocab, c.hidden, c.balanced_wg[2], fk[3], 0, nullptr);
            if (fd.base) { int rc = fork_pf(fd, true); if (rc != LA_OK) return rc; }
        }
    after_down:
        if (!((m->fuse & 2) && l + 1 < nl)) {        // otherwise fused into the next layer's QKV launch
            if (l + 1 < nl) pf_qkv(l + 1, &pd);
            else {
                pd = PfDesc{};
                if (pf_kib > 0 && c.balanced_wg[2] > 0) lk_pf_planned(&pd, m->w.lm_head, 0, c.vocab, c.hidden, c.balanced_wg[2], pf_kib, pf_dly, nullptr);
            }
#if LA_LAB
            if (norm4) KCHK(lk_resid_norm4(st, m->h, m->slabs, m->down_ks, nw, c.hidden, c.rms_eps, m->xp, cf, m->norm_gran + (size_t)(2 * l + 1) * 256));
            else
#endif
            KCHK(lk_resid_norm(st, m->h, m->slabs, m->down_ks, nw, c.hidden, c.rms_eps, m->xp, cf, &pd));
        }
    }
    P(KC_LMHEAD);
    int* am_rows = batch ? m->bstate + LA_BST_ARGMAX : m->state + LA_ST_ARGMAX;
    if (c.balanced_wg[2] > 0) {
        KCHK(lk_gemm64r_logits(st, m->w.lm_head, m->xp, c.vocab, c.hidden, c.balanced_wg[2], m->logits, m->cand_val, m->cand_idx));
        P(KC_OTHER);
        if (!batch && !g_la_split_head_tail) {
            // argmax finalize + accept walk + result hand-over in ONE launch, the KV commit after it: the host reads the accepted
            // tokens (and starts its trie update / next query) while the commit kernel still runs
            KCHK(lk_step_tail(st, m->cand_val, m->cand_idx, c.balanced_wg[2], m->ids, m->rowmask, m->state, (int*)zc_out));
            KCHK(lk_kv_commit(st, m->kfresh, m->vfresh, m->kmain, m->vmain, m->state, c.n_layers, c.n_kv_heads, m->total_keys, ring));
            P(KC_N);
            return join_pf();
        }
        KCHK(lk_argmax_finalize(st, m->cand_val, m->cand_idx, c.balanced_wg[2], am_rows));
    } else {
        KCHK(lk_gemm64_logits(st, m->w.lm_head, m->xp, c.vocab, c.hidden, m->lm_rb, m->logits, m->cand_val, m->cand_idx));
        P(KC_OTHER);
        KCHK(lk_argmax_finalize(st, m->cand_val, m->cand_idx, lk_logits_cand_slots(c.vocab, m->lm_rb), am_rows));
    }
    if (batch) {
        KCHK(lk_accept_scan_b(st, m->bin, m->ids, m->rowmask, m->bstate, m->n_slots, c.max_keys, ring));
        KCHK(lk_kv_commit_b(st, m->kfresh, m->vfresh, m->kmain, m->vmain, m->bstate, c.n_layers, c.n_kv_heads, m->total_keys));
    } else {
        KCHK(lk_accept_scan(st, m->ids, m->rowmask, m->state));
        KCHK(lk_kv_commit(st, m->kfresh, m->vfresh, m->kmain, m->vmain, m->state, c.n_layers, c.n_kv_heads, m->total_keys, ring));
        if (zc_out) KCHK(lk_publish(st, m->state, (int*)zc_out));
    }
    P(KC_N);
    return join_pf();
}

static int build_graph(la_llama* m, hipStream_t st, bool batch = false, const int32_t* zc_in = nullptr,
                       int32_t* zc_out = nullptr, int bsplit = 0, bool long_ctx = false) {
    hipGraph_t g = nullptr;
    bool fork = false;
    if (!batch) for (int i = 0; i < 5; ++i) fork = fork || g_la_fork_pf[i] > 0;
    if (fork && !m->pf_stream) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);          // lo = the numerically largest value = the LOWEST priority
        HIPCHK(hipStreamCreateWithPriority(&m->pf_stream, hipStreamNonBlocking, lo));
        HIPCHK(hipEventCreateWithFlags(&m->pf_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&m->pf_join, hipEventDisableTiming));
    }
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    // measurement knob (la_debug_set key 11): the single-sequence graph holds the step n times (the same input block each time, only
    // the last repetition publishes) — what a launch costs beyond its kernels shows as time per repetition vs n
    const int reps = (!batch && g_la_graph_reps > 1) ? g_la_graph_reps : 1;
    int rc = LA_OK;
    for (int r = 0; r < reps && rc == LA_OK; ++r) rc = enqueue_step(m, st, nullptr, batch, zc_in, r + 1 == reps ? zc_out : nullptr, bsplit, long_ctx, fork);
    hipError_t e = hipStreamEndCapture(st, &g);
    if (rc != LA_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    HIPCHK(e);
    HIPCHK(hipGraphInstantiate(batch ? &m->bgraph_exec : (long_ctx ? &m->graph_long : &m->graph_exec), g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    (batch ? m->bgraph_ready : (long_ctx ? m->graph_long_ready : m->graph_ready)) = true;
    m->graph_stream = st;
    return LA_OK;
}

static int bstep(la_llama* m, void* stream, const int32_t* host_in, int32_t* host_out, bool eager) {
    if (!m || !host_in) return LA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(m->bin, host_in, LA_BIN_WORDS * sizeof(int), hipMemcpyHostToDevice, st));
    // The attention grid is (heads, key splits, slots): with many active slots fewer key splits fill the chip just as well
    // and leave fewer partials to merge.  splits = configured count / active slots, as a power of two; one captured graph
    // per value.
    unsigned seen = 0;
    const int T = host_in[LA_BIN_T];
    for (int r = 0; r < T && r < LA_TREE_MAX; ++r) {
        const int sl = host_in[LA_BIN_SEQ + r];
        if (sl >= 0 && sl < LA_MAX_SEQ) seen |= 1u << sl;
    }
    const int active = __builtin_popcount(seen) > 0 ? __builtin_popcount(seen) : 1;
    int v = 0, bsplit = m->nsplit;
    const int target = m->nsplit / active > 1 ? m->nsplit / active : 1;
    while (v < 3 && bsplit > target && bsplit % 2 == 0) { bsplit /= 2; ++v; }
    if (eager) {
        int rc = enqueue_step(m, st, nullptr, true, nullptr, nullptr, bsplit);
        if (rc != LA_OK) return rc;
    } else {
        if (m->bready[v] && m->bepoch[v] != g_la_graph_epoch) {      // a capture-time knob changed (la_debug_set): capture again
            (void)hipGraphExecDestroy(m->bgraphs[v]);
            m->bgraphs[v] = nullptr; m->bready[v] = false;
        }
        if (!m->bready[v]) {
            int rc = build_graph(m, st, true, nullptr, nullptr, bsplit);
            if (rc != LA_OK) return rc;
            m->bgraphs[v] = m->bgraph_exec; m->bgraph_exec = nullptr; m->bgraph_ready = false;
            m->bready[v] = true; m->bepoch[v] = g_la_graph_epoch;
        }
        HIPCHK(hipGraphLaunch(m->bgraphs[v], st));
    }
    if (host_out)
        HIPCHK(hipMemcpyAsync(host_out, m->bstate, LA_BST_DST * sizeof(int), hipMemcpyDeviceToHost, st));
    return LA_OK;
}

extern "C" int la_llama_bstep(la_llama* m, void* stream, const int32_t* host_in, int32_t* host_out) {
    return bstep(m, stream, host_in, host_out, false);
}
extern "C" int la_llama_bstep_eager(la_llama* m, void* stream, const int32_t* host_in, int32_t* host_out) {
    return bstep(m, stream, host_in, host_out, true);
}

// ---- multi-block step: nblk x 64 rows through the LDS-staged GEMM family (la_mblock.hip) -------------------------------
// key splits of the multi-block attention: heads x splits x blocks workgroups should fill the 256 CUs about once
static int mb_split(int nblk, int n_heads = 32, int cus = 256) {
    int s = nblk > 4 ? 1 : nblk > 2 ? 2 : nblk == 2 ? 4 : 8;                                   // splits x blocks <= 8 (partial buffers)
    while (s > 1 && n_heads * s * nblk > cus) s >>= 1;                                         // ... and one wave of workgroups (40 heads: 13B)
    return s;
}

static int enqueue_mstep(la_llama* m, hipStream_t st, int nblk, bool wide = false) {
    const la_llama_config& c = m->cfg;
    if (!m->qkv_fused) { la_set_error("mstep needs the fused QKV image (gemm_cfg[1] >= 0)"); return LA_E_ARG; }
    const int M = nblk * 64;
    const int npass_rows = (nblk <= 2 ? nblk : ((nblk + 3) / 4) * 4) * 64;     // rows whole passes write (slab stride)
    // la_debug_set key 12 (measurement, off by default): at >= 5 blocks the slab GEMMs of a dense model run 2 K splits over 4 token
    // groups instead of 4 splits over 2 — the same 256 workgroups, half the fp32 partials written and read back by the row kernels
    const bool ks2 = g_la_mb_ks2 && nblk >= 5 && c.n_experts == 0 && m->o_ks > 2 && m->down_ks > 2 && (c.hidden / 128) * 2 * 4 <= 256;
    const int o_ks = ks2 ? 2 : m->o_ks, down_ks = ks2 ? 2 : m->down_ks;
    const int cf = c.norm_cast_first;
    KCHK(lk_mb_build_inputs(st, m->mb_in, m->bstate, nblk, m->mb_meta, m->mb_pos, m->mb_rowmask, m->mb_ids));
    KCHK(lk_mb_embed_norm(st, m->w.embed, m->mb_ids, m->layers[0].norm1, c.hidden, c.rms_eps, m->mb_h, m->mb_xp, M, cf));
    for (int l = 0; l < c.n_layers; ++l) {
        const la_llama_layer_weights& L = m->layers[l];
        uint16_t* kf = m->mb_kfresh + (size_t)l * m->mb_fresh_layer;
        uint16_t* vf = m->mb_vfresh + (size_t)l * m->mb_fresh_layer;
        MbGemm q{}; q.xp = m->mb_xp; q.N = m->qkv_n; q.K = c.hidden; q.nblk = nblk; q.ksplit = 1;
        // >= 3 blocks (the wide launches): the image planned for fewer, fuller workgroups when the model carries one (cfg.qkv_mb_wg)
        const bool mbq = c.qkv_mb_wg > 0 && L.wqkv_mb && nblk >= 3;
        q.wp = mbq ? L.wqkv_mb : L.wqkv; q.n_wg = mbq ? c.qkv_mb_wg : c.balanced_wg[0];
        q.pos = m->mb_pos; q.rcos = m->w.rope_cos; q.rsin = m->w.rope_sin; q.qf = m->mb_qf; q.kfresh = kf; q.vfresh = vf;
        q.nh = c.n_heads; q.nkv = c.n_kv_heads;
        KCHK(lk_mb_gemm(st, 2, q));
        KCHK(lk_mb_tree_attn(st, m->mb_qf, m->kmain + (size_t)l * m->kv_layer_elems, m->vmain + (size_t)l * m->kv_layer_elems, kf, vf,
                             m->mb_rowmask, m->mb_meta, nblk, c.n_heads, c.n_kv_heads, c.max_keys, m->n_slots, mb_split(nblk, c.n_heads, c.balanced_wg[1] > 0 ? c.balanced_wg[1] : 256),
                             m->mb_opart, m->mb_mpart, m->mb_lpart, m->mb_attn_xp, c.sliding_window, c.kv_ring ? 1 : 0,
                             wide ? (const uint64_t*)(m->mb_in + LA_MIN_XMASK) : nullptr, c.head_dim));
        MbGemm o{}; o.wp = L.wo; o.xp = m->mb_attn_xp; o.N = c.hidden; o.K = m->o_k; o.nblk = nblk; o.ksplit = o_ks;
        o.slabs = m->mb_slabs; o.slab_rows = npass_rows;
        KCHK(lk_mb_gemm(st, 0, o));
        const void* nw = (l + 1 < c.n_layers) ? m->layers[l + 1].norm1 : m->w.final_norm;
        if (c.n_experts > 0) {
            // sparse MoE MLP over M rows: router fused into the norm; per expert {gate/up + SwiGLU, down} over ALL rows (an expert no
            // row routes to returns at once and its weights are never read), then the weighted accumulation in expert order
            float* const route_w = m->mb_route_w + (size_t)l * LA_MB_MAX * 64 * LA_MOE_MAX_E;
            KCHK(lk_mb_resid_norm_router(st, m->mb_h, m->mb_slabs, m->o_ks, npass_rows, L.norm2, c.hidden, c.rms_eps, m->mb_xp, M, cf,
                                         L.router, c.n_experts, c.top_k, route_w, m->mb_meta));
            // K splits of the experts' down projection in the gathered multi-block form: grid.z already carries experts x passes (>= 8
            // workgroup layers), so the launch fills the chip without split-K — and every split less is 1/4 of the fp32 slab round trip
            // (4 splits: 67 MB written + read per layer at 256 rows).  la_lab_set(22, n) overrides (0 = the library's choice).
            const int ex_ks = (nblk >= 2 && m->ex_merged && !(g_la_ex_split & 1)) ? (g_la_ex_down_ks > 0 ? g_la_ex_down_ks : m->ex_down_ks) : m->down_ks;
            const size_t act_stride = (size_t)LA_MB_MAX * 64 * c.ffn, slab_stride = (size_t)ex_ks * npass_rows * c.hidden;
            // M >= 128: every expert works on the rows it received, packed into their own blocks (la_mblock.hip "Gathered MoE")
            const bool gathered = nblk >= 2;
            const long xg_stride = (long)LA_MB_MAX * 64 * c.hidden;
            if (gathered) {
                if (!(g_la_ex_split & 4)) {              // round 5: plan + gather in one launch (la_lab_set(16, 4) = the two launches of round 3)
                    KCHK(lk_mb_moe_plan_gather(st, route_w, M, m->mb_xp, c.hidden, nblk, c.n_experts, m->mb_xg, xg_stride, m->mb_moe_perm,
                                               m->mb_moe_pos, m->mb_moe_cnt));
                } else {
                    KCHK(lk_mb_moe_plan(st, route_w, M, c.n_experts, m->mb_moe_perm, m->mb_moe_pos, m->mb_moe_cnt));
                    KCHK(lk_mb_moe_gather(st, m->mb_xp, m->mb_moe_perm, m->mb_moe_cnt, c.hidden, nblk, c.n_experts, m->mb_xg, xg_stride));
                }
            }
            if (gathered && m->ex_merged && !(g_la_ex_split & 1)) {
                // equally spaced expert images: ONE gate/up launch and ONE down launch for all experts (grid.z = expert x pass)
                MbGemm g{}; g.wp = m->ex_gateup[(size_t)l * c.n_experts]; g.xp = m->mb_xg; g.N = c.ffn; g.K = c.hidden; g.nblk = nblk;
                g.n_wg = c.balanced_wg[1]; g.ksplit = 1; g.act_xp = m->mb_act_ex; g.nblk_dev = m->mb_moe_cnt + LA_MOE_MAX_E;
                g.ex_n = c.n_experts; g.ex_w_stride = m->ex_gu_stride[l]; g.ex_x_stride = xg_stride; g.ex_o_stride = (long)act_stride;
                KCHK(lk_mb_gemm(st, 1, g));
                MbGemm d{}; d.wp = m->ex_down[(size_t)l * c.n_experts]; d.xp = m->mb_act_ex; d.N = c.hidden; d.K = c.ffn; d.nblk = nblk;
                d.ksplit = ex_ks; d.slabs = m->mb_slabs_ex; d.slab_rows = npass_rows; d.nblk_dev = m->mb_moe_cnt + LA_MOE_MAX_E;
                d.ex_n = c.n_experts; d.ex_w_stride = m->ex_dn_stride[l]; d.ex_x_stride = (long)act_stride; d.ex_o_stride = (long)slab_stride;
                KCHK(lk_mb_gemm(st, 0, d));
            } else
            for (int e = 0; e < c.n_experts; ++e) {
                MbGemm g{}; g.wp = m->ex_gateup[(size_t)l * c.n_experts + e]; g.xp = gathered ? m->mb_xg + e * xg_stride : m->mb_xp;
                g.N = c.ffn; g.K = c.hidden; g.nblk = nblk;
                g.n_wg = c.balanced_wg[1]; g.ksplit = 1; g.act_xp = m->mb_act_ex + e * act_stride;
                if (gathered) g.nblk_dev = m->mb_moe_cnt + LA_MOE_MAX_E + e; else g.route_col = route_w + e;
                KCHK(lk_mb_gemm(st, 1, g));
                MbGemm d{}; d.wp = m->ex_down[(size_t)l * c.n_experts + e]; d.xp = m->mb_act_ex + e * act_stride; d.N = c.hidden; d.K = c.ffn;
                d.nblk = nblk; d.ksplit = m->down_ks; d.slabs = m->mb_slabs_ex + e * slab_stride; d.slab_rows = npass_rows;
                if (gathered) d.nblk_dev = m->mb_moe_cnt + LA_MOE_MAX_E + e; else d.route_col = route_w + e;
                KCHK(lk_mb_gemm(st, 0, d));
            }
            if (!(g_la_ex_split & 1)) {
                // accumulation in expert order + residual + next RMSNorm in one launch (the accumulated row stays in registers)
                KCHK(lk_mb_moe_accum_norm(st, m->mb_slabs_ex, (long)slab_stride, ex_ks, npass_rows, route_w, c.n_experts, c.hidden, M,
                                          gathered ? m->mb_moe_pos : nullptr, m->mb_h, nw, c.rms_eps, m->mb_xp, cf));
                continue;
            }
            KCHK(lk_mb_moe_accum(st, m->mb_slabs_ex, (long)slab_stride, m->down_ks, npass_rows, route_w, c.n_experts, c.hidden,
                                 m->mb_moe_acc, M, gathered ? m->mb_moe_pos : nullptr));
            KCHK(lk_mb_resid_norm_addend(st, m->mb_h, m->mb_moe_acc, nw, c.hidden, c.rms_eps, m->mb_xp, M, cf));
            continue;
        }
        KCHK(lk_mb_resid_norm(st, m->mb_h, m->mb_slabs, o_ks, npass_rows, L.norm2, c.hidden, c.rms_eps, m->mb_xp, M, cf));
        MbGemm g{}; g.wp = L.wgateup; g.xp = m->mb_xp; g.N = c.ffn; g.K = c.hidden; g.nblk = nblk; g.n_wg = c.balanced_wg[1]; g.ksplit = 1;
        g.act_xp = m->mb_act;
        KCHK(lk_mb_gemm(st, 1, g));
        MbGemm d{}; d.wp = L.wdown; d.xp = m->mb_act; d.N = c.hidden; d.K = c.ffn; d.nblk = nblk; d.ksplit = down_ks;
        d.slabs = m->mb_slabs; d.slab_rows = npass_rows;
        KCHK(lk_mb_gemm(st, 0, d));
        KCHK(lk_mb_resid_norm(st, m->mb_h, m->mb_slabs, down_ks, npass_rows, nw, c.hidden, c.rms_eps, m->mb_xp, M, cf));
    }
    const int lwg = lk_mb_logits_wgs(c.vocab, c.balanced_wg[2]);
    MbGemm h{}; h.wp = m->w.lm_head; h.xp = m->mb_xp; h.N = c.vocab; h.K = c.hidden; h.nblk = nblk; h.n_wg = c.balanced_wg[2]; h.ksplit = 1;
    h.logits = m->mb_logits; h.cand_val = m->mb_cand_val; h.cand_idx = m->mb_cand_idx;
    KCHK(lk_mb_gemm(st, 3, h));
    KCHK(lk_mb_argmax(st, m->mb_cand_val, m->mb_cand_idx, lk_mb_cand_slots(lwg), nblk, m->mb_out + LA_MOUT_ARGMAX));
    KCHK(lk_mb_accept_scan(st, m->mb_meta, m->mb_ids, m->mb_rowmask, (const uint64_t*)(m->mb_in + LA_MIN_XMASK), m->mb_out + LA_MOUT_ARGMAX, nblk, c.max_keys, c.kv_ring ? 1 : 0, m->bstate, m->mb_out));
    KCHK(lk_mb_kv_commit(st, m->mb_kfresh, m->mb_vfresh, m->kmain, m->vmain, m->mb_out, nblk, c.n_layers, c.n_kv_heads, m->total_keys));
    return LA_OK;
}

static int mstep(la_llama* m, void* stream, const int32_t* host_in, int32_t* host_out, bool eager) {
    if (!m || !host_in) return LA_E_ARG;
    const int nblk = host_in[LA_MIN_NBLK];
    if (!m->mb_max || nblk < 1 || nblk > m->mb_max) { la_set_error("mstep: block count outside cfg.max_blocks"); return LA_E_ARG; }
    bool wide = false;                                   // any wide-tree piece in this pass: the masked attention instantiation
    for (int b = 0; b < nblk; ++b) wide = wide || host_in[LA_MIN_BLK + 4 * b + 2] == LA_MODE_TREE_PIECE;
    const int gi = (wide ? LA_MB_MAX + 1 : 0) + nblk;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(m->mb_in, host_in, LA_MIN_WORDS * sizeof(int), hipMemcpyHostToDevice, st));
    if (eager) {
        int rc = enqueue_mstep(m, st, nblk, wide);
        if (rc != LA_OK) return rc;
    } else {
        if (m->mready[gi] && m->mepoch[gi] != g_la_graph_epoch) {
            (void)hipGraphExecDestroy(m->mgraphs[gi]);
            m->mgraphs[gi] = nullptr; m->mready[gi] = false;
        }
        if (!m->mready[gi]) {
            hipGraph_t g = nullptr;
            HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            int rc = enqueue_mstep(m, st, nblk, wide);
            hipError_t e = hipStreamEndCapture(st, &g);
            if (rc != LA_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
            HIPCHK(e);
            HIPCHK(hipGraphInstantiate(&m->mgraphs[gi], g, nullptr, nullptr, 0));
            (void)hipGraphDestroy(g);
            m->mready[gi] = true; m->mepoch[gi] = g_la_graph_epoch;
        }
        HIPCHK(hipGraphLaunch(m->mgraphs[gi], st));
    }
    if (host_out) HIPCHK(hipMemcpyAsync(host_out, m->mb_out, LA_MOUT_DST * sizeof(int), hipMemcpyDeviceToHost, st));
    return LA_OK;
}
// Drafts from the device trie: the step input is assembled on the device (k_mb_fill_from_trie) from the trie kernel's outputs, then
// the captured graph of `nblk` plain blocks runs — nothing crosses PCIe but the result header.
extern "C" int la_llama_mstep_trie(la_llama* m, void* stream, int nblk, const int32_t* slots, const int32_t* limits, const int32_t* last_tok,
                                   const int32_t* d_ids, const uint64_t* d_rowmask, const int32_t* d_n, int32_t* host_out) {
    if (!m || !slots || !limits || !last_tok || !d_ids || !d_rowmask || !d_n) return LA_E_ARG;
    if (!m->mb_max || nblk < 1 || nblk > m->mb_max) { la_set_error("mstep_trie: block count outside cfg.max_blocks"); return LA_E_ARG; }
    for (int b = 0; b < nblk; ++b) if (slots[b] < 0 || slots[b] >= m->n_slots) return LA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    KCHK(lk_mb_fill_from_trie(st, d_ids, d_rowmask, d_n, slots, limits, last_tok, nblk, m->mb_in));
    const int gi = nblk;
    if (m->mready[gi] && m->mepoch[gi] != g_la_graph_epoch) {
        (void)hipGraphExecDestroy(m->mgraphs[gi]);
        m->mgraphs[gi] = nullptr; m->mready[gi] = false;
    }
    if (!m->mready[gi]) {
        hipGraph_t g = nullptr;
        HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        int rc = enqueue_mstep(m, st, nblk, false);
        hipError_t e = hipStreamEndCapture(st, &g);
        if (rc != LA_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
        HIPCHK(e);
        HIPCHK(hipGraphInstantiate(&m->mgraphs[gi], g, nullptr, nullptr, 0));
        (void)hipGraphDestroy(g);
        m->mready[gi] = true; m->mepoch[gi] = g_la_graph_epoch;
    }
    HIPCHK(hipGraphLaunch(m->mgraphs[gi], st));
    if (host_out) HIPCHK(hipMemcpyAsync(host_out, m->mb_out, LA_MOUT_DST * sizeof(int), hipMemcpyDeviceToHost, st));
    return LA_OK;
}

extern "C" int la_llama_mstep(la_llama* m, void* stream, const int32_t* host_in, int32_t* host_out) {
    return mstep(m, stream, host_in, host_out, false);
}
extern "C" int la_llama_mstep_eager(la_llama* m, void* stream, const int32_t* host_in, int32_t* host_out) {
    return mstep(m, stream, host_in, host_out, true);
}

extern "C" int la_llama_step(la_llama* m, void* stream, const int32_t* host_in, int32_t* host_out) {
    if (!m || !host_in || !host_out) return LA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    // other staging blocks, or a capture-time knob (la_debug_set keys 7 / 8) changed since the capture: capture again
    if ((m->graph_ready || m->graph_long_ready) && (m->zc_in != host_in || m->zc_out != host_out || m->graph_epoch != g_la_graph_epoch)) {
        if (m->graph_exec) (void)hipGraphExecDestroy(m->graph_exec);
        if (m->graph_long) (void)hipGraphExecDestroy(m->graph_long);
        m->graph_exec = m->graph_long = nullptr;
        m->graph_ready = m->graph_long_ready = false;
    }
    // two captured forms of the step: the tree attention as ONE launch while the K/V of the kv heads of an XCD fit its L2 (every
    // workgroup of a head streams the head's whole K/V, the sharers find it in L2), key splits + combine beyond (distinct bytes per
    // CU).  The caller's hint word picks; both are exact at any context.
    const bool long_ctx = host_in[LA_IN_NKEYS_HINT] + LA_TREE_MAX > m->attn_thr;
    if (!(long_ctx ? m->graph_long_ready : m->graph_ready)) {
        hipPointerAttribute_t pa;
        if (hipPointerGetAttributes(&pa, host_in) != hipSuccess || hipPointerGetAttributes(&pa, host_out) != hipSuccess) {
            (void)hipGetLastError();
            la_set_error("la_llama_step: host_in / host_out must be pinned host memory (zero-copy step I/O)");
            return LA_E_ARG;
        }
        int rc = build_graph(m, st, false, host_in, host_out, 0, long_ctx);
        if (rc != LA_OK) return rc;
        m->zc_in = host_in; m->zc_out = host_out; m->graph_epoch = g_la_graph_epoch;
        // the OTHER form too, now (first call = set-up / prefill time) rather than on the first step that crosses attn_thr: capture +
        // instantiate of the whole model is a multi-ms stall that would otherwise land inside a decode loop (and inside
        // la_lookahead_decode).  Only when the cache can reach the other side of the threshold at all.
        // The speculative capture is an optimisation, never a condition of THIS step: a failure is dropped (the form is captured lazily by the
        // step that first needs it, which then reports its own error), and a recapture in the middle of a sequence (a lab knob / epoch change)
        // takes the other form only when the context is within two steps' worth of rows of the threshold.
        const bool other_reachable = long_ctx ? true : (m->cfg.max_keys > m->attn_thr);
        const int dist_thr = host_in[LA_IN_NKEYS_HINT] + LA_TREE_MAX - m->attn_thr;
        const bool mid_sequence = host_in[LA_IN_NKEYS_HINT] > 0 && m->seq_expected > 0;
        const bool worth_it = !mid_sequence || (dist_thr > -2 * LA_TREE_MAX && dist_thr < 2 * LA_TREE_MAX);
        if (other_reachable && worth_it && !(long_ctx ? m->graph_ready : m->graph_long_ready)) {
            if (build_graph(m, st, false, host_in, host_out, 0, !long_ctx) != LA_OK) {
                (void)hipGetLastError();                 // clear the sticky error of the failed capture; the required graph is ready
                la_set_error("");
            }
        }
    }
    m->seq_expected += 1;
    HIPCHK(hipGraphLaunch(long_ctx ? m->graph_long : m->graph_exec, st));
    return LA_OK;
}

extern "C" int la_llama_wait(la_llama* m, void* stream) {
    if (!m) return LA_E_ARG;
    if (m->zc_out) {
        volatile int32_t* flag = m->zc_out + LA_ST_SEQ;
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (*flag != m->seq_expected) {
            if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (*flag == m->seq_expected) return LA_OK;
    }
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (m->zc_out && m->zc_out[LA_ST_SEQ] != m->seq_expected) { la_set_error("la_llama_wait: step was not published"); return LA_E_HIP; }
    return LA_OK;
}

extern "C" int la_llama_step_eager(la_llama* m, void* stream, const int32_t* host_in, int32_t* host_out) {
    if (!m || !host_in) return LA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(m->in, host_in, LA_IN_WORDS * sizeof(int), hipMemcpyHostToDevice, st));
    int rc = enqueue_step(m, st, nullptr, false, nullptr, nullptr, 0, host_in[LA_IN_NKEYS_HINT] + LA_TREE_MAX > m->attn_thr);
    if (rc != LA_OK) return rc;
    if (host_out)
        HIPCHK(hipMemcpyAsync(host_out, m->state, (LA_ST_OUTTOK + 64) * sizeof(int), hipMemcpyDeviceToHost, st));
    return LA_OK;
}

extern "C" int la_lookahead_decode(la_llama* m, la_cache* c, void* stream, const la_decode_params* p, int32_t* seq,
                                   int32_t* seq_len, int32_t* host_in, int32_t* host_out, int32_t* dls, int32_t* edls,
                                   int32_t* n_steps, int32_t* finished, double* fts, double* qts) {
    if (!m || !c || !p || !seq || !seq_len || !host_in || !host_out || !n_steps || p->decoding_length > LA_TREE_MAX ||
        p->max_query_length < 1 || p->max_query_length > 8 || p->n_eos < 0 || p->n_eos > 8) return LA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    int32_t ids[LA_TREE_MAX], parent[LA_TREE_MAX], sizes[2], nsz = 0, T = 0;
    uint64_t rows[LA_TREE_MAX];
    int len = *seq_len, steps = 0, done = 0;
    auto t_prev = std::chrono::steady_clock::now();
    int nkeys = len - 1;                           // committed keys = context without the newest token
    while (steps < p->max_steps) {
        const int ubl = std::min(p->branch_length, p->max_length - len - 1);          // pretrained_model.py:680
        if (ubl < 0) { la_set_error("decode: no room left below max_length"); return LA_E_RANGE; }
        const int nq = std::min(p->max_query_length, len);
        auto t0 = std::chrono::steady_clock::now();
        int rc = la_cache_hier_get(c, seq + len - nq, nq, p->decoding_length, ubl, 0, std::max(p->decoding_length / 2, 1),
                                   p->mode, p->idx, LA_TREE_MAX, ids, parent, rows, nullptr, sizes, &nsz, &T);
        if (rc != LA_OK) return rc;
        if (T == 0) { ids[0] = seq[len - 1]; rows[0] = 1ull; T = 1; }
        if (!m->cfg.kv_ring && nkeys + T > m->cfg.max_keys) { la_set_error("decode: KV cache capacity exceeded"); return LA_E_RANGE; }
        if (nkeys + T + 1 > m->cfg.max_pos) { la_set_error("decode: position beyond the RoPE tables"); return LA_E_RANGE; }
        host_in[LA_IN_T] = T;
        host_in[LA_IN_MODE] = 0;
        host_in[LA_IN_NKEYS_HINT] = nkeys;
        memcpy(host_in + LA_IN_IDS, ids, sizeof(int32_t) * T);
        memcpy(host_in + LA_IN_ROWMASK, rows, sizeof(uint64_t) * T);
        if (qts) qts[steps] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        rc = la_llama_step(m, st, host_in, host_out);
        if (rc != LA_OK) return rc;
        rc = la_llama_wait(m, st);
        if (rc != LA_OK) return rc;
        const int n = host_out[LA_ST_NOUT];
        nkeys = host_out[LA_ST_NKEYS];
        if (n < 1 || n > LA_TREE_MAX) { la_set_error("decode: bad step output"); return LA_E_HIP; }   // a step emits <= branch_length + 1 <= T tokens
        memcpy(seq + len, host_out + LA_ST_OUTTOK, sizeof(int32_t) * n);
        rc = la_cache_stream_put(c, seq + len, n, p->branch_length + 1, 0, p->idx);
        if (rc != LA_OK) return rc;
        if (dls) dls[steps] = T;
        if (edls) edls[steps] = n;
        if (fts) {
            auto t1 = std::chrono::steady_clock::now();
            fts[steps] = std::chrono::duration<double>(t1 - t_prev).count();
            t_prev = t1;
        }
        ++steps;
        bool eos = false;
        for (int i = 0; i < n && !eos; ++i)
            for (int e = 0; e < p->n_eos; ++e) if (seq[len + i] == p->eos[e]) { eos = true; break; }
        len += n;
        if (len >= p->max_length || eos) { done = 1; break; }
    }
    *seq_len = len;
    *n_steps = steps;
    if (finished) *finished = done;
    return LA_OK;
}

// Host-decided commit after a mode-2 (verify only) step: rows[0..n) of the last block become main-cache rows
// nkeys..nkeys+n (the sequential accept of pretrained_model.py:825-875 with a non-empty logits-processor list).
extern "C" int la_llama_commit(la_llama* m, void* stream, const int32_t* rows, int n, int32_t* host_out) {
    if (!m || !rows || n < 1 || n > LA_TREE_MAX) return LA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    int hdr[LA_ST_WORDS];
    HIPCHK(hipMemcpyAsync(hdr, m->state, sizeof(hdr), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < n; ++i) {
        if (rows[i] < 0 || rows[i] >= hdr[LA_ST_T]) { la_set_error("commit: row outside the last block"); return LA_E_RANGE; }
        hdr[LA_ST_SRCIDX + i] = rows[i];
    }
    hdr[LA_ST_DSTBASE] = hdr[LA_ST_NKEYS];
    hdr[LA_ST_NCOMMIT] = n;
    hdr[LA_ST_NKEYS] += n;
    HIPCHK(hipMemcpyAsync(m->state, hdr, sizeof(hdr), hipMemcpyHostToDevice, st));
    KCHK(lk_kv_commit(st, m->kfresh, m->vfresh, m->kmain, m->vmain, m->state, m->cfg.n_layers, m->cfg.n_kv_heads, m->total_keys,
                      m->cfg.kv_ring ? m->cfg.max_keys : 0));
    if (host_out) HIPCHK(hipMemcpyAsync(host_out, m->state, (LA_ST_OUTTOK + 64) * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));       // hdr lives on this stack frame
    return LA_OK;
}

// Host-decided commit after a cursor-batch step whose slots ran in mode 2 (the batch twin of la_llama_commit: the per-sample
// sequential accept walk of pretrained_model_batch.py:814-931 with a non-empty logits-processor list happens on the host over the
// logits rows).  keep[r] for block row r = k >= 0: the row is the k-th kept key of its slot (k = 0: the root) and moves to main-
// cache row cursor + k of that slot (the in-place moves of :893-904 / _update_cache :982-985); -1: dropped.  The slots' cursors
// advance by their kept rows.  d2h of the LA_BST_DST header words into host_out, stream synchronised.
extern "C" int la_llama_bcommit(la_llama* m, void* stream, const int32_t* keep, int32_t* host_out) {
    if (!m || !keep || m->n_slots < 1) return LA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const la_llama_config& c = m->cfg;
    std::vector<int> bs(LA_BST_WORDS);
    HIPCHK(hipMemcpyAsync(bs.data(), m->bstate, LA_BST_WORDS * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int cnt[LA_MAX_SEQ] = {0};
    unsigned seen[LA_MAX_SEQ] = {0};
    for (int r = 0; r < LA_TREE_MAX; ++r) {
        const int sl = bs[LA_BST_SEQ + r], k = keep[r];
        bs[LA_BST_DST + r] = -1;
        if (k < 0) continue;
        if (sl < 0 || sl >= m->n_slots || k >= 32 || (seen[sl] >> k & 1u)) { la_set_error("bcommit: bad keep plan"); return LA_E_RANGE; }
        seen[sl] |= 1u << k;
        ++cnt[sl];
        const int pos = bs[LA_BST_NKEYS + sl] + k;
        bs[LA_BST_DST + r] = sl * c.max_keys + (c.kv_ring ? pos % c.max_keys : pos);
    }
    for (int sl = 0; sl < m->n_slots; ++sl) {
        if (seen[sl] != (cnt[sl] >= 32 ? 0xffffffffu : (1u << cnt[sl]) - 1u)) { la_set_error("bcommit: kept positions of a slot are not 0..n-1"); return LA_E_RANGE; }
        if (!c.kv_ring && bs[LA_BST_NKEYS + sl] + cnt[sl] > c.max_keys) { la_set_error("bcommit: KV capacity of the slot exceeded"); return LA_E_RANGE; }
        bs[LA_BST_NKEYS + sl] += cnt[sl];
        bs[LA_BST_NOUT + sl] = 0;
    }
    HIPCHK(hipMemcpyAsync(m->bstate, bs.data(), LA_BST_ARGMAX * sizeof(int), hipMemcpyHostToDevice, st));   // NKEYS, NOUT, OUTTOK, DST
    KCHK(lk_kv_commit_b(st, m->kfresh, m->vfresh, m->kmain, m->vmain, m->bstate, c.n_layers, c.n_kv_heads, m->total_keys));
    if (host_out) HIPCHK(hipMemcpyAsync(host_out, m->bstate, LA_BST_DST * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));       // bs lives on this stack frame
    return LA_OK;
}

// The same after a multi-block step (la_llama_mstep) whose blocks ran in mode 2: keep[b * 64 + r] for row r of block b.
extern "C" int la_llama_mcommit(la_llama* m, void* stream, int nblk, const int32_t* keep, int32_t* host_out) {
    if (!m || !keep || !m->mb_max || nblk < 1 || nblk > m->mb_max) return LA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const la_llama_config& c = m->cfg;
    std::vector<int> meta(nblk * LA_MB_META), out(LA_MOUT_ARGMAX), nk(LA_MAX_SEQ);
    HIPCHK(hipMemcpyAsync(meta.data(), m->mb_meta, meta.size() * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(nk.data(), m->bstate + LA_BST_NKEYS, LA_MAX_SEQ * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < LA_MOUT_ARGMAX; ++i) out[i] = (i >= LA_MOUT_DST) ? -1 : 0;
    // one mode-2 block per slot, optionally followed by the mode-3 pieces of the same wide tree: the kept positions k of a
    // slot are 0..n-1 over all its blocks (n <= LA_MOUT_TOKS)
    unsigned used = 0;
    unsigned long long seen[LA_MAX_SEQ] = {0};
    int cnt[LA_MAX_SEQ] = {0}, base_nk[LA_MAX_SEQ];
    for (int i = 0; i < LA_MAX_SEQ; ++i) base_nk[i] = nk[i];
    int prev_slot = -1;
    for (int b = 0; b < nblk; ++b) {
        const int* mt = &meta[b * LA_MB_META];
        const int sl = mt[LA_MBM_SLOT], T = mt[LA_MBM_T], md = mt[LA_MBM_MODE];
        const bool piece = md == LA_MODE_TREE_PIECE && sl == prev_slot;
        if (sl < 0 || sl >= m->n_slots || (!piece && ((used >> sl & 1u) || md != 2))) {
            la_set_error("mcommit: the last step's blocks were not one mode-2 block (+ its wide-tree pieces) per slot"); return LA_E_RANGE;
        }
        used |= 1u << sl;
        prev_slot = sl;
        for (int r = 0; r < LA_TREE_MAX; ++r) {
            const int k = keep[b * 64 + r];
            if (k < 0) continue;
            if (r >= T || k >= LA_MOUT_TOKS || (seen[sl] >> k & 1ull)) { la_set_error("mcommit: bad keep plan"); return LA_E_RANGE; }
            seen[sl] |= 1ull << k;
            ++cnt[sl];
            const int pos = base_nk[sl] + k;
            out[LA_MOUT_DST + b * 64 + r] = sl * c.max_keys + (c.kv_ring ? pos % c.max_keys : pos);
        }
    }
    for (int sl = 0; sl < LA_MAX_SEQ; ++sl) {
        if (!(used >> sl & 1u)) continue;
        if (seen[sl] != ((1ull << cnt[sl]) - 1ull)) { la_set_error("mcommit: kept positions of a sequence are not 0..n-1"); return LA_E_RANGE; }
        if (!c.kv_ring && nk[sl] + cnt[sl] > c.max_keys) { la_set_error("mcommit: KV capacity of the slot exceeded"); return LA_E_RANGE; }
        nk[sl] += cnt[sl];
    }
    for (int i = 0; i < LA_MAX_SEQ; ++i) out[LA_MOUT_NKEYS + i] = nk[i];
    HIPCHK(hipMemcpyAsync(m->bstate + LA_BST_NKEYS, nk.data(), LA_MAX_SEQ * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(m->mb_out, out.data(), LA_MOUT_ARGMAX * sizeof(int), hipMemcpyHostToDevice, st));    // NOUT, NKEYS, OUTTOK, DST
    KCHK(lk_mb_kv_commit(st, m->mb_kfresh, m->mb_vfresh, m->kmain, m->vmain, m->mb_out, nblk, c.n_layers, c.n_kv_heads, m->total_keys));
    if (host_out) HIPCHK(hipMemcpyAsync(host_out, m->mb_out, LA_MOUT_DST * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return LA_OK;
}

// Set the committed-key cursor of a slot from the host (slot 0 is also the single-sequence cursor LA_ST_NKEYS): lets a
// prompt prefilled by la_llama_mstep continue on la_llama_step, and a finished slot be rewound without clearing the others.
extern "C" int la_llama_set_nkeys(la_llama* m, void* stream, int slot, int nkeys) {
    if (!m || slot < 0 || slot >= m->n_slots || nkeys < 0 || (!m->cfg.kv_ring && nkeys > m->cfg.max_keys)) return LA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(m->bstate + LA_BST_NKEYS + slot, &nkeys, sizeof(int), hipMemcpyHostToDevice, st));
    if (slot == 0) HIPCHK(hipMemcpyAsync(m->state + LA_ST_NKEYS, &nkeys, sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    return LA_OK;
}

extern "C" void* la_llama_buffer(la_llama* m, int which) {
    if (!m) return nullptr;
    switch (which) {
        case 0: return m->logits;
        case 1: return m->state;
        case 2: return m->h;
        case 3: return m->xp;
        case 4: return m->kmain;
        case 5: return m->vmain;
        case 6: return m->kfresh;
        case 7: return m->vfresh;
        case 8: return m->bstate;
        case 9: return m->route_w;
        case 10: return m->moe_acc;
        case 11: return m->mb_max ? m->mb_logits : nullptr;
        case 12: return m->mb_max ? m->mb_out : nullptr;
        case 13: return m->mb_max ? m->mb_h : nullptr;
        case 14: return m->mb_max ? m->mb_route_w : nullptr;
        default: return nullptr;
    }
}

// Mean kernel-class durations of one step measured with HIP events on `stream` (eager launches).
// out_ms[0..6] = per-step sum of {qkv, o, gate/up, down, lm_head, attention(+combine), other};
// out_ms[7] = whole step; out_launches[0..6] = launches of that class per step.
// The sequence state is saved and restored so profiling does not advance the context.
// Duration of the dominant kernel without the event packets of la_llama_profile between launches: the gate/up launch of every
// layer (its own weights, the activation operand left by the last step), back to back, bracketed by ONE pair of HIP events per
// pass; out_ms = mean time per launch (kernel + the dependent-launch boundary, ~1.5 us).  Dense models with the balanced image.
extern "C" int la_llama_profile_gateup(la_llama* m, void* stream, int iters, float* out_ms) {
    if (!m || iters <= 0 || !out_ms) return LA_E_ARG;
    const la_llama_config& c = m->cfg;
    if (c.n_experts > 0 || c.balanced_wg[1] <= 0) { la_set_error("profile_gateup: dense model with the balanced gate/up image only"); return LA_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    double acc = 0;
    for (int it = 0; it <= iters; ++it) {                 // pass 0 warms up
        HIPCHK(hipEventRecord(e0, st));
        for (int l = 0; l < c.n_layers; ++l)
            KCHK(lk_gemm64r_swiglu(st, m->layers[l].wgateup, m->xp, c.ffn, c.hidden, c.balanced_wg[1], m->act_xp));
        HIPCHK(hipEventRecord(e1, st));
        HIPCHK(hipStreamSynchronize(st));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0) acc += ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *out_ms = (float)(acc / iters / c.n_layers);
    return LA_OK;
}

extern "C" int la_llama_profile(la_llama* m, void* stream, const int32_t* host_in, int iters,
                                float* out_ms, int32_t* out_launches) {
    if (!m || !host_in || iters <= 0 || !out_ms) return LA_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    int saved[LA_ST_WORDS];
    HIPCHK(hipMemcpyAsync(m->in, host_in, LA_IN_WORDS * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(saved, m->state, sizeof(saved), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    double acc[KC_N + 1] = {0};
    int launches[KC_N] = {0};
    Prof pf; pf.on = true; pf.st = st;
    for (int it = 0; it < iters; ++it) {
        pf.used = 0; pf.cls.clear();
        HIPCHK(hipMemcpyAsync(m->state, saved, sizeof(saved), hipMemcpyHostToDevice, st));
        int rc = enqueue_step(m, st, &pf);
        if (rc != LA_OK) return rc;
        HIPCHK(hipStreamSynchronize(st));
        for (size_t i = 0; i + 1 < pf.used; ++i) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, pf.ev[i], pf.ev[i + 1]));
            acc[pf.cls[i]] += ms;
            if (it == 0) launches[pf.cls[i]]++;
        }
        float tot = 0;
        HIPCHK(hipEventElapsedTime(&tot, pf.ev[0], pf.ev[pf.used - 1]));
        acc[KC_N] += tot;
    }
    HIPCHK(hipMemcpyAsync(m->state, saved, sizeof(saved), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < KC_N; ++i) out_ms[i] = (float)(acc[i] / iters);
    out_ms[KC_N] = (float)(acc[KC_N] / iters);
    if (out_launches) for (int i = 0; i < KC_N; ++i) out_launches[i] = launches[i];
    for (auto e : pf.ev) (void)hipEventDestroy(e);
    return LA_OK;
}
