/*
 * lookahead_hip.h — C ABI of liblookahead_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for LOOKAHEAD's trie-draft / tree-verify decoding loop
 * (alipay/PainlessInferenceAcceleration, lookahead/).  The reference has no
 * native boundary: its hot path is two Python surfaces.  Every entry point
 * below names the reference symbol (file:line, relative to the reference
 * checkout) whose behaviour it replaces; the Python package
 * painlessinferenceacceleration_amd/ re-exposes them under the reference's
 * own names (LookaheadCache, lookahead_generation).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every function returns int: 0 = ok, <0 = LA_E_* error (never throws).
 *   - handles are opaque; buffers are caller-owned; device pointers are raw
 *     HIP device addresses; `stream` is a hipStream_t passed as void*.
 *   - no allocation on the per-step path (la_llama_step / la_cache_*_get).
 *   - la_cache_* is host code (works without a GPU); everything taking a
 *     `stream` launches hand-written gfx950 kernels and needs a device.
 */
#ifndef LOOKAHEAD_HIP_H
#define LOOKAHEAD_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LA_OK            0
#define LA_E_ARG        -1   /* bad argument (the reference would `assert`)          */
#define LA_E_RANGE      -2   /* output buffer too small / size over a device limit   */
#define LA_E_HIP        -3   /* HIP runtime error (see la_last_error)                */
#define LA_E_STATE      -4   /* call order violated (e.g. step before create)        */
#define LA_E_IO         -5   /* save/load failure                                    */

#define LA_MODE_INPUT    0
#define LA_MODE_OUTPUT   1
#define LA_MODE_MIX      2

#define LA_TREE_MAX     64   /* tree tokens per sequence handled by the device path  */
#define LA_MOE_MAX_E      8   /* experts per mixture-of-experts layer (Mixtral: 8, top-2) */

/* ABI version: bumped when a signature changes. */
#define LA_ABI_VERSION  12   /* bumped whenever a struct layout or an entry point changes */
int          la_abi_version(void);   /* == LA_ABI_VERSION of the header the library was built from */
/* Storage / MFMA-input type of the loaded library: 0 = bfloat16 (liblookahead_hip.so), 1 = float16 (liblookahead_hip_f16.so, the
 * dtype the reference's examples and benchmarks run, lookahead/benchmarks/llama_benchmark.py:27).  The two libraries are the same
 * sources compiled twice and export this same ABI; every `bf16` buffer below holds the library's 16-bit type. */
#define LA_DTYPE_BF16    0
#define LA_DTYPE_F16     1
int          la_abi_dtype(void);
const char*  la_last_error(void);
/* Parity aid (tests/test_gpu_e2e.py, per-depth residual probe): key 13 = n > 0: the single-sequence step runs the first n layers,
 * then the final norm + lm_head (0 = the whole model).  No other key is accepted here: the measurement knobs and A/B switches of
 * the kernel lab live behind la_lab_* (include/lookahead_hip_lab.h), which this header does not declare. */
int          la_debug_set(int key, int value);
int          la_debug_get(int key);

/* ------------------------------------------------------------------------
 * 1. Trie cache (host).  Replaces class LookaheadCache / Tree / Node,
 *    lookahead/lookahead/common/lookahead_cache.py:13-587.
 * --------------------------------------------------------------------- */
typedef struct la_cache la_cache;

/* LookaheadCache.__init__ (lookahead_cache.py:337-347). */
la_cache* la_cache_create(int max_node, int max_output_node);
void      la_cache_destroy(la_cache* c);
/* attrs mutated by callers: .max_node/.max_output_node (benchmarks/benchmark.py:271-272),
 * .eos_ids/.stop_words (pretrained_model.py:1088-1089).  n_eos==0 <=> eos_ids=[None]. */
int la_cache_set_limits(la_cache* c, int max_node, int max_output_node);
int la_cache_set_eos(la_cache* c, const int32_t* eos_ids, int n_eos);
int la_cache_set_stop_words(la_cache* c, const int32_t* ids, int n);
/* fresh() (lookahead_cache.py:563-564): drops the forest, keeps dirty-set counts and stream buffers. */
int la_cache_fresh(la_cache* c);
/* put() (lookahead_cache.py:349-373) -> Tree.put/_put/_pack (:33-63). mode: LA_MODE_INPUT|OUTPUT. */
int la_cache_put(la_cache* c, const int32_t* token_ids, int n, int branch_length,
                 int final_, int mode, int idx);
/* stream_put() (lookahead_cache.py:375-406). */
int la_cache_stream_put(la_cache* c, const int32_t* token_ids, int n, int branch_length,
                        int final_, int idx);
/* n stream_put calls in one (a batch step's accepted tokens, pretrained_model_batch.py:1254-1259): put k appends
 * toks[offsets[k] .. offsets[k + 1]) to slot idxs[k], in order. */
int la_cache_stream_put_many(la_cache* c, const int32_t* toks, const int32_t* offsets /*[n + 1]*/, const int32_t* idxs /*[n]*/, int n,
                             int branch_length, int final_);
/* hier_get() (lookahead_cache.py:408-439) -> Tree.get/_match/_dfs_get_freqs/_ravel (:65-154, 224-293).
 * Outputs (caller-owned): out_ids[cap], out_parent[cap] (index of the parent row, -1 for row 0),
 * out_rowmask: REQUIRED size cap * W words, W = ceil(decoding_length / 64) — every path writes whole rows of W words (bit j of
 *   row i <=> mask[i][j]; W = 1, i.e. decoding_length <= 64: one word per row, the layout of ABI <= 7); a caller that sized the
 *   buffer as out_rowmask[cap] must pass decoding_length <= 64 or grow it (the token_ids[-1:] fallbacks write row 0 = {1, 0, ...}),
 * out_mask (optional, row-major int64 [*out_n][*out_n], needs cap*cap entries),
 * out_sizes[2], *out_nsizes in {0,2} (the reference returns [] on the early exit, :413-414). */
int la_cache_hier_get(la_cache* c, const int32_t* token_ids, int n,
                      int decoding_length, int branch_length,
                      int min_input_size, int min_output_size, int mode, int idx,
                      int cap, int32_t* out_ids, int32_t* out_parent, uint64_t* out_rowmask,
                      int64_t* out_mask, int32_t out_sizes[2], int32_t* out_nsizes, int32_t* out_n);
/* one_get() (lookahead_cache.py:490-517) -> Tree.get_one_branch (:171-222).
 * *out_nsizes in {0,1,2}; the mask is lower-triangular of size *out_n. */
int la_cache_one_get(la_cache* c, const int32_t* token_ids, int n,
                     int decoding_length, int branch_length, int mode, int idx,
                     int cap, int32_t* out_ids, int32_t out_sizes[2], int32_t* out_nsizes,
                     int32_t* out_n);
/* par_get() (lookahead_cache.py:441-488): hier_get's draft re-laid as independent root-to-leaf chains (maximal ancestor sets only,
 * draft order, truncated to len(hier draft) - 1 rows) under a block mask in which a row sees the root and the rows of its own chain
 * up to itself.  Same buffers as la_cache_hier_get: out_ids[cap]; out_rowmask (optional) cap * W words, W = ceil(decoding_length / 64);
 * out_mask (optional) row-major int64 [*out_n][*out_n]; out_sizes[0] = rows behind the root, *out_nsizes = 1.  An empty query yields
 * *out_n = 0 (the reference raises IndexError there). */
int la_cache_par_get(la_cache* c, const int32_t* token_ids, int n,
                     int decoding_length, int branch_length,
                     int min_input_size, int min_output_size, int mode, int idx,
                     int cap, int32_t* out_ids, uint64_t* out_rowmask,
                     int64_t* out_mask, int32_t out_sizes[2], int32_t* out_nsizes, int32_t* out_n);
/* bat_get() (lookahead_cache.py:519-561) in its device-path form: the per-sample drafts of a batch in one call, unpadded
 * (no [bs,T,W] canvas).  queries: [bs][q_stride] (first nq[b] used); per-sample budget decoding_length // bs and
 * min_output_size = max(budget // 2, 1) as in the reference; one_branch != 0 selects one_get.  Outputs: rows of `cap`
 * (<= 64) entries: ids, 64-bit row masks, out_n[bs], out_sizes[bs][2], out_nsizes[bs]. */
int la_cache_bat_get_packed(la_cache* c, const int32_t* queries, const int32_t* nq, int q_stride, int bs, int decoding_length,
                            int branch_length, int mode, const int32_t* indices, int one_branch, int cap, int32_t* out_ids,
                            uint64_t* out_rowmask, int32_t* out_n, int32_t* out_sizes, int32_t* out_nsizes);
/* reset_input_freqs() (:566-570), squeeze_branch_counts() (:572-576). */
int la_cache_reset_input_freqs(la_cache* c, int idx);
int la_cache_squeeze(la_cache* c);
/* Introspection used by tests/benchmarks (len(cache.mem), sum of tree.n_node, ...). */
int la_cache_stats(la_cache* c, int64_t* n_trees, int64_t* n_nodes_live,
                   int64_t* n_dirty_trees, int64_t* n_dirty_input_trees);
int la_cache_tree_counters(la_cache* c, int32_t token, int64_t* n_node, int64_t* n_output_node);
/* save_mem()/load_mem() (:578-587): the reference pickles Python objects; this is a portable
 * little-endian arena snapshot ("LATRIE01").  Same role, different wire format. */
int la_cache_save(la_cache* c, const char* path);
int la_cache_load(la_cache* c, const char* path);

/* Incremental device mirror of the forest (the layout la_trie_hier_get_dev reads), owned by the host trie: every put /
 * stream_put / reset_input_freqs is logged as the handful of words it changed; squeeze / fresh / load mark the mirror for a
 * rebuild.  idx_planes: the input-frequency slots (LookaheadCache idx values, e.g. the batch indices 0..B-1) mirrored as fi
 * planes.  Per sync: la_cache_mirror_state -> if *full: la_cache_mirror_image into host buffers + upload, else
 * la_cache_mirror_patch + upload + la_trie_patch_dev on the stream that runs the queries. */
int la_cache_mirror_enable(la_cache* c, const int32_t* idx_planes, int n_planes);
int la_cache_mirror_state(la_cache* c, int32_t* n_records, int32_t* full, int32_t* n_ipatch, int32_t* n_dpatch);
int la_cache_mirror_image(la_cache* c, int32_t cap, int32_t* tok, double* fo, double* fi /*[planes][cap]*/, int32_t* cstart,
                          int32_t* ccount);
int la_cache_mirror_patch(la_cache* c, int32_t* ipatch /*[n_i][3]*/, int32_t* dkey /*[n_d][2]*/, double* dval /*[n_d]*/);
/* Apply a patch to the device image (one thread per word; fi planes are `fi_stride` records apart).  d_ccap (block capacities,
 * patch array 3) may be NULL when the image is only queried. */
int la_trie_patch_dev(void* stream, int32_t* d_tok, double* d_fo, double* d_fi, int64_t fi_stride, int32_t* d_cstart,
                      int32_t* d_ccount, int32_t* d_ccap, const int32_t* d_ipatch, int n_i, const int32_t* d_dkey, const double* d_dval,
                      int n_d);

/* Device-side trie UPDATE: LookaheadCache.stream_put(final=False, mode='output') (lookahead_cache.py:369-406, Tree.put/_put/_pack
 * :33-63) applied to the device image by the device, from tokens that are already in HBM (the accepted tokens of a verify step:
 * d_src_tok = la_llama_mstep's device output block + LA_MOUT_OUTTOK, stride LA_MOUT_TOKS, counts = the LA_MOUT_NOUT words).  The
 * image grows by the host mirror's rule and in the host's order, so the host REPLAYS the same puts on its trie afterwards
 * (la_cache_stream_put with the tokens it read back, then la_cache_mirror_discard) and the two agree on every live record (id,
 * token, child block, capacity, frequencies; dead copies left behind by moved blocks are never read again); no patch crosses
 * PCIe for these updates.  Everything else (input-mode put, final flush, reset_input_freqs, squeeze, load) stays a
 * host update that reaches the device as a patch or a full image, as before.
 *   la_trie_image          the device arrays: records [0, meta[0]) of `cap`; ccap = block capacities (la_cache_mirror_ccap);
 *                          meta int32[4] = {records in use, overflow (sticky), branches inserted, records appended};
 *                          root_of int32[n_root_of] = token -> record of its tree root (la_trie_root_index_dev rebuilds it after
 *                          every host image / patch).
 *   d_obuf / d_olen        int32[n_idx][128] / [n_idx]: the hold-back buffers _output_ids[idx] (la_cache_stream_buffer = the
 *                          host's copy, uploaded once when the device takes over).
 *   puts                   put k appends d_src_tok[k * src_stride ..][0 .. d_src_cnt[k]) (-1 entries dropped, cut at the first
 *                          eos id) to buffer d_put_idx[k]; indices must be distinct within a call; <= 64 puts, <= 40 tokens each.
 * An insert that would pass `cap` records sets meta[1] and stops all further device inserts: the host's replay passes the same
 * capacity at the same insert and uploads a larger image. */
typedef struct la_trie_image {
    int32_t* tok; double* fo; double* fi; int64_t fi_stride; int32_t n_planes;
    int32_t* cstart; int32_t* ccount; int32_t* ccap; int32_t* meta; int32_t cap;
    int32_t* root_of; int32_t n_root_of;
} la_trie_image;
int la_cache_mirror_ccap(la_cache* c, int32_t cap, int32_t* ccap);
int la_cache_mirror_discard(la_cache* c, int32_t* n_records);
int la_cache_stream_buffer(la_cache* c, int idx, int32_t cap, int32_t* out, int32_t* n);
int la_trie_root_index_dev(void* stream, const la_trie_image* img, int n_roots_max);
int la_trie_stream_put_dev(void* stream, const la_trie_image* img, int32_t* d_obuf, int32_t* d_olen, const int32_t* d_src_tok,
                           int src_stride, const int32_t* d_src_cnt, const int32_t* d_put_idx, int n_put, int branch_length,
                           const int32_t* d_stop, int n_stop, const int32_t* d_eos, int n_eos, int32_t* d_items /*[n_put][40][2]*/);
/* la_trie_hier_get_dev with one fi plane per query (d_plane[b], planes fi_stride records apart) and the per-query branch
 * length / stop rule of a batch step. */
int la_trie_hier_get_dev2(void* stream, const int32_t* d_tok, const double* d_fo, const double* d_fi, int64_t fi_stride,
                          const int32_t* d_cstart, const int32_t* d_ccount, int32_t n_records, const int32_t* d_queries /*[B][8]*/,
                          const int32_t* d_nq, const int32_t* d_plane, const int32_t* d_branch_length /*[B] or NULL*/, int B,
                          int decoding_length, int branch_length, int min_in, int min_out, int mode, const int32_t* d_stop,
                          int n_stop, int32_t* d_scratch_q, double* d_scratch_v, int32_t* d_out_ids, uint64_t* d_out_rowmask,
                          int32_t* d_out_n, int32_t* d_out_sizes, int32_t* d_out_nsizes);
/* The same retrieval with ONE WORKGROUP PER QUERY (csrc/la_trie_wg.hip; round 6): level-synchronous passes of 256 threads instead of an
 * ordered DFS by one wavefront, draft trees of up to LA_TREE_WIDE_MAX rows (the reference's best published setting is decoding_length = 128,
 * branch_length = 32: lookahead/README.md:100) with multi-word row masks.  Semantics: lookahead_cache.py:65-144 (Tree.get), :224-246
 * (_match), :146-154 (_dfs_get_freqs), :248-293 (_ravel), :408-439 (hier_get); bit-identical to la_cache_hier_get.
 *   image                tok / fo / fi (planes fi_stride records apart) / cstart / ccount of `n_records` records, as la_trie_hier_get_dev2;
 *                        root_of (optional) = token -> tree root table of la_trie_image.
 *   queries              rows of 8 int32 (first nq[b] used), per-query fi plane and branch length (NULL: plane 0 / branch_length).
 *   scratch              int32 [B][16][n_records], double [B][3][n_records].
 *   results              out_ids int32 [B][row_stride]; out_rowmask uint64 [B][row_stride][mask_words] (word w of a row = tree columns
 *                        64 w ..); row_stride >= decoding_length, mask_words >= ceil(decoding_length / 64), <= 4.  row_stride = 64,
 *                        mask_words = 1 is the layout of la_trie_hier_get_dev2 / la_llama_mstep_trie.  out_nsizes[b] = -1: the subtree is
 *                        deeper than 128 levels (the 1-row answer is returned). */
typedef struct la_trie_query {
    const int32_t* tok; const double* fo; const double* fi; int64_t fi_stride; const int32_t* cstart; const int32_t* ccount;
    int32_t n_records;
    const int32_t* root_of; int32_t n_root_of;
    const int32_t* queries; const int32_t* nq; const int32_t* plane; const int32_t* branch_lengths; int32_t B;
    int32_t decoding_length, branch_length, min_in, min_out, mode;
    const int32_t* stop; int32_t n_stop;
    int32_t* scratch_i; double* scratch_v;
    int32_t* out_ids; uint64_t* out_rowmask; int32_t row_stride, mask_words;
    int32_t* out_n; int32_t* out_sizes; int32_t* out_nsizes;
    int32_t lds_level_cap, lds_cand_cap, one_wave_cap;   /* 0 = the library's limits (4096 entries per level / 3072 candidates in LDS, 256
                                                            candidates ordered by one wave; one_wave_cap < 0: never).  Smaller values send
                                                            small sets down the global-scratch paths — for the parity tests. */
} la_trie_query;
int la_trie_hier_get_wg(void* stream, const la_trie_query* q);
/* one_get on the device mirror (LookaheadCache.one_get, lookahead_cache.py:490-517 with Tree.get_one_branch :171-222): one wavefront
 * per query, the single most frequent chain (<= branch_length tokens behind the root token) with lower-triangular row masks; the
 * same buffers and per-query plane / branch length as la_trie_hier_get_dev2 (d_out_sizes[b][0] = the chain length when
 * d_out_nsizes[b] == 1).  Bit-identical to la_cache_one_get. */
int la_trie_one_get_dev2(void* stream, const int32_t* d_tok, const double* d_fo, const double* d_fi, int64_t fi_stride,
                         const int32_t* d_cstart, const int32_t* d_ccount, int32_t n_records, const int32_t* d_queries /*[B][8]*/,
                         const int32_t* d_nq, const int32_t* d_plane, const int32_t* d_branch_length /*[B] or NULL*/, int B,
                         int decoding_length, int branch_length, int mode, const int32_t* d_stop, int n_stop, int32_t* d_out_ids,
                         uint64_t* d_out_rowmask, int32_t* d_out_n, int32_t* d_out_sizes, int32_t* d_out_nsizes);
/* Device-side retrieval.  la_cache_export snapshots the forest for one input slot `idx` into host arrays (call with
 * cap = 0 to get *n_nodes): live nodes renumbered breadth-first, children of node u = ids [cstart[u], cstart[u]+ccount[u])
 * in dict insertion order, node 0 = super-root over the per-token trees.  la_trie_hier_get_dev runs hier_get
 * (lookahead_cache.py:408-439, Tree.get :65-144) for B queries on that mirror, one wavefront per query (ballot prefix
 * match, wave-parallel live-subtree scan, radix-select cut-offs, ordered DFS); results are bit-identical to
 * la_cache_hier_get.  Queries are rows of 8 int32 (first nq[b] used).  Scratch: int32[B][n_nodes], double[B][2][n_nodes]. */
int la_cache_export(la_cache* c, int idx, int32_t cap, int32_t* tok, double* fo, double* fi, int32_t* cstart,
                    int32_t* ccount, int32_t* n_nodes);
int la_trie_hier_get_dev(void* stream, const int32_t* d_tok, const double* d_fo, const double* d_fi, const int32_t* d_cstart,
                         const int32_t* d_ccount, int n_nodes, const int32_t* d_queries, const int32_t* d_nq, int B,
                         int decoding_length, int branch_length, int min_input_size, int min_output_size, int mode,
                         const int32_t* d_stop, int n_stop, int32_t* d_scratch_q, double* d_scratch_v, int32_t* d_out_ids,
                         uint64_t* d_out_rowmask, int32_t* d_out_n, int32_t* d_out_sizes /*[B][2]*/, int32_t* d_out_nsizes);

/* ------------------------------------------------------------------------
 * 2. Step kernels (device).  Each is also reachable through la_llama_step;
 *    exported singly for unit parity tests.
 * --------------------------------------------------------------------- */

/* Device-resident per-sequence step state (int32 words). */
#define LA_ST_NKEYS      0   /* committed keys in the main KV cache (= context_length-1) */
#define LA_ST_T          1   /* valid tree tokens in this block (1..64)                   */
#define LA_ST_MODE       2   /* 0 = verify (accept scan), 1 = prefill chain (commit all), 2 = forward only */
#define LA_ST_NOUT       3   /* out: number of emitted tokens (= matches+1)               */
#define LA_ST_DSTBASE    4   /* out: first main-cache row written by the commit           */
#define LA_ST_NCOMMIT    5   /* out: rows committed                                       */
#define LA_ST_MAXKEYS    6   /* capacity of the main KV cache in keys (multiple of 32)    */
#define LA_ST_SEQ        7   /* out: steps published so far (zero-copy completion word)   */
#define LA_ST_OUTTOK     8   /* out: [64] emitted tokens, path order then bonus           */
#define LA_ST_SRCIDX    72   /* out: [64] tree rows committed, in order                   */
#define LA_ST_ARGMAX   136   /* out: [64] argmax token per tree row                       */
#define LA_ST_WORDS    200

/* Host->device step input block: {T, mode, nkeys hint, pad, ids[64] (int32), rowmask[64] (uint64)}. */
#define LA_IN_T          0
#define LA_IN_MODE       1
#define LA_IN_NKEYS_HINT 2   /* the caller's count of committed keys before this block (la_llama_step reads it ON THE HOST to pick the
                                tree-attention form: one launch while a head group's K/V fits its XCD's L2, key splits + combine beyond;
                                a hint only — both forms are exact at any context, 0 / stale values cost time, never correctness: a caller
                                that leaves the word 0 always runs the one-launch form, which is slower past ~2000 committed keys, so
                                fill it on long contexts.  Both forms are captured on the first call, not on the first crossing) */
#define LA_IN_IDS        4
#define LA_IN_ROWMASK   68   /* int32 word offset; 8-byte aligned */
#define LA_IN_WORDS    196

/* Tree-mask / position construction.  Replaces lookahead_prepare_inputs_for_generation's mask
 * concat (pretrained_model.py:725-734) + the model hook (models/llama/modeling_llama.py:584-588):
 * pos[t] = nkeys + popcount(rowmask[t]) - 1; pad rows (t >= T) get rowmask = 1<<t. */
int la_build_tree_inputs(void* stream, const int32_t* d_in, int32_t* d_state,
                         int32_t* d_pos /*[64]*/, uint64_t* d_rowmask /*[64]*/, int32_t* d_ids /*[64]*/);

/* Accept scan.  Replaces _lookahead_update_model_kwargs_for_generation
 * (pretrained_model.py:764-892) with an empty logits-processor list. */
int la_accept_scan(void* stream, const int32_t* d_ids, const uint64_t* d_rowmask, int32_t* d_state);

/* KV commit ("compaction").  Replaces _update_cache_with_axis_2 (pretrained_model.py:894-907):
 * rows state[SRCIDX][0..NCOMMIT) of the fresh (tree) K/V tiles become main-cache rows
 * DSTBASE.. ; all layers in one launch. */
int la_kv_commit(void* stream, const void* d_kfresh, const void* d_vfresh, void* d_kmain, void* d_vmain,
                 const int32_t* d_state, int n_layers, int n_kv_heads, int max_keys);

/* Weight repack into MFMA-fragment tile order (one-off at load).  W: [N][K] bf16 row-major
 * (torch nn.Linear layout), N%32==0, K%16==0.  interleave2=1 packs two [N][K] matrices
 * (gate, up) as alternating 32-row blocks. */
int la_pack_weight(void* stream, const void* d_w, const void* d_w2, int N, int K, int interleave2,
                   void* d_out);
/* Activations [64][K] bf16 row-major -> packed operand order (tests / debugging). */
int la_pack_x(void* stream, const void* d_x, int K, void* d_out);

/* Skinny GEMM  out[64][N] = x[64][K] . W[N][K]^T  (bf16 in, fp32 accumulate), split-K slabs.
 * Replaces the nn.Linear calls of LlamaAttention/LlamaMLP (modeling_llama.py:172-186, 222-224, 296). */
/* `rb`: low byte = 32-row blocks of W per workgroup (1 or 2); bits 8.. = pipeline variant
 * (0 = 4 waves x 8 tile-sets in flight [default], 1 = 8 waves x 4, 2 = 4 waves x 6, 3 = 8 waves x 8). */
int la_gemm64_slab(void* stream, const void* d_wp, const void* d_xp, int N, int K, int rb, int ksplit,
                   float* d_slabs /*[ksplit][64][N]*/);
int la_gemm64_swiglu(void* stream, const void* d_wp_gateup, const void* d_xp, int F, int K,
                     void* d_act_packed /*[64][F] packed*/, int variant);
/* QKV projection fused with RoPE and the Q / fresh-K / fresh-V fragment writes (replaces la_gemm64_slab +
 * la_qkv_post; models/llama/modeling_llama.py:222-232).  d_wp must be packed from [Wq;Wk;Wv] with its rows gathered
 * by la_qkv_row_perm (packed row r <- original row perm[r]) so that every workgroup owns RoPE pairs (d, d+64). */
int la_gemm64_qkv(void* stream, const void* d_wp, const void* d_xp, int n_heads, int n_kv_heads, int K,
                  const int32_t* d_pos, const void* d_rope_cos, const void* d_rope_sin,
                  void* d_qf, void* d_kfresh, void* d_vfresh, int variant);
int la_qkv_row_perm(int n_heads, int n_kv_heads, int32_t* perm /*[(nh+2*nkv)*128], host*/);
/* Heads narrower than 128 features (LlamaAttention is shape-generic, models/llama/modeling_llama.py:189-308): every kernel of this library
 * lays a head out as a 128-feature LANE (RoPE pairs (d, d + 64), 8192-element Q / K / V fragments, o_proj's K = n_heads * 128).  A model
 * with head_dim < 128 (even) runs in the same lanes with zero padding: lane_src[j] = the head's own feature that lane j carries, or -1 for
 * a padding lane — feature d < head_dim/2 sits in lane d, its rotary partner d + head_dim/2 in lane 64 + d.  The caller builds
 * [Wq;Wk;Wv] with 128 rows per head (row of lane j = the head's row lane_src[j], zero rows for -1) and o_proj with 128 columns per head
 * the same way BEFORE la_qkv_row_perm / la_rowplan / la_pack_*, and RoPE tables of 64 columns whose column d < head_dim/2 holds
 * cos / sin(pos * theta^(-2d / head_dim)) (the rest is never multiplied with a non-zero value).  Zero lanes add exact zeros to every dot
 * product (q.k, P.v, o_proj), so the results are those of the unpadded model; la_llama_config.head_dim (the real one) sets the softmax
 * scale 1 / sqrt(head_dim).  Cost: K/V rows, the attention kernels and QKV / o_proj run at the 128-lane width. */
int la_head_lane_map(int head_dim, int32_t* lane_src /*[128], host*/);
/* Balanced variants: exactly n_wg workgroups (one per CU: n_wg = CU count, 256 on MI355X), each owning
 * R = rows/n_wg rows per matrix as 32-row blocks with a partial last block.  The weight image is packed by
 * la_pack_planned following la_rowplan (out[i] = source row of packed row i, -1 = zero pad row;
 * kind 0 = single matrix (lm_head), 1 = gate/up pair (rows >= n_rows index the second matrix), 2 = qkv RoPE pairs).
 * la_rowplan returns the number of packed rows, or LA_E_RANGE if the shape cannot be balanced this way. */
int la_rowplan(int kind, int n_rows, int n_wg, int32_t* out /* may be NULL to query */);
/* Pack for the balanced kernels: workgroup-major, every block stored compactly as [k-tile][half][stored row][8] where
 * stored rows = valid rows rounded up to a multiple of 4 (whole 128-byte lines per tile); la_planned_elems gives the
 * bf16 element count of the image.  d_plan = the la_rowplan array copied to the device. */
int64_t la_planned_elems(int kind, int n_rows, int K, int n_wg);
int la_pack_planned(void* stream, const void* d_w, const void* d_w2, const int32_t* d_plan, int kind, int n_rows, int K,
                    int n_wg, void* d_out);
int la_gemm64r_swiglu(void* stream, const void* d_wp, const void* d_xp, int F, int K, int n_wg, void* d_act_packed);
int la_gemm64r_logits(void* stream, const void* d_wp, const void* d_xp, int V, int K, int n_wg,
                      void* d_logits_bf16, float* d_cand_val /*[n_wg][64]: one candidate per workgroup and token*/, int32_t* d_cand_idx);
int la_gemm64r_qkv(void* stream, const void* d_wp, const void* d_xp, int n_heads, int n_kv_heads, int K, int n_wg,
                   const int32_t* d_pos, const void* d_rope_cos, const void* d_rope_sin,
                   void* d_qf, void* d_kfresh, void* d_vfresh);
int la_gemm64_logits(void* stream, const void* d_wp, const void* d_xp, int V, int K, int rb,
                     void* d_logits_bf16 /*[64][V] or NULL*/, float* d_cand_val, int32_t* d_cand_idx);
int la_argmax_finalize(void* stream, const float* d_cand_val, const int32_t* d_cand_idx, int n_tiles,
                       int32_t* d_state);

/* Fused elementwise stages (RMSNorm: modeling_llama.py:76-90; RoPE: :93-169). */
int la_embed_norm(void* stream, const void* d_embed, const int32_t* d_ids, const void* d_norm_w,
                  int hidden, float eps, void* d_h, void* d_xp);
int la_resid_norm(void* stream, void* d_h, const float* d_slabs, int n_slabs, const void* d_norm_w,
                  int hidden, float eps, void* d_xp);
int la_qkv_post(void* stream, const float* d_slabs, int n_slabs, int n_heads, int n_kv_heads,
                const int32_t* d_pos, const void* d_rope_cos, const void* d_rope_sin,
                void* d_qf, void* d_kfresh, void* d_vfresh);
/* Tree attention: mask-free prefix from the packed main cache + 64 fresh keys under rowmask. */
int la_tree_attn(void* stream, const void* d_qf, const void* d_kmain, const void* d_vmain,
                 const void* d_kfresh, const void* d_vfresh, const uint64_t* d_rowmask,
                 const int32_t* d_state, int n_heads, int n_kv_heads, int max_keys, int n_split,
                 float* d_opart, float* d_mpart, float* d_lpart, void* d_attn_xp);

/* ------------------------------------------------------------------------
 * 3. Whole verify step for Llama-family models (captured as one hipGraph).
 *    Replaces LlamaForCausalLM.forward under the rank-4 mask hook
 *    (models/llama/modeling_llama.py:544-677, 710-794) + a16/a17 above.
 * --------------------------------------------------------------------- */
typedef struct la_llama la_llama;

typedef struct la_llama_config {
    int32_t n_layers, hidden, n_heads, n_kv_heads, head_dim, ffn, vocab;   /* head_dim: even, 8 .. 128; < 128: padded lanes, la_head_lane_map */
    int32_t max_keys;        /* KV capacity per sequence, multiple of 32, >= max_length + 64 */
    int32_t max_pos;         /* rows in the RoPE tables                                       */
    int32_t attn_split;      /* key-range splits per head (0 = auto)                          */
    float   rms_eps;
    int32_t gemm_cfg[8];     /* {qkv_rb,qkv_ks,o_rb,o_ks,down_rb,down_ks,lm_rb,gateup_variant}; 0 = auto */
    int32_t balanced_wg[3];  /* {qkv, gate/up, lm_head}: > 0 = weights were packed with la_rowplan for that many workgroups */
    int32_t n_slots;         /* sequence slots of the cursor-batch path (0/1 = single sequence); each slot owns a
                                max_keys region of the main KV cache */
    int32_t n_experts;       /* > 0: Mixtral-style sparse MoE MLP (mixtral/modeling_mixtral.py:692-759), <= LA_MOE_MAX_E */
    int32_t top_k;           /* experts per token (Mixtral: 2) */
    int32_t fuse;            /* in-kernel norm->GEMM fusion, opt-in (0 / -1 = off): bit 0 = post-attention norm into the
                                gate/up launch, bit 1 = input norm of layers > 0 into the QKV launch; bits 2 / 3 = gate/up + down_proj
                                as one role-fused launch (bits 0 + 2 together: norm -> gate/up -> down_proj in ONE launch, with bit 1
                                four launches per layer); bit 4 (16) = the fused producers publish WRITE-THROUGH (sc1 stores + drained
                                flag, no release fence).  Bitwise identical results, all measured slower than the separate kernels
                                 */
    int32_t sliding_window;  /* > 0: sliding-window attention over the committed keys (Mistral: 4096; visible iff
                                pos_row - pos_key <= window, the transformers mask rule).  An EXTENSION: the reference's
                                lookahead path feeds the full mask (mistral/modeling_mistral.py:979-983) */
    int32_t kv_ring;         /* 1 (needs sliding_window > 0): a sequence's main KV cache is a RING of max_keys rows — position p lives
                                in row p mod max_keys — so memory is O(window) and generation is bounded by max_pos, not by max_keys;
                                max_keys >= sliding_window + 64 * max(max_blocks, 1) + 32 */
    int32_t max_blocks;      /* > 1: allocate the multi-block step (la_llama_mstep) for up to this many 64-row blocks (<= LA_MB_MAX) */
    int32_t norm_cast_first; /* RMSNorm flavour: 0 = LlamaRMSNorm (llama/modeling_llama.py:86-90, one rounding), 1 = Mistral/
                                MixtralRMSNorm (mixtral/modeling_mixtral.py:160-165, normalised value rounded first) */
    int32_t qkv_mb_wg;       /* > 0: the layers carry a SECOND QKV image (la_llama_layer_weights.wqkv_mb) packed with la_rowplan for
                                this many workgroups, read by the multi-block step only.  The 64-row step wants one workgroup per CU
                                (HBM streaming); at 192-512 rows the launch is MFMA-bound and a GQA model's 24 rows per workgroup
                                (12 RoPE pairs in a 32-row MFMA block) waste 5/8 of the matrix-core work: fewer, fuller workgroups
                                (Mistral: 96 x 32 pairs) x token quarters instead */
} la_llama_config;

typedef struct la_llama_layer_weights {   /* device pointers, packed by la_pack_weight */
    const void* wqkv;        /* [(nh+2*nkv)*128][hidden] (128-feature lanes, la_head_lane_map), rows gathered by la_qkv_row_perm unless gemm_cfg[1] < 0 */
    const void* wo;          /* [hidden][nh*128]               */
    const void* wgateup;     /* interleaved gate/up [2*ffn][hidden] */
    const void* wdown;       /* [hidden][ffn]                  */
    const void* norm1;       /* input_layernorm weight bf16 [hidden]          */
    const void* norm2;       /* post_attention_layernorm weight bf16 [hidden] */
    /* n_experts > 0 (wgateup / wdown above are then unused): */
    const void* router;              /* block_sparse_moe.gate weight, bf16 [n_experts][hidden] row-major        */
    const void* const* ex_gateup;    /* host array [n_experts]: packed interleaved w1/w3 of each expert; when the images of a
                                        layer are equally spaced in memory (ptr[e] = ptr[0] + e * stride, stride % 16 == 0) all
                                        experts of a stage run in ONE launch, otherwise one launch per expert             */
    const void* const* ex_down;      /* host array [n_experts]: packed w2 of each expert                        */
    const void* wqkv_mb;             /* cfg.qkv_mb_wg > 0: the QKV rows once more, planned for cfg.qkv_mb_wg workgroups (multi-block step) */
} la_llama_layer_weights;

typedef struct la_llama_weights {
    const void* embed;       /* [vocab][hidden] bf16 row-major (gather source) */
    const void* lm_head;     /* packed [vocab][hidden]                         */
    const void* final_norm;  /* bf16 [hidden]                                  */
    const void* rope_cos;    /* bf16 [max_pos][64]: column d < head_dim/2 = cos(pos * theta^(-2d/head_dim)) */
    const void* rope_sin;    /* bf16 [max_pos][64]                             */
    const la_llama_layer_weights* layers;   /* host array [n_layers]           */
} la_llama_weights;

/* Bytes of device memory the caller must provide for KV caches + scratch. */
int64_t   la_llama_workspace_bytes(const la_llama_config* cfg);
la_llama* la_llama_create(const la_llama_config* cfg, const la_llama_weights* w,
                          void* d_workspace, int64_t workspace_bytes);
void      la_llama_destroy(la_llama* m);
/* Reset the sequence (nkeys = 0). */
int la_llama_reset(la_llama* m, void* stream);
/* One block.  host_in (LA_IN_WORDS int32) and host_out (LA_ST_OUTTOK+64 int32) MUST be pinned host memory
 * (hipHostMalloc / hipHostRegister; torch pin_memory()): the first kernel reads host_in and the last kernel writes
 * host_out directly (zero-copy), there are no copy commands around the graph.  Call la_llama_wait before reading host_out
 * (it polls the completion word LA_ST_SEQ; hipStreamSynchronize alone is sufficient too).  Passing other addresses than
 * in the previous call re-captures the graph.
 * The captured graph: build inputs -> embed -> L x {qkv (+RoPE, fresh K/V), tree attention, o, norm, gate/up, down, norm}
 * -> lm_head + argmax -> accept scan -> kv commit -> publish (first LA_ST_OUTTOK+64 state words into host_out).
 * Asynchronous on `stream`. */
int la_llama_step(la_llama* m, void* stream, const int32_t* host_in, int32_t* host_out);
/* Wait for the step launched last by la_llama_step (spins on host_out[LA_ST_SEQ], falls back to a stream sync). */
int la_llama_wait(la_llama* m, void* stream);
/* Sequential accept path (non-empty logits-processor list / sampling, pretrained_model.py:825-875): run a step with
 * host_in[LA_IN_MODE] = 2 (forward only: no accept walk, nothing committed), read the logits rows the walk needs
 * (la_llama_buffer(m, 0)), then commit the accepted tree rows rows[0..n) (rows[0] = 0, the root).  Synchronous. */
int la_llama_commit(la_llama* m, void* stream, const int32_t* rows, int n, int32_t* host_out);
/* Same work launched kernel-by-kernel (no graph), with copy commands around it (host blocks need not be pinned; wait with
 * hipStreamSynchronize): for profiling and as a cross-check. */
int la_llama_step_eager(la_llama* m, void* stream, const int32_t* host_in, int32_t* host_out);
/* Device addresses of internal buffers for parity tests: 0 logits bf16 [64][vocab], 1 state,
 * 2 hidden h bf16 [64][hidden], 3 final normed x (packed), 4/5 main K/V cache, 6/7 fresh K/V tiles,
 * 8 batch state block (LA_BST_*), 9 routing weights fp32 [n_layers][64][LA_MOE_MAX_E] of the last block, 10 accumulated
 * expert output bf16 [64][hidden] of the last MoE layer. */
void* la_llama_buffer(la_llama* m, int which);
/* Kernel-class timing of one block with HIP events recorded on `stream` between eager launches
 * (bench.py's roofline block).  out_ms[0..6] = per-step time summed over the launches of
 * {qkv GEMM, o GEMM, gate/up GEMM, down GEMM, lm_head GEMM, tree attention(+combine), everything else},
 * out_ms[7] = the whole step; out_launches[0..6] = intervals of that class per step.  Mean over `iters`
 * steps; the sequence state is saved and restored, so the context does not advance. */
int la_llama_profile(la_llama* m, void* stream, const int32_t* host_in, int iters,
                     float* out_ms /*[8]*/, int32_t* out_launches /*[7] or NULL*/);
/* Mean duration (ms) of one gate/up launch — the dominant kernel of the step — measured without event packets between launches:
 * every layer's launch back to back inside one pair of HIP events per pass (kernel + the dependent-launch boundary).  bench.py's
 * `roofline.achieved`; the rocprofv3 kernel average under profiles/ is the cross-check. */
int la_llama_profile_gateup(la_llama* m, void* stream, int iters, float* out_ms);

/* ---- native decode loop: the per-step host work of lookahead_generation (pretrained_model.py:1172-1240) without the
 * interpreter: hier_get(last max_query_length tokens) -> la_llama_step -> stream_put, until max_length / eos / max_steps.
 * Greedy, empty logits-processor list, decoding_mode hier.  The prompt must already be prefilled (seq[0..seq_len) holds
 * prompt + first generated token, its trie `put` done).  Appends tokens to seq (capacity >= max_length + 16), fills
 * dls/edls (capacity >= max_steps) and returns the number of steps in *n_steps; *finished = 1 when a stop condition
 * other than max_steps ended the loop.  host_in / host_out: pinned staging blocks (LA_IN_WORDS / LA_ST_OUTTOK+64). */
typedef struct la_decode_params {
    int32_t decoding_length, branch_length, max_query_length;
    int32_t mode;                 /* LA_MODE_* of the retrieval (hier_mix = LA_MODE_MIX) */
    int32_t idx;                  /* trie request slot of this sequence */
    int32_t max_length;           /* stop when seq_len >= max_length */
    int32_t max_steps;
    int32_t n_eos;
    int32_t eos[8];
} la_decode_params;
int la_lookahead_decode(la_llama* m, la_cache* c, void* stream, const la_decode_params* p, int32_t* seq, int32_t* seq_len,
                        int32_t* host_in, int32_t* host_out, int32_t* dls, int32_t* edls, int32_t* n_steps,
                        int32_t* finished, double* fts /*[max_steps] seconds per step or NULL*/,
                        double* qts /*[max_steps] seconds of the trie query or NULL*/);

/* ---- mixture of experts (Mixtral) row stages; the expert GEMMs are the la_gemm64* kernels launched once per expert
 * with an early exit when no row routes to that expert ------------------------------------------------------- */
/* post-attention residual + RMSNorm with the router fused: route_w[64][LA_MOE_MAX_E] fp32 (bf16-valued, 0 = not routed);
 * rows >= *d_n_rows get no expert. */
int la_resid_norm_router(void* stream, void* d_h, const float* d_slabs, int n_slabs, const void* d_norm_w, int hidden,
                         float eps, void* d_xp, const void* d_router_w, int n_experts, int top_k, float* d_route_w,
                         const int32_t* d_n_rows);
/* acc[t] (+)= bf16(bf16(sum slabs[t]) * route_w[t][expert]) for routed rows; first != 0 zero-fills first. */
int la_moe_accum(void* stream, const float* d_slabs, int n_slabs, const float* d_route_w, int expert, int hidden,
                 void* d_acc, int first);
/* h = bf16(h + addend); xp = RMSNorm(h) packed. */
int la_resid_norm_addend(void* stream, void* d_h, const void* d_addend, const void* d_norm_w, int hidden, float eps,
                         void* d_xp);

/* ---- cursor batch (bs>1): several sequences share the 64 rows of one verify block --------------------------------
 * Replaces the batch twin of the loop: bat_get's padded drafts + [bs,T,W] masks (lookahead_cache.py:519-561,
 * pretrained_model_batch.py:706-731), the batched forward writing K/V at each sample's cursor
 * (modeling_llama_batch.py:340-420), the per-sample accept walk (pretrained_model_batch.py:814-886) and the in-place
 * KV row moves (:893-904, 986-989).  Rows are NOT padded: each active sequence ("slot") contributes its T_b tree rows,
 * sum(T_b) <= 64 — exactly the budget the reference batch driver allows (decoding_length // bs per sample) —
 * so the weights are still streamed once per step for the whole batch.  Ancestor bits of a row mask are BLOCK row
 * indices (the host shifts each sample's local mask by its first row). */
#define LA_MAX_SEQ      16
/* batch step input block (int32 words) */
#define LA_BIN_T          0   /* block rows in use (<= 64)                                   */
#define LA_BIN_IDS        4   /* [64] token ids                                              */
#define LA_BIN_ROWMASK   68   /* uint64[64] ancestor masks over block rows                   */
#define LA_BIN_SEQ      196   /* [64] slot of each row, -1 = row unused                      */
#define LA_BIN_MODE     260   /* [16] per slot: 0 = verify tree, 1 = prefill chain, 2 = forward only (la_llama_bcommit) */
#define LA_BIN_LIMIT    276   /* [16] per slot: max tokens to emit (max_length - cursor - 1) */
#define LA_BIN_WORDS    292
/* batch device state block (int32 words) */
#define LA_BST_NKEYS      0   /* [16] committed keys per slot (= the sample's cursor)         */
#define LA_BST_NOUT      16   /* out: [16] tokens emitted by the last step (0 = slot idle)    */
#define LA_BST_OUTTOK    32   /* out: [16][16] emitted tokens per slot                        */
#define LA_BST_DST      288   /* out: [64] main-cache key row each block row was committed to, -1 = dropped */
#define LA_BST_ARGMAX   352   /* out: [64] argmax token per block row                         */
#define LA_BST_SEQ      416   /* [64] row -> slot map of the last step                        */
#define LA_BST_WORDS    480
/* step-control kernels of the batch path (also used inside la_llama_bstep) */
int la_build_batch_inputs(void* stream, const int32_t* d_in, int32_t* d_bstate, int32_t* d_pos, uint64_t* d_rowmask,
                          int32_t* d_ids);
int la_accept_scan_batch(void* stream, const int32_t* d_in, const int32_t* d_ids, const uint64_t* d_rowmask,
                         int32_t* d_bstate, int n_slots, int slot_keys);
int la_kv_commit_batch(void* stream, const void* d_kfresh, const void* d_vfresh, void* d_kmain, void* d_vmain,
                       const int32_t* d_bstate, int n_layers, int n_kv_heads, int total_keys);
int la_tree_attn_batch(void* stream, const void* d_qf, const void* d_kmain, const void* d_vmain, const void* d_kfresh,
                       const void* d_vfresh, const uint64_t* d_rowmask, const int32_t* d_bstate, int n_heads,
                       int n_kv_heads, int slot_keys, int n_slots, int n_split, float* d_opart, float* d_mpart,
                       float* d_lpart, void* d_attn_xp);
/* ------------------------------------------------------------------------
 * Multi-GPU: the accepted-token all-gather (the only exchange on the path).  One process per GPU; every rank owns
 * B_loc sequences, a model replica and a trie replica.  Per step: d_local int32[B_loc][words] = {n, tokens...} per sequence ->
 * d_global int32[world][B_loc][words] by ONE ncclAllGather on `stream` (RCCL over xGMI); the host then applies stream_put for
 * every sequence in global batch-index order (pretrained_model_batch.py:1254-1259), so all trie replicas equal the reference's
 * single-process batch run.  The reference has no counterpart (single process, device_map only): its contract is the ORDER.
 * librccl is resolved at run time; la_comm_unique_id is called on rank 0 and its 128 bytes are distributed by the launcher
 * (bench.py / distributed.py broadcast them through torch.distributed).
 * --------------------------------------------------------------------- */
typedef struct la_comm la_comm;
int      la_comm_unique_id(uint8_t* out128);
la_comm* la_comm_create(const uint8_t* id128, int world, int rank);     /* binds to the current HIP device */
int      la_comm_destroy(la_comm* c);
int      la_gather_accepted(la_comm* c, void* stream, const int32_t* d_local, int b_loc, int words, int32_t* d_global);

/* ------------------------------------------------------------------------
 * Multi-block step: nblk <= LA_MB_MAX blocks of 64 rows in ONE pass over the weights (M = nblk*64 rows through the LDS-
 * staged GEMM family of csrc/la_mblock.hip).  A block is one sequence's 64-token draft tree (BASELINE configs 3-5: a full
 * tree per sample) or one 64-token piece of a prompt (prefill: consecutive blocks of the same slot form a
 * causal chain; every block but the last of a chain must hold 64 rows).  Reference: the batched forward
 * models/llama/modeling_llama_batch.py:340-420 and pretrained_model_batch.py:706-931, per sample.
 * Needs cfg.max_blocks >= nblk and cfg.n_slots >= 1; slots are those of the cursor-batch path (LA_BST_NKEYS).
 * --------------------------------------------------------------------- */
#define LA_MB_MAX          8
#define LA_MIN_NBLK        0
#define LA_MIN_BLK         4    /* [8][4] per block: slot, T (1..64), mode (0 verify tree, 1 prefill chain, 2 forward only,
                                   3 = next 64 rows of the WIDE tree the preceding block(s) of the same slot started), limit */
#define LA_MIN_IDS        36    /* [8][64] token ids                                                                   */
#define LA_MIN_ROWMASK   548    /* uint64[8][64] ancestor masks over the block's own rows (8-byte aligned offset)      */
#define LA_MIN_XMASK    1572    /* uint64[8][64][3] mode-3 blocks: ancestor masks over the rows of the 1st / 2nd / 3rd block of the tree */
#define LA_MIN_WORDS    4644
/* Wide trees (round 3): the reference grid-searches decoding_length x branch_length freely and publishes its best numbers at
 * decoding_length = 128, branch_length = 32 (lookahead/README.md:100, benchmarks/benchmark.py:256-288).  A tree of up to
 * LA_TREE_WIDE_MAX rows is ceil(T / 64) consecutive blocks of one slot: block 0 in mode 0, the others in mode 3; row r of the
 * tree lives in block r / 64; a row's ancestor mask is its LA_MIN_ROWMASK word (own block) plus the LA_MIN_XMASK words of the
 * earlier blocks.  Positions = cursor + popcount(all words) - 1; attention sees the earlier blocks' fresh keys under those
 * words; ONE wavefront walks the whole tree (4 rows per lane) and emits up to LA_MOUT_TOKS tokens into the first block's
 * LA_MOUT_OUTTOK record; LA_MOUT_DST carries the kept rows of every block. */
#define LA_TREE_WIDE_MAX 256
#define LA_MODE_TREE_PIECE 3
#define LA_MOUT_TOKS      40    /* tokens a block (or a wide tree through its first block) can emit per step */
#define LA_MOUT_NOUT       0    /* out: [8] tokens emitted per block                                                    */
#define LA_MOUT_NKEYS      8    /* out: [16] committed keys per slot after the step                                     */
#define LA_MOUT_OUTTOK    24    /* out: [8][LA_MOUT_TOKS] emitted tokens per block                                      */
#define LA_MOUT_T        344    /* out: [8] valid rows of each block (the draft length when the drafts came from the device trie) */
#define LA_MOUT_DST      352    /* out: [8][64] main-cache key row each block row was committed to, -1 = dropped        */
#define LA_MOUT_ARGMAX   864    /* out: [8][64] argmax token per block row                                              */
#define LA_MOUT_WORDS   1376
/* h2d of host_in (LA_MIN_WORDS), captured graph (one per nblk), d2h of the first LA_MOUT_DST words into host_out. */
int la_llama_mstep(la_llama* m, void* stream, const int32_t* host_in, int32_t* host_out);
int la_llama_mstep_eager(la_llama* m, void* stream, const int32_t* host_in, int32_t* host_out);
/* The same step with the drafts taken ON THE DEVICE from the outputs of la_trie_hier_get_dev2 (one query per block, block b =
 * query b): d_ids int32[nblk][64], d_rowmask uint64[nblk][64], d_n int32[nblk] stay where the trie kernel wrote them — no D2H
 * of drafts, no host packing, no H2D of the step input; slots / limits / last_tok: host arrays [nblk] (slot of each block, its
 * emit limit, and the token the block falls back to as a 1-row tree when the query returned nothing).  Everything is queued on
 * `stream` behind the trie kernels; host_out receives the first LA_MOUT_DST words (LA_MOUT_T = the draft lengths).  This is the
 * trie walk chained in front of the verify forward (lookahead_prepare_inputs_for_generation, pretrained_model_batch.py:706-743,
 * without its host round trip). */
int la_llama_mstep_trie(la_llama* m, void* stream, int nblk, const int32_t* slots, const int32_t* limits, const int32_t* last_tok,
                        const int32_t* d_ids, const uint64_t* d_rowmask, const int32_t* d_n, int32_t* host_out);
/* One GEMM of the multi-block family (unit parity): kind 0 = split-K slabs [ksplit][slab_rows][N] over a la_pack_weight image,
 * 1 = gate/up + SwiGLU -> act_xp [blk][64 x N packed], 2 = QKV + RoPE -> qf [blk][nh][8192], kfresh / vfresh [blk][nkv][8192],
 * 3 = lm_head -> logits bf16 [nblk*64][N] + argmax candidates [blk][4*workgroups][64].  n_wg > 0: image packed by
 * la_pack_planned for that many workgroups; 0: classic images (la_pack_weight; interleaved gate/up; la_qkv_row_perm rows).
 * x: nblk consecutive 64-row la_pack_x images.  Replaces nn.Linear of the batched forward (modeling_llama_batch.py:340-420). */
int la_mb_gemm(void* stream, int kind, const void* d_wp, const void* d_xp, int N, int K, int nblk, int n_wg, int ksplit,
               float* d_slabs, int slab_rows, void* d_act_xp, void* d_logits, float* d_cand_val, int32_t* d_cand_idx,
               const int32_t* d_pos, const void* d_rope_cos, const void* d_rope_sin, void* d_qf, void* d_kfresh, void* d_vfresh,
               int n_heads, int n_kv_heads);
/* Set a slot's committed-key cursor (slot 0 = also the single-sequence cursor LA_ST_NKEYS); synchronises the stream. */
int la_llama_set_nkeys(la_llama* m, void* stream, int slot, int nkeys);

/* Whole batch step (needs cfg.n_slots >= 1): h2d of host_in (LA_BIN_WORDS), captured graph, d2h of
 * LA_BST_DST words (NKEYS, NOUT, OUTTOK) into host_out.  Same asynchrony rules as la_llama_step. */
int la_llama_bstep(la_llama* m, void* stream, const int32_t* host_in, int32_t* host_out);
int la_llama_bstep_eager(la_llama* m, void* stream, const int32_t* host_in, int32_t* host_out);
/* Sequential accept path of the batch twin (pretrained_model_batch.py:814-931 with a non-empty logits-processor list or
 * sampling: the processors see the tokens accepted so far, so the walk runs on the host).  Run la_llama_bstep / la_llama_mstep
 * with mode 2 for the slots / blocks concerned (forward only: logits rows in la_llama_buffer(m, 0) / (m, 11), nothing emitted,
 * nothing committed, cursors unchanged), walk each sample's tree over its logits rows, then hand the commit plan back:
 * keep[r] (bcommit: r = block row, 64 entries; mcommit: r = 64 * block + row, nblk * 64 entries) = k >= 0 if row r is the k-th
 * kept key of its sequence in this step (k = 0: the root; the kept positions of a sequence must be 0..n-1), -1 if dropped.
 * The rows move to main-cache rows cursor + k (the in-place moves of :893-904, _update_cache :982-985) and the cursors advance.
 * Synchronous; host_out receives the header words (LA_BST_DST / LA_MOUT_DST of them) with the new cursors. */
int la_llama_bcommit(la_llama* m, void* stream, const int32_t* keep, int32_t* host_out);
int la_llama_mcommit(la_llama* m, void* stream, int nblk, const int32_t* keep, int32_t* host_out);
/* Forget a slot's sequence (committed keys = 0); slot < 0 resets every slot. */
int la_llama_reset_slot(la_llama* m, void* stream, int slot);

#ifdef __cplusplus
}
#endif
#endif /* LOOKAHEAD_HIP_H */
