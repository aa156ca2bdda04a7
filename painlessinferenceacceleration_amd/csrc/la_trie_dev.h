// la_trie_dev.h — argument blocks shared by the device-side retrieval kernels (la_trie_dev.hip: one wavefront per query;
// la_trie_wg.hip: one workgroup per query, round 6).  Internal; the public surface is include/lookahead_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct TrieDev {
    const int* tok; const double* fo; const double* fi; const int* cstart; const int* ccount; int n_nodes;
};
struct TrieQueryArgs {
    TrieDev t;
    const int* queries;   // [B][8]
    const int* nq;        // [B]
    int decoding_length, branch_length, min_in, min_out, mode;
    const int* stop; int n_stop;
    int* scratch_q;       // [B][n_nodes]
    double* scratch_v;    // [B][2][n_nodes]
    int* out_ids;         // [B][64]
    unsigned long long* out_rowmask;   // [B][64]
    int* out_n;           // [B]
    int* out_sizes;       // [B][2]
    int* out_nsizes;      // [B]
    const int* plane;     // [B] fi plane of each query (null: plane 0)
    long fi_stride;       // records between fi planes
    const int* bl;        // [B] per-query branch length (null: branch_length)
    long long* dbg;       // measurement aid (la_debug_set_ptr(0, .)): [B][8] wall_clock64 stamps {start, matched, scanned, cut-offs, emitted} + {rows, n_out}
};
// one workgroup per query (la_trie_wg.hip): rows of `row_stride` ids / `row_stride` x `mask_words` mask words per query
struct TrieWgArgs {
    TrieQueryArgs q;
    const int* root_of; int n_root_of;   // token -> record of its tree root (null: the root block is searched)
    int* scr_i;                          // [B][16][n_nodes]
    double* scr_v;                       // [B][3][n_nodes]
    int row_stride, mask_words;
    int lcap, mcap, onewave;             // set-size limits of the LDS paths (0 = the library's; onewave < 0: never)
};
int lk_trie_hier_get_wg(hipStream_t st, const TrieWgArgs& a, int B);
