// la_comm.cpp — the only exchange on the path: the per-step all-gather of accepted tokens over RCCL/xGMI.
//
// Sequences are independent: every GPU owns B_loc of them, a full model replica and a trie replica.  Per verify
// step each rank contributes int32[B_loc][W] = {n, tokens...}; ONE ncclAllGather on the caller's stream hands every rank every
// sequence's accepted tokens, which the host then applies to its trie in global batch-index order — the order in which the
// reference's single-process batch loop calls stream_put (common/pretrained_model_batch.py:1254-1259).  64 B per sequence:
// latency-bound; ring-vs-tree topology is irrelevant.
//
// librccl is resolved at run time (dlopen of the copy the process already loaded — PyTorch-ROCm ships its own — else the
// system one), so liblookahead_hip.so keeps loading on boxes without RCCL and never pulls a second RCCL into a torch process.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include "../../include/lookahead_hip.h"

extern void la_set_error(const std::string& s);

namespace {
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*fn_get_unique_id)(UniqueId*);
typedef int (*fn_comm_init_rank)(Comm*, int, UniqueId, int);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, Comm, hipStream_t);
typedef int (*fn_comm_destroy)(Comm);
typedef const char* (*fn_get_error_string)(int);

struct Rccl {
    void* h = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_get_error_string err = nullptr;
    bool tried = false;
};
Rccl g_rccl;

bool load_rccl() {
    if (g_rccl.tried) return g_rccl.h != nullptr;
    g_rccl.tried = true;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) if (!g_rccl.h) g_rccl.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);       // the copy already in the process
    for (const char* n : names) if (!g_rccl.h) g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!g_rccl.h) g_rccl.h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!g_rccl.h) { la_set_error("la_comm: librccl.so not found (needed only for multi-GPU runs)"); return false; }
    g_rccl.get_unique_id = (fn_get_unique_id)dlsym(g_rccl.h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(g_rccl.h, "ncclCommInitRank");
    g_rccl.all_gather = (fn_all_gather)dlsym(g_rccl.h, "ncclAllGather");
    g_rccl.comm_destroy = (fn_comm_destroy)dlsym(g_rccl.h, "ncclCommDestroy");
    g_rccl.err = (fn_get_error_string)dlsym(g_rccl.h, "ncclGetErrorString");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_gather || !g_rccl.comm_destroy) {
        la_set_error("la_comm: librccl.so lacks ncclGetUniqueId/ncclCommInitRank/ncclAllGather/ncclCommDestroy");
        g_rccl.h = nullptr;
        return false;
    }
    return true;
}
int fail(const char* what, int rc) {
    la_set_error(std::string(what) + ": " + (g_rccl.err ? g_rccl.err(rc) : "rccl error") + " (" + std::to_string(rc) + ")");
    return LA_E_HIP;
}
}  // namespace

struct la_comm { Comm comm; int world, rank; };

extern "C" {

int la_comm_unique_id(uint8_t* out128) {
    if (!out128) return LA_E_ARG;
    if (!load_rccl()) return LA_E_HIP;
    UniqueId id;
    memset(&id, 0, sizeof(id));
    int rc = g_rccl.get_unique_id(&id);
    if (rc != 0) return fail("ncclGetUniqueId", rc);
    memcpy(out128, id.internal, 128);
    return LA_OK;
}

la_comm* la_comm_create(const uint8_t* id128, int world, int rank) {
    if (!id128 || world < 1 || rank < 0 || rank >= world) { la_set_error("la_comm_create: bad arguments"); return nullptr; }
    if (!load_rccl()) return nullptr;
    UniqueId id;
    memcpy(id.internal, id128, 128);
    Comm c = nullptr;
    int rc = g_rccl.comm_init_rank(&c, world, id, rank);       // binds the communicator to the CURRENT HIP device
    if (rc != 0) { fail("ncclCommInitRank", rc); return nullptr; }
    return new la_comm{c, world, rank};
}

int la_comm_destroy(la_comm* c) {
    if (!c) return LA_E_ARG;
    if (c->comm && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(c->comm);
    delete c;
    return LA_OK;
}

int la_gather_accepted(la_comm* c, void* stream, const int32_t* d_local, int b_loc, int words, int32_t* d_global) {
    if (!c || !d_local || !d_global || b_loc < 1 || words < 1) return LA_E_ARG;
    const int kNcclInt32 = 2;
    int rc = g_rccl.all_gather(d_local, d_global, (size_t)b_loc * words, kNcclInt32, c->comm, (hipStream_t)stream);
    if (rc != 0) return fail("ncclAllGather", rc);
    return LA_OK;
}

}  // extern "C"
