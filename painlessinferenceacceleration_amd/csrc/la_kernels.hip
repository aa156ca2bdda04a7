// la_kernels.hip — hand-written gfx950 (CDNA4) kernels of the LOOKAHEAD verify step.
// wave64, v_mfma_f32_32x32x16_bf16, weights streamed HBM -> VGPR in MFMA fragment order.
// Reference semantics: lookahead/lookahead/models/llama/modeling_llama.py (cited per kernel).
#include "la_common.h"
#include <type_traits>
#include "la_kernels.h"
#include "la_knobs.h"

// ---------------------------------------------------------------------------------------------
// Layout converters (one-off at load time / tests)
// ---------------------------------------------------------------------------------------------
__global__ void k_pack_weight(const bf16_t* __restrict__ w, const bf16_t* __restrict__ w2, int N, int K,
                              int interleave2, bf16_t* __restrict__ out) {
    // one thread per (tile, lane): 16 B in, 16 B out (coalesced on the write side)
    const int K16 = K >> 4;
    const int NB = interleave2 ? (2 * N) >> 5 : N >> 5;
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)NB * K16 * 64;
    if (gid >= total) return;
    int lane = (int)(gid & 63);
    size_t tile = gid >> 6;
    int kb = (int)(tile % K16);
    int nb = (int)(tile / K16);
    const bf16_t* src = w;
    int row_blk = nb;
    if (interleave2) { src = (nb & 1) ? w2 : w; row_blk = nb >> 1; }
    int n = row_blk * 32 + (lane & 31);
    int k = kb * 16 + (lane >> 5) * 8;
    bf16x8 v = *(const bf16x8*)(src + (size_t)n * K + k);
    *(bf16x8*)(out + gid * 8) = v;
}

// Balanced ("planned") packing: workgroup-major, each virtual row-block stored compactly as
// [k-tile][half][valid row][8 elems] (tile = nv*32 bytes), so a partial block is one contiguous run per k-tile.
// One thread per 16-byte chunk of the output.
struct PackPlanArgs { int n_wg, RB, K16, n_rows; int nv[4]; int boff[4]; int wg_chunks; };   // nv = stored rows (multiple of 4)
__global__ void k_pack_planned(const bf16_t* __restrict__ w, const bf16_t* __restrict__ w2, const int* __restrict__ plan,
                               PackPlanArgs pa, bf16_t* __restrict__ out) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)pa.n_wg * pa.wg_chunks) return;
    const int wgi = (int)(gid / pa.wg_chunks);
    int c = (int)(gid % pa.wg_chunks);
    int rb = 0;
    while (rb + 1 < pa.RB && c >= pa.boff[rb + 1]) ++rb;
    c -= pa.boff[rb];
    const int nv = pa.nv[rb];
    const int kb = c / (2 * nv), rem = c % (2 * nv), h = rem / nv, r = rem % nv;
    int src = plan[(wgi * pa.RB + rb) * 32 + r];
    const bf16_t* base = w;
    if (src >= pa.n_rows) { base = w2; src -= pa.n_rows; }
    const int K = pa.K16 * 16;
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (src >= 0) v = *(const bf16x8*)(base + (size_t)src * K + kb * 16 + h * 8);      // -1 = alignment pad row
    *(bf16x8*)(out + gid * 8) = v;
}

__global__ void k_pack_x(const bf16_t* __restrict__ x, int K, bf16_t* __restrict__ out) {
    int gid = blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (t, k8)
    int K8 = K >> 3;
    if (gid >= LA_TB * K8) return;
    int t = gid / K8, k = (gid % K8) * 8;
    *(bf16x8*)(out + xp_offset(t, k)) = *(const bf16x8*)(x + (size_t)t * K + k);
}

// ---------------------------------------------------------------------------------------------
// Row kernels: embedding gather + RMSNorm, residual add + RMSNorm  (LlamaRMSNorm, :76-90;
// LlamaDecoderLayer residual adds, :352-363).  One workgroup per token row.
// ---------------------------------------------------------------------------------------------
template <int NWAVES>
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < NWAVES; ++i) tot += sh[i];
    return tot;
}

// h_new = bf16(h + bf16(sum slabs)); x = bf16(w * (h_new * rsqrt(mean(h_new^2)+eps)))  (fp32 math).
// 512 threads, <= 2 chunks of 8 elements per thread (hidden <= 8192).  NS is a template parameter so that every
// load of the row (h, NS slabs, norm weight) is issued before the first use: one memory round trip, not NS+2.
// WT = true (fused producers of a norm->GEMM launch, cfg.fuse bit 4): the packed operand leaves as WRITE-THROUGH 16-byte stores (sc1:
// straight to memory, no dirty line stays in this XCD's L2), so the hand-over needs no release fence (buffer_wbl2) — the cheap
// publish form of MI355X_MICROARCH.md (publish-large: 3.0 vs 8.2 us), cdna_hip_programming.md Guideline 16 R1.
template <int NS, bool MOE, bool WT = false>
__device__ __forceinline__ void row_norm_body(const int t, float* sh, float (*shr)[LA_MOE_MAX_E],
                                              const bf16_t* __restrict__ embed_row,
                                              bf16_t* __restrict__ h, const float* __restrict__ slabs,
                                              const bf16_t* __restrict__ nw, int hidden, float eps,
                                              bf16_t* __restrict__ xp, const bf16_t* __restrict__ addend,
                                              const bf16_t* __restrict__ wrouter, int n_experts, int top_k,
                                              float* __restrict__ route_w, const int* __restrict__ n_rows,
                                              int cast_first) {
    const int nchunk = hidden >> 3;
    const bf16_t* src = embed_row ? embed_row : h + (size_t)t * hidden;         // embedding row of the token, or the residual row
    bf16x8 hv[2], wv[2], av[2];
    bf16x8 gwv[MOE ? LA_MOE_MAX_E : 1][2];          // router rows: requested with everything else (one memory round trip)
    f32x4 sl[NS > 0 ? NS : 1][2][2];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int c = threadIdx.x + ci * 512;
        if (c < nchunk) {
            hv[ci] = *(const bf16x8*)(src + c * 8);
            wv[ci] = *(const bf16x8*)(nw + c * 8);
            if (MOE) {
#pragma unroll
                for (int e = 0; e < LA_MOE_MAX_E; ++e)
                    if (e < n_experts) gwv[e][ci] = *(const bf16x8*)(wrouter + (size_t)e * hidden + c * 8);
            }
            if (NS == 0 && addend) av[ci] = *(const bf16x8*)(addend + (size_t)t * hidden + c * 8);
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                const float* sp = slabs + ((size_t)s2 * LA_TB + t) * hidden + c * 8;
                sl[s2][ci][0] = *(const f32x4*)sp;
                sl[s2][ci][1] = *(const f32x4*)(sp + 4);
            }
        }
    }
    float vals[2][8];
    float ss = 0.f;
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int c = threadIdx.x + ci * 512;
        if (c < nchunk) {
            bf16x8 ho;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = bf2f((bf16_t)hv[ci][j]);
                if (NS > 0) {
                    float add = 0.f;
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2) add += sl[s2][ci][j >> 2][j & 3];
                    v = bfr(v + bfr(add));
                } else if (addend) {
                    v = bfr(v + bf2f((bf16_t)av[ci][j]));      // MoE: residual + final_hidden_states (bf16 + bf16)
                }
                vals[ci][j] = v;
                ho[j] = (short)f2bf(v);
                ss += v * v;
            }
            *(bf16x8*)(h + (size_t)t * hidden + c * 8) = ho;
        }
    }
    const float tot = block_sum<8>(ss, sh);
    const float rs = 1.0f / sqrtf(tot / (float)hidden + eps);
    float rl[MOE ? LA_MOE_MAX_E : 1];
#pragma unroll
    for (int e = 0; e < (MOE ? LA_MOE_MAX_E : 1); ++e) rl[e] = 0.f;
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int c = threadIdx.x + ci * 512;
        if (c < nchunk) {
            bf16x8 xo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // LlamaRMSNorm rounds once; Mistral/MixtralRMSNorm round the normalised value before the weight multiply
                const float nv = cast_first ? bfr(vals[ci][j] * rs) : vals[ci][j] * rs;
                xo[j] = (short)f2bf(bf2f((bf16_t)wv[ci][j]) * nv);
            }
            if constexpr (WT) {
                const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xp, 0, LA_TB * hidden * 2, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xo), xr, (int)(xp_offset(t, c * 8) * 2), 0, 16);     // aux 16 = sc1
            } else {
                *(bf16x8*)(xp + xp_offset(t, c * 8)) = xo;
            }
            if (MOE) {
#pragma unroll
                for (int e = 0; e < LA_MOE_MAX_E; ++e) {
                    if (e < n_experts) {
                        const bf16x8 gw = gwv[e][ci];
#pragma unroll
                        for (int j = 0; j < 8; ++j) rl[e] += bf2f((bf16_t)xo[j]) * bf2f((bf16_t)gw[j]);
                    }
                }
            }
        }
    }
    if (!MOE) return;
    // Router (MixtralSparseMoeBlock.forward, mixtral/modeling_mixtral.py:723-729): logits = gate(x) in the activation
    // dtype, softmax in fp32, top-k, renormalise, cast back.  route_w[t][e] = weight of expert e for this row or 0.
#pragma unroll
    for (int e = 0; e < LA_MOE_MAX_E; ++e) rl[e] = wave_sum(rl[e]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int e = 0; e < LA_MOE_MAX_E; ++e) shr[threadIdx.x >> 6][e] = rl[e];
    }
    __syncthreads();
    if (threadIdx.x < 64) moe_router_tail<LA_MOE_MAX_E>(shr, n_experts, top_k, t < n_rows[0], route_w + t * LA_MOE_MAX_E);
}

// Prefetch for consumer workgroup b (see PfDesc, la_kernels.h): every thread issues NT 16-byte loads back to back over the
// first bytes each wave of that workgroup will stream (chunk indices past the end re-read the last chunk: an L2 hit).  The loads
// are asm statements — hipcc otherwise consumes each pair before issuing the next — with default cache policy, and their data
// is dropped: the destination registers only become operands of the closing wait (pf_wait*), so that nothing is allocated over a
// load that is still in flight.
// ASM = false issues plain loads instead (tail prefetch inside a GEMM launch): the compiler then counts them in the vmcnt waits
// of the code that follows — an asm load is invisible to it, and a later s_waitcnt vmcnt(n) meant for an older load would drain
// the prefetch too — and pf_keep* at the end of the kernel is their only consumer.
template <int NT, bool ASM = true>
__device__ __forceinline__ void pf_issue(const PfDesc& p, int b, f32x4 (&v)[NT]) {
    b = b < p.n_consumers ? b : p.n_consumers - 1;
    const int bx = b % p.nbx, ks = b / p.nbx;
    const char* __restrict__ start = p.base + (size_t)bx * p.A + (size_t)ks * p.A2;
    const unsigned n0 = p.L[0] >> 4, n1 = p.RB > 1 ? p.L[1] >> 4 : 0u, n2 = p.RB > 2 ? p.L[2] >> 4 : 0u, n3 = p.RB > 3 ? p.L[3] >> 4 : 0u;
    const unsigned cps = n0 + n1 + n2 + n3;              // 16-byte chunks per consumer wave (> 0: the host builds no empty descriptor)
    const unsigned total = cps * (unsigned)p.NW;
    const unsigned e1 = n0, e2 = n0 + n1, e3 = n0 + n1 + n2;
    const unsigned bo0 = p.boff[0], bo1 = p.boff[1], bo2 = p.boff[2], bo3 = p.boff[3];      // scalars (kernel-argument loads)
    const unsigned cw0 = p.C[0], cw1 = p.C[1], cw2 = p.C[2], cw3 = p.C[3];
    // chunk index i = tid + j * blockDim -> (wave w, chunk r of the wave's cps): one division, then uniform steps
    const unsigned step_w = blockDim.x / cps, step_r = blockDim.x - step_w * cps;
    unsigned w = threadIdx.x / cps, r = threadIdx.x - w * cps, i = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const bool in = i < total;
        const unsigned ww = in ? w : (unsigned)p.NW - 1u, rr = in ? r : cps - 1u;
        const bool a1 = rr >= e1, a2 = rr >= e2, a3 = rr >= e3;
        const unsigned rb0 = a3 ? e3 : a2 ? e2 : a1 ? e1 : 0u;
        const unsigned bo = a3 ? bo3 : a2 ? bo2 : a1 ? bo1 : bo0;
        const unsigned cw = a3 ? cw3 : a2 ? cw2 : a1 ? cw1 : cw0;
        const char* addr = start + (size_t)bo + (size_t)ww * cw + (size_t)(rr - rb0) * 16;
        if constexpr (ASM) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[j]) : "v"(addr));
        else v[j] = *(const f32x4*)addr;
        i += blockDim.x; w += step_w; r += step_r;
        if (r >= cps) { r -= cps; ++w; }
    }
}
__device__ __forceinline__ void pf_keep8(f32x4 (&v)[8]) {        // consumer of 8 plain prefetch loads (nothing is emitted)
    asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
}
__device__ __forceinline__ void pf_wait16(f32x4 (&v)[16]) {
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
                 "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]) : "memory");
}
// An appended prefetch workgroup of an idle-window launch: optional start delay (the launch's own loads go first), 16 loads per
// thread, and the wave retires once they have landed in L2.
__device__ __forceinline__ void pf_body(const PfDesc& p, int b) {
    if (b >= p.n_consumers) return;
    for (int i = 0; i < p.delay; ++i) __builtin_amdgcn_s_sleep(32);
    f32x4 v[16];
    pf_issue<16>(p, b, v);
    pf_wait16(v);
}

#if LA_LAB
#include "lab/k_pf_only.inc"            // the prefetch workgroups of a descriptor as their own launch (lab knobs 26-30: forked graph branch, measured slower)
#endif

template <int NS, bool MOE>
__global__ __launch_bounds__(512) void k_row_norm(const bf16_t* __restrict__ embed, const int* __restrict__ ids,
                                                   bf16_t* __restrict__ h, const float* __restrict__ slabs,
                                                   const bf16_t* __restrict__ nw, int hidden, float eps,
                                                   bf16_t* __restrict__ xp, const bf16_t* __restrict__ addend,
                                                   const bf16_t* __restrict__ wrouter, int n_experts, int top_k,
                                                   float* __restrict__ route_w, const int* __restrict__ n_rows,
                                                   int cast_first, PfDesc pf) {
    __shared__ float sh[8];
    __shared__ float shr[MOE ? 8 : 1][LA_MOE_MAX_E];
    if (blockIdx.x >= LA_TB) { pf_body(pf, (int)blockIdx.x - LA_TB); return; }      // appended idle-window prefetch workgroups
    row_norm_body<NS, MOE>(blockIdx.x, sh, shr, embed ? embed + (size_t)ids[blockIdx.x] * hidden : nullptr, h, slabs, nw, hidden, eps,
                           xp, addend, wrouter, n_experts, top_k, route_w, n_rows, cast_first);
}

#if LA_LAB
#include "lab/k_row_norm4.inc"          // four workgroups per norm row (lab knob 19: measured neutral, never default)
#endif

// Step head (single-sequence step): k_build_tree_inputs + the embedding row kernel in ONE launch.  Workgroup t expands its own
// row of the step input (ids / 64-bit ancestor mask / position = committed keys + popcount - 1, the model hook of
// modeling_llama.py:584-588; pad rows t >= T see themselves only) straight from the caller's input block — the zero-copy pinned
// host block of la_llama_step — and runs the embedding gather + RMSNorm of that row; workgroup 0 also latches T and the mode.
__global__ __launch_bounds__(512) void k_step_head(const int* __restrict__ in, int* __restrict__ state, int* __restrict__ pos,
                                                    unsigned long long* __restrict__ rowmask, int* __restrict__ ids,
                                                    const bf16_t* __restrict__ embed, bf16_t* __restrict__ h,
                                                    const bf16_t* __restrict__ nw, int hidden, float eps, bf16_t* __restrict__ xp,
                                                    int cast_first, PfDesc pf, unsigned long long* gran, int n_gran) {
    __shared__ float sh[8];
    if (blockIdx.x >= LA_TB) { pf_body(pf, (int)blockIdx.x - LA_TB); return; }
    const int t = blockIdx.x;
    // the granule words of this step's k_row_norm4 launches (each used once per step): zeroed here, a kernel boundary ahead of their first use
    for (int i = t * 512 + (int)threadIdx.x; i < n_gran; i += LA_TB * 512) gran[i] = 0ull;
    const int T = in[LA_IN_T];
    const unsigned long long rm = (t < T) ? ((const unsigned long long*)(in + LA_IN_ROWMASK))[t] : (1ull << t);
    const int id = (t < T) ? in[LA_IN_IDS + t] : 0;
    if (threadIdx.x == 0) {
        rowmask[t] = rm;
        ids[t] = id;
        pos[t] = state[LA_ST_NKEYS] + __popcll(rm) - 1;
        if (t == 0) { state[LA_ST_T] = T; state[LA_ST_MODE] = in[LA_IN_MODE]; }
    }
    row_norm_body<0, false>(t, sh, nullptr, embed + (size_t)id * hidden, h, nullptr, nw, hidden, eps, xp, nullptr, nullptr, 0, 0,
                            nullptr, nullptr, cast_first);
}

// In-kernel hand-over from the 64 producer workgroups (lowest block ids, dispatched first, so a consumer never waits on a
// workgroup that could be queued behind it) to every workgroup of the grid: producers publish their stores with an
// agent-scope release and bump `counter`; everybody spins until it reaches `target`, then acquires.
__device__ __forceinline__ void handover_signal(int* counter, bool wt = false) {
    if (wt) {
        // write-through payload (sc1 stores): EVERY storing wave drains its own stores, the barrier collects the waves, one lane
        // bumps the counter — no L2 write-back (Guideline 16 R1; pitfall 14: draining lane 0 alone is not enough)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    __syncthreads();                 // every wave's stores have completed (the barrier carries vmcnt(0)) ...
    if (threadIdx.x == 0) {
        // ... one lane writes the XCD's dirty L2 lines back (agent-scope release), DRAINS that write-back — hipcc may drop the
        // s_waitcnt behind buffer_wbl2 when it thinks the wave has nothing outstanding, and the flag would overtake the data —
        // and only then bumps the counter
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void handover_wait(int* counter, int target) {
    if (threadIdx.x == 0) {
        // relaxed polls (an acquire load would invalidate the caches on every iteration), one acquire at the end
        // bounded: the producers are the lowest block ids; should a dispatcher ever start them late enough for ~1 s of
        // polling to pass, abort the launch (HIP error) instead of hanging the queue
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 22)) __builtin_trap();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Skinny GEMM: out[64][N] = x[64][K] . W[N][K]^T, W streamed exactly once from HBM.
//   grid = (N/(32*RB), ksplit); 8 waves per workgroup split the workgroup's K range, each wave
//   streams RB row-blocks of W (A operand, nontemporal 1 KiB loads) and the matching x tiles
//   (B operand, L2-resident), D tiles deep; partial sums are tree-reduced through LDS in a fixed
//   order (deterministic), then wave 0 runs the epilogue.
//   Replaces nn.Linear in LlamaAttention/LlamaMLP/lm_head (modeling_llama.py:172-186,222-224,296,769).
// ---------------------------------------------------------------------------------------------
enum { EPI_SLAB = 0, EPI_SWIGLU = 1, EPI_LOGITS = 2, EPI_QKV = 3 };

struct GemmArgs {
    const bf16_t* wp;
    const bf16_t* xp;
    int K16;             // K / 16
    int N;               // output features (for SWIGLU: ffn, the packed matrix has 2*ffn rows)
    float* slabs;        // EPI_SLAB: [ksplit][64][N] fp32
    bf16_t* act_xp;      // EPI_SWIGLU: packed [64][N]
    bf16_t* logits;      // EPI_LOGITS: [64][N] bf16 row-major or null
    float* cand_val;     // EPI_LOGITS: [gridDim.x][64]
    int* cand_idx;
    // EPI_QKV: fused RoPE + fragment writes (rows permuted at pack time, see lk_qkv_row_perm)
    const int* pos;
    const bf16_t* rcos;
    const bf16_t* rsin;
    bf16_t* qf;
    bf16_t* kfresh;
    bf16_t* vfresh;
    int nh, nkv;
    // mixture-of-experts: routing weights of this expert, one float per token row at stride LA_MOE_MAX_E; the whole
    // launch returns at once when no row routes to the expert (its weights are then never read)
    const float* route_col;
    // merged expert launch (ex_on): one grid dimension enumerates the experts of a MoE layer, whose weight images, SwiGLU
    // outputs and split-K slabs are equally spaced; expert e uses route_col + e and returns at once when no row routes to it
    int ex_on;
    long ex_w, ex_x, ex_act, ex_slab;      // element strides per expert (weights, x operand, act_xp, slabs)
    int prio_hi;        // 8-wave kernels: s_setprio level of waves 4..7 (the SIMD partners that are served last), 0 = default
    int kskew;          // 8-wave kernels: share (1/64ths) of the K range given to waves 0..3; 0 = even split (see wave_krange)
    int slab_wt;        // EPI_SLAB: 1 = store the split-K partials write-through (sc1)
    int dbg_noepi;      // measurement aid (scripts/gpu_ab.py): return after the streaming loop, before the reduction/epilogue
    long long* dbg_times;   // measurement aid: [workgroup][wave][4] wall_clock64() at entry / loop end / exit (null in production)
};

__device__ __forceinline__ bool expert_unused(const float* route_col) {
    return route_col != nullptr && __ballot(route_col[(threadIdx.x & 63) * LA_MOE_MAX_E] != 0.f) == 0ull;
}

// K range of one wave.  Measured on MI355X (scripts/gpu_ab.py timeline): with two waves per SIMD the wave in hardware slot 0
// (waves 0..3 of a 512-thread workgroup) is served ~2.2x faster than its SIMD partner while both are streaming, so an even
// split leaves the partner alone for the last third of the launch with half the bytes in flight.  kskew/64 of the k-tiles go
// to waves 0..3 so that both halves finish together; the split is static, i.e. the summation order stays deterministic.
template <int NW>
__device__ __forceinline__ void wave_krange(int t0, int twg, int wave, int kskew, int& wb, int& cnt) {
    if (NW == 8 && kskew > 0) {
        int told = (int)(((long)twg * kskew) >> 6);
        told = told > twg ? twg : told;
        const int grp = wave >> 2, wi = wave & 3;
        const int tg = grp ? twg - told : told, base = grp ? told : 0;
        const int q = tg >> 2, r = tg & 3;
        wb = t0 + base + wi * q + (wi < r ? wi : r);
        cnt = q + (wi < r ? 1 : 0);
    } else {
        const int q = twg / NW, r = twg - q * NW;
        wb = t0 + wave * q + (wave < r ? wave : r);
        cnt = q + (wave < r ? 1 : 0);
    }
}

// Kernel-argument preload (hipcc -mllvm -amdgpu-kernarg-preload-count=16, build.sh): the first 16 dwords of SCALAR arguments
// arrive in SGPRs with the wave, aggregates do not.  Everything the prologue needs before the first weight load is issued
// (image bases, K, the expert switch) is therefore passed as leading scalars — the same values as the struct fields, which the
// launchers keep filling — so that no s_load round trip stands between dispatch and the first HBM request (measured on the
// single-wave kernels of the step: -0.2 us per launch).  kfl = ex_on | kskew << 1 | prio_hi << 8.
// HW = true: the x operand is produced by OTHER workgroups of the same launch (role-fused launches, k_gateup_down): the first
// weight tile-sets are requested, then the wave waits for ho_counter == ho_target (handover_wait), then loads x.
struct BlockPos { int bx, by, bz, gx, gy; };
template <int RB, int EPI, int D, int NW, bool HW = false>
__device__ __forceinline__ void gemm64_body(const bf16_t* __restrict__ wp_s, const bf16_t* xp_s,
                                            const float* __restrict__ route_col_s, int K16_s, int kfl, const GemmArgs& a,
                                            const BlockPos bp, float* red_, int* ho_counter, int ho_target) {
    float (*red)[RB * 2 * 16 * 64] = (float (*)[RB * 2 * 16 * 64])red_;
    const int ex = (kfl & 1) ? bp.bz : 0;
    if (expert_unused(route_col_s ? route_col_s + ex : nullptr)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nb0 = bp.bx * RB;
    const int ksplit = bp.gy, ks = bp.by;
    long long* const stamp = a.dbg_times ? a.dbg_times + ((size_t)(bp.by * bp.gx + bp.bx) * NW + wave) * 8 : nullptr;
    if (stamp && lane == 0) { stamp[0] = wall_clock64(); stamp[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); }
    const int t0 = (int)(((long)K16_s * ks) / ksplit);
    const int t1 = (int)(((long)K16_s * (ks + 1)) / ksplit);
    // this wave's contiguous k-tile range
    int wb, cnt;
    const int kskew = (kfl >> 1) & 127, prio_hi = kfl >> 8;
    wave_krange<NW>(t0, t1 - t0, wave, kskew, wb, cnt);
    if (NW == 8 && prio_hi > 0 && wave >= 4) { if (prio_hi == 1) __builtin_amdgcn_s_setprio(1); else if (prio_hi == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); }
    const int ngroups = (cnt + D - 1) / D;            // groups of D tiles; the last one may be partial
    const int last_valid = cnt - (ngroups - 1) * D;   // valid slots in the last group (1..D)

    // integer offsets from the kernel-argument bases (not mutated pointers) keep the loads in the global
    // address space: a loop-carried pointer degrades to flat_load, which ties vmcnt and lgkmcnt together
    const bf16x8* __restrict__ wbase = (const bf16x8*)(wp_s + ((kfl & 1) ? (size_t)ex * a.ex_w : (size_t)0));
    typedef const bf16x8* __restrict__ xptr_r;
    typedef const bf16x8* xptr_n;                                       // HW: written by other workgroups of this launch
    typename std::conditional<HW, xptr_n, xptr_r>::type xbase = (const bf16x8*)(xp_s + ((kfl & 1) ? (size_t)ex * a.ex_x : (size_t)0));
    unsigned woff[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) woff[rb] = (unsigned)(((nb0 + rb) * K16_s + wb) * 64 + lane);
    unsigned xoff = (unsigned)(wb * 128 + lane);

    f32x16 acc[RB][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][tb][i] = 0.f;

    if (ngroups > 0) {
        // Branch-free software pipeline: D tile-sets (W fragment(s) + 2 x fragments each) are always in flight;
        // slot d is consumed and immediately refilled with the tile D ahead, so hipcc emits counted
        // s_waitcnt vmcnt((D-1)*(RB+2)).  Throughput of a weight-streaming wave = bytes in flight / latency,
        // so D is chosen to fill the 256-VGPR budget.  Slots past the end of the range reload the wave's
        // first tile (always valid memory) and their MFMAs are skipped by a wave-uniform branch.
        bf16x8 fa[D][RB], fb[D][2];
        const int n1 = ngroups == 1 ? last_valid : D;     // valid slots of group 0
        if constexpr (HW) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int dd = d < n1 ? d : 0;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) fa[d][rb] = __builtin_nontemporal_load(wbase + woff[rb] + dd * 64);
            }
            handover_wait(ho_counter, ho_target);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int dd = d < n1 ? d : 0;
                fb[d][0] = xbase[xoff + dd * 128];
                fb[d][1] = xbase[xoff + dd * 128 + 64];
            }
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int dd = d < n1 ? d : 0;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) fa[d][rb] = __builtin_nontemporal_load(wbase + woff[rb] + dd * 64);
                fb[d][0] = xbase[xoff + dd * 128];
                fb[d][1] = xbase[xoff + dd * 128 + 64];
            }
        }
        for (int g = 1; g < ngroups; ++g) {
            if (stamp && lane == 0 && g == (ngroups >> 1)) stamp[3] = wall_clock64();
            const int nv = (g == ngroups - 1) ? last_valid : D;   // valid slots of the group being fetched
#pragma unroll
            for (int d = 0; d < D; ++d) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    acc[rb][0] = LA_MFMA(fa[d][rb], fb[d][0], acc[rb][0], 0, 0, 0);
                    acc[rb][1] = LA_MFMA(fa[d][rb], fb[d][1], acc[rb][1], 0, 0, 0);
                }
                const int dd = (d < nv ? g * D + d : 0);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) fa[d][rb] = __builtin_nontemporal_load(wbase + woff[rb] + dd * 64);
                fb[d][0] = xbase[xoff + dd * 128];
                fb[d][1] = xbase[xoff + dd * 128 + 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (d < last_valid) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    acc[rb][0] = LA_MFMA(fa[d][rb], fb[d][0], acc[rb][0], 0, 0, 0);
                    acc[rb][1] = LA_MFMA(fa[d][rb], fb[d][1], acc[rb][1], 0, 0, 0);
                }
            }
        }
    }

    // ---- deterministic cross-wave reduction: every wave parks its partial tile in LDS, ONE barrier, then wave w
    //      sums (fixed order p = 0..NW-1) and finishes the slice {token block w&1, register groups of w>>1} for all
    //      row-blocks — the epilogue runs on all waves instead of serialising on wave 0.  The buffer is laid out in
    //      16-byte units [wave][rb][tb][i/4][lane] so that both sides use ds_write_b128 / ds_read_b128.
    if (stamp && lane == 0) stamp[1] = wall_clock64();
    if (a.dbg_noepi) return;
    f32x4* red4 = (f32x4*)&red[0][0];
    {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const f32x4 v = {acc[rb][tb][4 * i4], acc[rb][tb][4 * i4 + 1], acc[rb][tb][4 * i4 + 2], acc[rb][tb][4 * i4 + 3]};
                    red4[((wave * RB * 2 + rb * 2 + tb) * 4 + i4) * 64 + lane] = v;
                }
    }
    constexpr int GPW = 8 / NW;                  // register groups (of 4 features) per wave: NW=4 -> 2, NW=8 -> 1
    const int tb = wave & 1, g0 = (wave >> 1) * GPW;
    const int tl = lane & 31, hh = lane >> 5;
    // EPI_QKV: this lane's RoPE operands (dependent pos -> cos/sin loads) are requested before the barrier
    bf16x4 rc[GPW], rs4[GPW];
    if constexpr (EPI == EPI_QKV) {
        const int u = bp.bx & 1;
        const int ps = a.pos[tb * 32 + tl];
#pragma unroll
        for (int gg = 0; gg < GPW; ++gg) {
            const int dlo = 32 * u + 8 * (g0 + gg) + 4 * hh;      // d in [0,64): cos/sin column
            rc[gg] = *(const bf16x4*)(a.rcos + (size_t)ps * 64 + dlo);
            rs4[gg] = *(const bf16x4*)(a.rsin + (size_t)ps * 64 + dlo);
        }
    }
    __syncthreads();
    float fin[RB][GPW][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int gg = 0; gg < GPW; ++gg) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < NW; ++p) v += red4[((p * RB * 2 + rb * 2 + tb) * 4 + (g0 + gg)) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) fin[rb][gg][j] = v[j];
        }
    const int tok = tb * 32 + tl;
    if constexpr (EPI == EPI_SLAB) {
        float* o = a.slabs + (size_t)ex * a.ex_slab + (size_t)ks * LA_TB * a.N;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int gg = 0; gg < GPW; ++gg) {
                f32x4 v = {fin[rb][gg][0], fin[rb][gg][1], fin[rb][gg][2], fin[rb][gg][3]};
                float* dstp = o + (size_t)tok * a.N + (nb0 + rb) * 32 + 8 * (g0 + gg) + 4 * hh;
                if (a.slab_wt) {
                    // write-through (sc1): the partials go straight to memory, where the row kernel of the NEXT launch reads them from any
                    // XCD — nothing is left dirty in this XCD's L2 for the kernel boundary to write back (la_lab_set key 23)
                    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)dstp, 0, 16, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), sr, 0, 0, 16);
                } else {
                    *(f32x4*)dstp = v;
                }
            }
    } else if constexpr (EPI == EPI_SWIGLU) {
        // rb 0 = gate rows, rb 1 = up rows of the same 32 features (interleaved packing).
        // act = bf16(silu(bf16(g)) * bf16(u))  — LlamaMLP.forward, modeling_llama.py:185-186
        static_assert(EPI != EPI_SWIGLU || RB == 2, "swiglu needs gate/up pair");
        const int jb = bp.bx;   // feature block
#pragma unroll
        for (int gg = 0; gg < GPW; ++gg) {
            bf16x4 pk;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float gv = bfr(fin[0][gg][j]);
                float uv = bfr(fin[RB - 1][gg][j]);
                float sv = bfr(gv / (1.0f + expf(-gv)));
                pk[j] = (short)f2bf(sv * uv);
            }
            const int f = jb * 32 + 8 * (g0 + gg) + 4 * hh;    // 4 consecutive features f..f+3
            *(bf16x4*)(a.act_xp + (size_t)ex * a.ex_act + xp_offset(tok, f)) = pk;
        }
    } else if constexpr (EPI == EPI_QKV) {
        // Workgroup b owns head slot b>>1, half u=b&1: row-block 0 = dims 32u+[0,32), row-block 1 = the RoPE
        // partners 64+32u+[0,32) (rows permuted at pack time).  q/k: rotate-half RoPE in bf16 arithmetic
        // (apply_rotary_pos_emb, modeling_llama.py:154-169) -> QF / fresh KF fragments; v -> fresh VF fragments.
        static_assert(EPI != EPI_QKV || RB == 2, "qkv epilogue needs the (d, d+64) row-block pair");
        const int slot = bp.bx >> 1, u = bp.bx & 1;
#pragma unroll
        for (int gg = 0; gg < GPW; ++gg) {
            const int dlo = 32 * u + 8 * (g0 + gg) + 4 * hh, dhi = dlo + 64;
            if (slot < a.nh + a.nkv) {
                bf16_t* dst = slot < a.nh ? a.qf + (size_t)slot * 8192 : a.kfresh + (size_t)(slot - a.nh) * 8192;
                bf16x4 olo, ohi;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xl = bfr(fin[0][gg][j]), xh = bfr(fin[RB - 1][gg][j]);
                    const float c = bf2f((bf16_t)rc[gg][j]), sn = bf2f((bf16_t)rs4[gg][j]);
                    olo[j] = (short)f2bf(bfr(xl * c) + bfr(-xh * sn));
                    ohi[j] = (short)f2bf(bfr(xh * c) + bfr(xl * sn));
                }
                *(bf16x4*)(dst + rf_offset(tok, dlo)) = olo;
                *(bf16x4*)(dst + rf_offset(tok, dhi)) = ohi;
            } else {
                bf16_t* dst = a.vfresh + (size_t)(slot - a.nh - a.nkv) * 8192;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dst[vf_offset(tok, dlo + j)] = f2bf(fin[0][gg][j]);
                    dst[vf_offset(tok, dhi + j)] = f2bf(fin[RB - 1][gg][j]);
                }
            }
        }
    } else {
        // logits rounded to bf16 (lm_head output dtype, modeling_llama.py:769); per-token argmax candidate over this
        // wave's features with lowest-index tie-break (torch.argmax on CPU returns the first maximum).
        float best = -INFINITY;
        int bidx = 0x7fffffff;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int gg = 0; gg < GPW; ++gg) {
                bf16x4 pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf16_t hv = f2bf(fin[rb][gg][j]);
                    pk[j] = (short)hv;
                    float v = bf2f(hv);
                    int idx = (nb0 + rb) * 32 + 8 * (g0 + gg) + 4 * hh + j;
                    if (v > best || (v == best && idx < bidx)) { best = v; bidx = idx; }
                }
                if (a.logits)
                    *(bf16x4*)(a.logits + (size_t)tok * a.N + (nb0 + rb) * 32 + 8 * (g0 + gg) + 4 * hh) = pk;
            }
        float ob = __shfl_xor(best, 32, 64);
        int oi = __shfl_xor(bidx, 32, 64);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
        if (hh == 0) {      // candidate slot: (workgroup, register-group owner) x token
            const size_t slot = (size_t)bp.bx * (NW / 2) + (wave >> 1);
            a.cand_val[slot * LA_TB + tok] = best;
            a.cand_idx[slot * LA_TB + tok] = bidx;
        }
    }
    if (stamp && lane == 0) stamp[2] = wall_clock64();
}

template <int RB, int EPI, int D, int NW>
__global__ __launch_bounds__(NW * 64) void k_gemm64(const bf16_t* __restrict__ wp_s, const bf16_t* __restrict__ xp_s,
                                                     const float* __restrict__ route_col_s, int K16_s, int kfl, GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float red[NW][RB * 2 * 16 * 64];
    const BlockPos bp = {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y};
    gemm64_body<RB, EPI, D, NW, false>(wp_s, xp_s, route_col_s, K16_s, kfl, a, bp, &red[0][0], nullptr, 0);
}

// ---------------------------------------------------------------------------------------------
// Balanced skinny GEMM ("one workgroup per CU"): the per-CU streaming rate is capped (~22 GB/s of the chip's
// ~5.7 TB/s, scripts/probe_stream.hip), so a launch is as slow as its busiest CU.  Here every workgroup owns exactly
// R = rows / n_wg weight rows per matrix, stored as RB "virtual" 32-row blocks whose last one is partial: the packed
// image pads it to 32 rows (zeros, never read), lanes of invalid rows re-read the last valid row (same cache line, no
// extra HBM bytes) and their accumulator rows are simply never stored.  Full K per workgroup (no split-K slabs);
// 8 waves split K.  EPI_SWIGLU: blocks {G0,G1,U0,U1} (R = ffn/n_wg <= 64); EPI_LOGITS: 4 blocks (R <= 128);
// EPI_QKV: blocks {lo,hi} of R RoPE pairs (R <= 32).
// ---------------------------------------------------------------------------------------------
struct GemmRArgs {
    GemmArgs g;
    int R;          // valid rows per matrix per workgroup
    int nv[4];      // valid rows of each virtual block (epilogue mask)
    int nvl[4];     // stored rows = nv rounded up to a multiple of 4, so that a tile is a whole number of 128-B lines
    int boff[4];    // 16-byte-chunk offset of each block inside the workgroup's region
    int wg_chunks;  // 16-byte chunks per workgroup region
    // fused producer (NSF > 0): workgroups 0..63 first run the residual + RMSNorm row kernel that produces g.xp
    // (row = block id) while every workgroup's first weight tiles are already in flight, then hand over in-kernel
    const float* fn_slabs;
    bf16_t* fn_h;
    const bf16_t* fn_nw;
    int fn_hidden, fn_cast, fn_wt;
    float fn_eps;
    int* fn_counter;
    // tail prefetch: after its streaming loop every workgroup pulls the first k-tiles the same-numbered workgroup of the NEXT GEMM
    // (down_proj after gate/up: no idle-window kernel sits between them) will stream into L2, under its reduction + epilogue
    PfDesc pf;
};

// Leading scalars = the prologue's operands (kernarg preload, see k_gemm64): nvl_pk = nvl[0..3] one byte each.
template <int RB, int EPI, int D, int NW, int NSF = 0>
__device__ __forceinline__ void gemm64r_body(const bf16_t* __restrict__ wp_s, const bf16_t* xp_s,
                                             const float* __restrict__ route_col_s, int K16_s, int kfl, int wg_chunks_s,
                                             unsigned nvl_pk, int boff0, int boff1, int boff2, int boff3, const GemmRArgs& ra,
                                             const BlockPos bp, float* redr) {
    const GemmArgs& a = ra.g;
    const int ex = (kfl & 1) ? bp.by : 0;
    if (expert_unused(route_col_s ? route_col_s + ex : nullptr)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long long* const stamp = a.dbg_times ? a.dbg_times + ((size_t)(bp.by * bp.gx + bp.bx) * NW + wave) * 8 : nullptr;
    if (stamp && lane == 0) { stamp[0] = wall_clock64(); stamp[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); }
    int wb, cnt;
    const int kskew = (kfl >> 1) & 127, prio_hi = kfl >> 8;
    wave_krange<NW>(0, K16_s, wave, kskew, wb, cnt);
    if (NW == 8 && prio_hi > 0 && wave >= 4) { if (prio_hi == 1) __builtin_amdgcn_s_setprio(1); else if (prio_hi == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); }
    const int ngroups = (cnt + D - 1) / D;
    const int last_valid = cnt - (ngroups - 1) * D;
    const bf16x8* __restrict__ wbase = (const bf16x8*)(wp_s + ((kfl & 1) ? (size_t)ex * a.ex_w : (size_t)0));
    typedef const bf16x8* __restrict__ xptr_r;
    typedef const bf16x8* xptr_n;                                       // fused producers write g.xp inside this kernel
    typename std::conditional<(NSF > 0), xptr_n, xptr_r>::type xbase = (const bf16x8*)xp_s;
    unsigned woff[RB], wstr[RB];      // per-lane chunk offset of k-tile wb, and chunks per k-tile (2 * valid rows)
    const int boff_s[4] = {boff0, boff1, boff2, boff3};
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int nvb = (int)((nvl_pk >> (8 * rb)) & 255u);
        const int rr = (lane & 31) < nvb ? (lane & 31) : nvb - 1;     // rows past the stored ones re-read the last one
        wstr[rb] = (unsigned)(2 * nvb);
        woff[rb] = (unsigned)bp.bx * (unsigned)wg_chunks_s + (unsigned)boff_s[rb] + (unsigned)wb * wstr[rb]
                   + (unsigned)((lane >> 5) * nvb + rr);
    }
    unsigned xoff = (unsigned)(wb * 128 + lane);
    f32x16 acc[RB][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][tb][i] = 0.f;
    if (ngroups > 0) {
        bf16x8 fa[D][RB], fb[D][2];
        const int n1 = ngroups == 1 ? last_valid : D;
        if constexpr (NSF > 0) {
            // The activations do not exist yet.  Producers run the row stage BEFORE touching their weights (a wave's loads
            // return in order: the row operands would queue behind ~16 KiB of weight tiles per wave); everybody else has
            // its first weight tiles in flight while waiting.
            static_assert(NW == 8, "the fused row kernel is written for 512 threads");
            const bool producer = bp.bx < LA_TB;
            if (producer) {
                if (ra.fn_wt)
                    row_norm_body<NSF, false, true>(bp.bx, redr, nullptr, nullptr, ra.fn_h, ra.fn_slabs, ra.fn_nw,
                                                    ra.fn_hidden, ra.fn_eps, (bf16_t*)xp_s, nullptr, nullptr, 0, 0, nullptr, nullptr,
                                                    ra.fn_cast);
                else
                    row_norm_body<NSF, false>(bp.bx, redr, nullptr, nullptr, ra.fn_h, ra.fn_slabs, ra.fn_nw,
                                              ra.fn_hidden, ra.fn_eps, (bf16_t*)xp_s, nullptr, nullptr, 0, 0, nullptr, nullptr,
                                              ra.fn_cast);
                handover_signal(ra.fn_counter, ra.fn_wt != 0);
            }
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int dd = d < n1 ? d : 0;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) fa[d][rb] = __builtin_nontemporal_load(wbase + woff[rb] + dd * wstr[rb]);
            }
            handover_wait(ra.fn_counter, LA_TB);
            const bf16x8* xv = (const bf16x8*)xp_s;            // not __restrict__: written by the producers above
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int dd = d < n1 ? d : 0;
                fb[d][0] = xv[xoff + dd * 128];
                fb[d][1] = xv[xoff + dd * 128 + 64];
            }
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int dd = d < n1 ? d : 0;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) fa[d][rb] = __builtin_nontemporal_load(wbase + woff[rb] + dd * wstr[rb]);
                fb[d][0] = xbase[xoff + dd * 128];
                fb[d][1] = xbase[xoff + dd * 128 + 64];
            }
        }
        for (int g = 1; g < ngroups; ++g) {
            if (stamp && lane == 0 && g == (ngroups >> 1)) stamp[3] = wall_clock64();
            const int nv2 = (g == ngroups - 1) ? last_valid : D;
#pragma unroll
            for (int d = 0; d < D; ++d) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    acc[rb][0] = LA_MFMA(fa[d][rb], fb[d][0], acc[rb][0], 0, 0, 0);
                    acc[rb][1] = LA_MFMA(fa[d][rb], fb[d][1], acc[rb][1], 0, 0, 0);
                }
                const int dd = (d < nv2 ? g * D + d : 0);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) fa[d][rb] = __builtin_nontemporal_load(wbase + woff[rb] + dd * wstr[rb]);
                fb[d][0] = xbase[xoff + dd * 128];
                fb[d][1] = xbase[xoff + dd * 128 + 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (d < last_valid) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    acc[rb][0] = LA_MFMA(fa[d][rb], fb[d][0], acc[rb][0], 0, 0, 0);
                    acc[rb][1] = LA_MFMA(fa[d][rb], fb[d][1], acc[rb][1], 0, 0, 0);
                }
            }
        }
        if constexpr (EPI == EPI_SWIGLU && NSF == 0) {
            // Tail prefetch below: make every tile slot a consumed value, so that the compiler's scoreboard is empty on BOTH sides
            // of the pf_on branch (slots past last_valid hold re-reads nobody uses; a wait for them placed after the join would
            // also drain the prefetch loads).  At run time everything has returned by now: no instruction is emitted but waits.
#pragma unroll
            for (int d = 0; d < D; ++d) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) asm volatile("" :: "v"(fa[d][rb]));
                asm volatile("" :: "v"(fb[d][0]), "v"(fb[d][1]));
            }
        }
    }

    if (stamp && lane == 0) stamp[1] = wall_clock64();
    if (a.dbg_noepi) return;
    f32x4 pfv[8];
    const bool pf_on = EPI == EPI_SWIGLU && NSF == 0 && ra.pf.base != nullptr;        // wave-uniform
    if constexpr (EPI == EPI_SWIGLU && NSF == 0) {
        if (pf_on) pf_issue<8, false>(ra.pf, bp.bx, pfv);
    }
    // ---- cross-wave reduction through LDS in 16-byte units [wave][rb][i/4][lane] (ds_write_b128 / ds_read_b128), fixed
    //      summation order p = 0..NW-1 (deterministic), then every wave finishes a fixed slice.
    const int tl = lane & 31, hh = lane >> 5;
    f32x4* red4 = (f32x4*)redr;
    if constexpr (EPI == EPI_QKV) {
        static_assert(EPI != EPI_QKV || (RB == 2 && NW == 8), "qkv layout");
        if ((ra.R & 3) == 0) {
            // ONE pass, both token blocks parked at once (128 KiB): wave -> (token block, register group); a lane owns 4
            // consecutive RoPE pairs, i.e. 8-byte runs of the QF/KF fragments and of the cos/sin rows.
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4) {
                        const f32x4 v = {acc[rb][tb][4 * i4], acc[rb][tb][4 * i4 + 1], acc[rb][tb][4 * i4 + 2], acc[rb][tb][4 * i4 + 3]};
                        red4[(((tb * NW + wave) * RB + rb) * 4 + i4) * 64 + lane] = v;
                    }
            const int tbe = wave >> 2, gi = wave & 3;
            const int tok = tbe * 32 + tl;
            const int f0 = 8 * gi + 4 * hh;
            const bool live = f0 < ra.nv[0];
            const int pr = ra.R * bp.bx + f0, slot = pr >> 6, dlo = pr & 63, dhi = dlo + 64;
            const bool rope = slot < a.nh + a.nkv;
            bf16x4 c4 = {0, 0, 0, 0}, s4 = {0, 0, 0, 0};
            if (live && rope) {          // dependent pos -> cos/sin loads are in flight across the barrier
                const int ps = a.pos[tok];
                c4 = *(const bf16x4*)(a.rcos + (size_t)ps * 64 + dlo);
                s4 = *(const bf16x4*)(a.rsin + (size_t)ps * 64 + dlo);
            }
            __syncthreads();
            if (!live) return;
            f32x4 xl = {0.f, 0.f, 0.f, 0.f}, xh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < NW; ++p) {
                xl += red4[(((tbe * NW + p) * RB + 0) * 4 + gi) * 64 + lane];
                xh += red4[(((tbe * NW + p) * RB + 1) * 4 + gi) * 64 + lane];
            }
            if (rope) {
                bf16_t* dst = slot < a.nh ? a.qf + (size_t)slot * 8192 : a.kfresh + (size_t)(slot - a.nh) * 8192;
                bf16x4 olo, ohi;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float c = bf2f((bf16_t)c4[j]), sn = bf2f((bf16_t)s4[j]);
                    const float bl = bfr(xl[j]), bh = bfr(xh[j]);
                    olo[j] = (short)f2bf(bfr(bl * c) + bfr(-bh * sn));
                    ohi[j] = (short)f2bf(bfr(bh * c) + bfr(bl * sn));
                }
                *(bf16x4*)(dst + rf_offset(tok, dlo)) = olo;
                *(bf16x4*)(dst + rf_offset(tok, dhi)) = ohi;
            } else {
                bf16_t* dst = a.vfresh + (size_t)(slot - a.nh - a.nkv) * 8192;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dst[vf_offset(tok, dlo + j)] = f2bf(xl[j]);
                    dst[vf_offset(tok, dhi + j)] = f2bf(xh[j]);
                }
            }
            if (stamp && lane == 0) stamp[2] = wall_clock64();
            return;
        }
    }
    constexpr int SL = RB * 4 / NW;            // (row-block, register-group) slices per wave and pass (SWIGLU/QKV pair up)
    static_assert(RB * 4 % NW == 0 || EPI == EPI_SWIGLU || EPI == EPI_QKV, "slice split");
    float best = -INFINITY;
    int bidx = 0x7fffffff;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        if (tb) __syncthreads();
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 v = {acc[rb][tb][4 * i4], acc[rb][tb][4 * i4 + 1], acc[rb][tb][4 * i4 + 2], acc[rb][tb][4 * i4 + 3]};
                red4[((wave * RB + rb) * 4 + i4) * 64 + lane] = v;
            }
        __syncthreads();
        const int tok = tb * 32 + tl;
        auto total4 = [&](int rb, int gi) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < NW; ++p) v += red4[((p * RB + rb) * 4 + gi) * 64 + lane];
            return v;
        };
        if constexpr (EPI == EPI_SWIGLU) {
            // wave -> (pair q, register group gi): gate block q, up block q + RB/2
            // (NW = 4, one wave per SIMD: every wave takes both pairs of its register group)
            static_assert(EPI != EPI_SWIGLU || (RB == 4 && (NW == 8 || NW == 4)), "swiglu layout");
            const int gi = wave & 3;
            for (int qq = (NW == 8 ? (wave >> 2) : 0); qq < (NW == 8 ? (wave >> 2) + 1 : 2); ++qq)
            if (8 * gi + 4 * hh < ra.nv[qq]) {
                const f32x4 g4 = total4(qq, gi), u4 = total4(qq + RB / 2, gi);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = 8 * gi + 4 * hh + j;
                    if (f < ra.nv[qq]) {
                        const float gv = bfr(g4[j]), uv = bfr(u4[j]);
                        const float sv = bfr(gv / (1.0f + expf(-gv)));
                        const int feat = ra.R * bp.bx + 32 * qq + f;
                        a.act_xp[(size_t)ex * a.ex_act + xp_offset(tok, feat)] = f2bf(sv * uv);
                    }
                }
            }
        } else if constexpr (EPI == EPI_QKV) {
            // general R (not a multiple of 4): wave -> register group gi, scalar stores
            const int gi = wave & 3;
            const int ps = a.pos[tok];
            if (wave < 4 && 8 * gi + 4 * hh < ra.nv[0]) {
                const f32x4 xl4 = total4(0, gi), xh4 = total4(1, gi);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = 8 * gi + 4 * hh + j;
                    if (f < ra.nv[0]) {
                        const int pr = ra.R * bp.bx + f, slot = pr >> 6, dlo = pr & 63, dhi = dlo + 64;
                        const float xl = xl4[j], xh = xh4[j];
                        if (slot < a.nh + a.nkv) {
                            bf16_t* dst = slot < a.nh ? a.qf + (size_t)slot * 8192 : a.kfresh + (size_t)(slot - a.nh) * 8192;
                            const float c = bf2f(a.rcos[(size_t)ps * 64 + dlo]), sn = bf2f(a.rsin[(size_t)ps * 64 + dlo]);
                            const float bl = bfr(xl), bh = bfr(xh);
                            dst[rf_offset(tok, dlo)] = f2bf(bfr(bl * c) + bfr(-bh * sn));
                            dst[rf_offset(tok, dhi)] = f2bf(bfr(bh * c) + bfr(bl * sn));
                        } else {
                            bf16_t* dst = a.vfresh + (size_t)(slot - a.nh - a.nkv) * 8192;
                            dst[vf_offset(tok, dlo)] = f2bf(xl);
                            dst[vf_offset(tok, dhi)] = f2bf(xh);
                        }
                    }
                }
            }
        } else {
            // EPI_LOGITS: wave -> SL slices (rb, gi); candidates per (workgroup, wave)
#pragma unroll
            for (int sidx = 0; sidx < SL; ++sidx) {
                const int sl = wave * SL + sidx, rb = sl >> 2, gi = sl & 3;
                if (8 * gi + 4 * hh < ra.nv[rb]) {
                    const f32x4 t4 = total4(rb, gi);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = 8 * gi + 4 * hh + j;
                        if (f < ra.nv[rb]) {
                            const bf16_t hv = f2bf(t4[j]);
                            const int idx = ra.R * bp.bx + 32 * rb + f;
                            if (a.logits) a.logits[(size_t)tok * a.N + idx] = hv;
                            const float v = bf2f(hv);
                            if (v > best || (v == best && idx < bidx)) { best = v; bidx = idx; }
                        }
                    }
                }
            }
            float ob = __shfl_xor(best, 32, 64);
            int oi = __shfl_xor(bidx, 32, 64);
            if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
            if (hh == 0) {
                // park the wave's candidate of this token: [wave][64] pairs behind the reduction buffer (lk_gemm64r_logits adds
                // LA_CAND_LDS bytes of dynamic LDS); ONE candidate per (workgroup, token) leaves the kernel (round 2: one per wave)
                float* cl = redr + NW * RB * 16 * 64;
                cl[(wave * LA_TB + tok) * 2] = best;
                ((int*)cl)[(wave * LA_TB + tok) * 2 + 1] = bidx;
            }
            best = -INFINITY; bidx = 0x7fffffff;
        }
    }
    if constexpr (EPI == EPI_LOGITS) {
        __syncthreads();
        if (wave == 0) {
            const float* cl = redr + NW * RB * 16 * 64;
            float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float v = cl[(w * LA_TB + lane) * 2];
                const int i = ((const int*)cl)[(w * LA_TB + lane) * 2 + 1];
                if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
            }
            a.cand_val[(size_t)bp.bx * LA_TB + lane] = bv;
            a.cand_idx[(size_t)bp.bx * LA_TB + lane] = bi;
        }
    }
    if constexpr (EPI == EPI_SWIGLU && NSF == 0) {
        if (pf_on) pf_keep8(pfv);              // the prefetch loads landed under the epilogue; their registers stay reserved until here
    }
    if (stamp && lane == 0) stamp[2] = wall_clock64();
}

template <int RB, int EPI, int D, int NW, int NSF = 0>
__global__ __launch_bounds__(NW * 64) void k_gemm64r(const bf16_t* __restrict__ wp_s, const bf16_t* xp_s,
                                                      const float* __restrict__ route_col_s, int K16_s, int kfl, int wg_chunks_s,
                                                      unsigned nvl_pk, int boff0, int boff1, int boff2, int boff3, GemmRArgs ra) {
    extern __shared__ __attribute__((aligned(16))) float redr[];      // [NW][RB][16][64]
    const BlockPos bp = {(int)blockIdx.x, (int)blockIdx.y, 0, (int)gridDim.x, (int)gridDim.y};
    gemm64r_body<RB, EPI, D, NW, NSF>(wp_s, xp_s, route_col_s, K16_s, kfl, wg_chunks_s, nvl_pk, boff0, boff1, boff2, boff3, ra, bp, redr);
}

// ---------------------------------------------------------------------------------------------
// Role-fused launch gate/up -> down_proj (cfg.fuse bit 2).  One grid: workgroups [0, n_gu) run the balanced gate/up + SwiGLU
// GEMM and publish act with an agent-scope release + counter; workgroups [n_gu, n_gu + n_dn) run the split-K down_proj: they
// are dispatched as gate/up workgroups EXIT (every workgroup of this launch fills a CU), request their first weight tile-sets
// at once — HBM stays busy through gate/up's ragged tail, reduction and epilogue — and wait for the counter before loading
// act.  What it replaces: a kernel boundary + the cold start of the down_proj launch; what it costs: a release / acquire
// pair.  Deadlock-free as long as workgroups are dispatched in block-id order (the down role only ever waits on LOWER ids;
// the spin is bounded and traps).  Same arithmetic in the same order as the two launches: bitwise identical results.
// ---------------------------------------------------------------------------------------------
// NSF = 4 (round 4, cfg.fuse bits 0 + 2 together): the post-attention RMSNorm runs inside the same launch as well — the first 64
// gate/up workgroups first sum o_proj's split-K slabs, add the residual, normalise their row and publish it (FusedNorm) — so the
// whole MLP half of a layer (norm -> gate/up + SwiGLU -> down_proj) is ONE launch with two in-launch hand-overs: the persistent-layer
// prototype the round-3 review asked for, built from the kernels of the step (same arithmetic and order: bitwise identical).
template <int RBD, int DD, int NSF = 0>
__global__ __launch_bounds__(512) void k_gateup_down(const bf16_t* __restrict__ wp_s, const bf16_t* xp_s, int K16_s, int kfl, int wg_chunks_s,
                                                      unsigned nvl_pk, int boff0, int boff1, int boff2, int boff3, int n_gu,
                                                      GemmRArgs gu, GemmArgs dn, int dn_gx, int dn_gy, int* counter) {
    extern __shared__ __attribute__((aligned(16))) float lds_gd[];     // gate/up reduction [8][4][16][64] == down_proj reduction [8][RBD*2*16*64]
    if ((int)blockIdx.x < n_gu) {
        const BlockPos bp = {(int)blockIdx.x, 0, 0, n_gu, 1};
        gemm64r_body<4, EPI_SWIGLU, 4, 8, NSF>(wp_s, xp_s, nullptr, K16_s, kfl, wg_chunks_s, nvl_pk, boff0, boff1, boff2, boff3, gu, bp, lds_gd);
        handover_signal(counter);
    } else {
        const int id = (int)blockIdx.x - n_gu;
        const BlockPos bp = {id % dn_gx, id / dn_gx, 0, dn_gx, dn_gy};
        gemm64_body<RBD, EPI_SLAB, DD, 8, true>(dn.wp, dn.xp, nullptr, dn.K16, 0, dn, bp, lds_gd, counter, n_gu);
    }
}

__global__ __launch_bounds__(256) void k_argmax_finalize(const float* __restrict__ cv, const int* __restrict__ ci,
                                                        int n_tiles, int* __restrict__ out) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;   // one token per block
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int i = threadIdx.x; i < n_tiles; i += 256) {
        float v = cv[(size_t)i * LA_TB + t];
        int idx = ci[(size_t)i * LA_TB + t];
        if (v > best || (v == best && idx < bidx)) { best = v; bidx = idx; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bidx, o, 64);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    if (lane == 0) { sv[wave] = best; si[wave] = bidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bidx)) { best = sv[w]; bidx = si[w]; }
        out[t] = bidx;
    }
}

// MoE accumulation (MixtralSparseMoeBlock.forward :731-756): final[t] (+)= bf16(bf16(expert_out[t]) * w[t][e]) for the
// rows routed to expert e, experts visited in index order (index_add_ into a bf16 buffer).  FIRST zero-fills.
template <int NS, bool FIRST>
__global__ __launch_bounds__(256) void k_moe_accum(const float* __restrict__ slabs, const float* __restrict__ route_col,
                                                    int hidden, bf16_t* __restrict__ acc) {
    const int t = blockIdx.x;
    const float w = route_col[t * LA_MOE_MAX_E];
    if (w == 0.f && !FIRST) return;
    for (int c = threadIdx.x; c < (hidden >> 3); c += 256) {
        bf16x8 cur = {0, 0, 0, 0, 0, 0, 0, 0};
        if (!FIRST) cur = *(const bf16x8*)(acc + (size_t)t * hidden + c * 8);
        if (w != 0.f) {
            float add[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                const float* sp = slabs + ((size_t)s2 * LA_TB + t) * hidden + c * 8;
                const f32x4 a0 = *(const f32x4*)sp, a1 = *(const f32x4*)(sp + 4);
                add[0] += a0[0]; add[1] += a0[1]; add[2] += a0[2]; add[3] += a0[3];
                add[4] += a1[0]; add[5] += a1[1]; add[6] += a1[2]; add[7] += a1[3];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float contrib = bfr(bfr(add[j]) * w);
                cur[j] = (short)f2bf(FIRST ? contrib : bf2f((bf16_t)cur[j]) + contrib);
            }
        }
        *(bf16x8*)(acc + (size_t)t * hidden + c * 8) = cur;
    }
}

// ---------------------------------------------------------------------------------------------
// Step-input expansion: ids / rowmask / positions.  Replaces the mask concat of
// lookahead_prepare_inputs_for_generation (pretrained_model.py:725-734) and the model hook
// position_ids = mask.sum(-1) - 1 (modeling_llama.py:584-588) without materialising [1,1,T,C+T].
// ---------------------------------------------------------------------------------------------
__global__ void k_build_tree_inputs(const int* __restrict__ in, int* __restrict__ state, int* __restrict__ pos,
                                    unsigned long long* __restrict__ rowmask, int* __restrict__ ids) {
    const int t = threadIdx.x;   // 64 threads
    const int T = in[LA_IN_T];
    const unsigned long long* rmin = (const unsigned long long*)(in + LA_IN_ROWMASK);
    unsigned long long rm = (t < T) ? rmin[t] : (1ull << t);
    rowmask[t] = rm;
    ids[t] = (t < T) ? in[LA_IN_IDS + t] : 0;
    pos[t] = state[LA_ST_NKEYS] + __popcll(rm) - 1;
    if (t == 0) { state[LA_ST_T] = T; state[LA_ST_MODE] = in[LA_IN_MODE]; }
}

// Multi-sequence form (pretrained_model_batch.py:706-731 + modeling_llama_batch.py:729-734): the 64 block rows are
// shared by up to LA_MAX_SEQ sequence slots; a row's position counts its own slot's committed keys.
__global__ void k_build_tree_inputs_b(const int* __restrict__ in, int* __restrict__ bstate, int* __restrict__ pos,
                                      unsigned long long* __restrict__ rowmask, int* __restrict__ ids) {
    const int t = threadIdx.x;   // 64 threads
    const int T = in[LA_BIN_T];
    const unsigned long long* rmin = (const unsigned long long*)(in + LA_BIN_ROWMASK);
    int s = (t < T) ? in[LA_BIN_SEQ + t] : -1;
    if (s >= LA_MAX_SEQ) s = -1;
    const unsigned long long rm = (s >= 0) ? rmin[t] : (1ull << t);
    rowmask[t] = rm;
    ids[t] = (s >= 0) ? in[LA_BIN_IDS + t] : 0;
    pos[t] = (s >= 0) ? bstate[LA_BST_NKEYS + s] + __popcll(rm) - 1 : 0;
    bstate[LA_BST_SEQ + t] = s;
}

// ---------------------------------------------------------------------------------------------
// QKV post-processing: slab sum -> bf16, RoPE (rotate-half, bf16 arithmetic as apply_rotary_pos_emb,
// modeling_llama.py:154-169, cos/sin tables cast to bf16 as LlamaRotaryEmbedding.forward :110-126),
// Q -> QF fragments, K/V of the 64 tree tokens -> fresh KF / VF tiles.
// grid = nh + 2*nkv head slots, 256 threads.
// ---------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(256) void k_qkv_post(const float* __restrict__ slabs, int nh, int nkv,
                                                   const int* __restrict__ pos, const bf16_t* __restrict__ rcos,
                                                   const bf16_t* __restrict__ rsin, bf16_t* __restrict__ qf,
                                                   bf16_t* __restrict__ kfresh, bf16_t* __restrict__ vfresh) {
    __shared__ __attribute__((aligned(16))) bf16_t sh[LA_TB][128 + 8];
    const int slot = blockIdx.x;
    const int N = (nh + 2 * nkv) * 128;
    const bool rope = slot < nh + nkv;
    // RoPE operands of this thread's 4 (token, 8-dim piece) items: issued first so that the dependent
    // pos -> cos/sin loads overlap the slab staging below
    bf16x8 cs[4], sn[4];
    if (rope) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = threadIdx.x + j * 256, t = i >> 4, p = i & 15;
            const int ps = pos[t];
            cs[j] = *(const bf16x8*)(rcos + (size_t)ps * 64 + (p & 7) * 8);
            sn[j] = *(const bf16x8*)(rsin + (size_t)ps * 64 + (p & 7) * 8);
        }
    }
    // stage [64][128] of this head slot, summed over slabs and rounded to bf16 (the nn.Linear output dtype)
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = threadIdx.x + j * 256, t = i >> 5, c4 = (i & 31) * 4;
        acc[j] = *(const f32x4*)(slabs + (size_t)t * N + slot * 128 + c4);
#pragma unroll
        for (int sl = 1; sl < NS; ++sl) acc[j] += *(const f32x4*)(slabs + ((size_t)sl * LA_TB + t) * N + slot * 128 + c4);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = threadIdx.x + j * 256, t = i >> 5, c4 = (i & 31) * 4;
        bf16x4 pk = {(short)f2bf(acc[j][0]), (short)f2bf(acc[j][1]), (short)f2bf(acc[j][2]), (short)f2bf(acc[j][3])};
        *(bf16x4*)&sh[t][c4] = pk;
    }
    __syncthreads();
    if (rope) {
        bf16_t* dst = slot < nh ? qf + (size_t)slot * 2 * 8 * 512 : kfresh + (size_t)(slot - nh) * 2 * 8 * 512;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = threadIdx.x + j * 256, t = i >> 4, p = i & 15;
            const int d0 = p * 8, dp = ((p + 8) & 15) * 8;
            const bf16x8 xv = *(const bf16x8*)&sh[t][d0];
            const bf16x8 xq = *(const bf16x8*)&sh[t][dp];
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = bf2f((bf16_t)xv[e]);
                float xr = bf2f((bf16_t)xq[e]);
                if (p < 8) xr = -xr;
                float a = bfr(x * bf2f((bf16_t)cs[j][e]));
                float b = bfr(xr * bf2f((bf16_t)sn[j][e]));
                o[e] = (short)f2bf(a + b);
            }
            *(bf16x8*)(dst + rf_offset(t, d0)) = o;
        }
    } else {
        bf16_t* dst = vfresh + (size_t)(slot - nh - nkv) * 2 * 8 * 512;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = threadIdx.x + j * 256;
            const int ln = i & 63, s2 = (i >> 6) & 1, db = (i >> 7) & 3, tb = i >> 9;
            const int d = db * 32 + (ln & 31), hh = ln >> 5;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int key = tb * 32 + 16 * s2 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                o[e] = (short)sh[key][d];
            }
            *(bf16x8*)(dst + ((size_t)(tb * 8 + db * 2 + s2) * 512 + ln * 8)) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Tree attention (LlamaAttention.forward, modeling_llama.py:270-296, under the rank-4 mask hook).
//   S^T = K.Q^T (swapped so a lane owns one token column: softmax reductions stay in-lane + one
//   xor-32), prefix keys mask-free from the packed main cache, the 64 fresh keys under the 64-bit
//   ancestor row mask; online softmax in fp32; O^T += V^T.P^T with P^T taken from the S^T
//   accumulator registers in place (VF key order).  grid = (heads, key splits), 4 waves:
//   wave = (token block, key-tile parity).
// ---------------------------------------------------------------------------------------------
struct AttnArgs {
    const bf16_t* qf;
    const bf16_t* kmain;
    const bf16_t* vmain;
    const bf16_t* kfresh;
    const bf16_t* vfresh;
    const unsigned long long* rowmask;
    const int* state;
    int nh, nkv, max_keys, nsplit;
    float qk;       // la_qk_scale(head_dim): softmax scale of the build (attn_scale, la_common.h)
    float* opart;   // [nh][nsplit][64][128]
    float* mpart;   // [nh][nsplit][64]
    float* lpart;
    // multi-sequence mode (seq != nullptr): grid.z = slot; a row only sees the committed keys of its own slot
    const int* seq;       // [64] row -> slot (-1 = unused row)
    const int* nkeys_b;   // [LA_MAX_SEQ] committed keys per slot
    int slot_tiles;       // 32-key tiles per slot region of the main cache
    int window;           // > 0: sliding-window attention, a row at position p sees committed keys j with p - j <= window
    long long* dbg_times; // measurement aid (scripts/gpu_ab.py): [workgroup][wave][8] wall-clock stamps, null in production
    int ring_tiles;       // > 0: the main cache of a sequence is a RING of this many 32-key tiles (position p lives in row p mod
                          // ring size): memory O(window) for sliding-window models; needs window + tree rows <= ring size
};

#define LA_NEG (-1.0e30f)

#define LA_ATT_PAR 4      // key-tile parities per workgroup (waves = 2 token blocks x LA_ATT_PAR)
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// ST = true ("staged"): the K and V tiles of a key step are copied ONCE per workgroup into LDS by LDS-DMA
// (global_load_lds_dwordx4, no staging registers) and read by both token-block waves of a key parity; the direct form loads
// every tile into the registers of BOTH waves, i.e. each K/V byte crosses the CU's vector-memory path twice — at long contexts
// that path (about 22 GB/s per CU) is the bound.  Two stages of LA_ATT_PAR tiles x 16 KiB are in flight per workgroup
// (the ring aliases the merge buffer).  Same arithmetic, same order of operations: bit-identical results.
// Leading scalars (kernarg preload, see k_gemm64): what stands between dispatch and the first K-tile request — the q / rowmask /
// cursor pointers and the cache geometry; `a` carries the same values once more plus everything the later phases use.
template <bool ST>
__global__ __launch_bounds__(2 * LA_ATT_PAR * 64) void k_tree_attn(const bf16_t* __restrict__ qf_s, const unsigned long long* __restrict__ rowmask_s,
                                                                    const int* __restrict__ state_s, const int* __restrict__ seq_s,
                                                                    const bf16_t* __restrict__ kmain_s, const bf16_t* __restrict__ vmain_s,
                                                                    int max_keys_s, int nsplit_s, int nh_nkv, int window_s, AttnArgs a0) {
    extern __shared__ __attribute__((aligned(16))) float mgbuf[];   // [LA_ATT_PAR][2][66][64] merge buffer (132 KiB)
    AttnArgs a = a0;
    a.qf = qf_s; a.rowmask = rowmask_s; a.state = state_s; a.seq = seq_s; a.kmain = kmain_s; a.vmain = vmain_s;
    a.max_keys = max_keys_s; a.nsplit = nsplit_s; a.nh = nh_nkv >> 16; a.nkv = nh_nkv & 0xffff; a.window = window_s;
    // GQA: head id from the block id so that the query heads of one kv head share an XCD (= an L2): see k_tree_attn_mb
    const int h = ((int)blockIdx.x % a.nkv) * (a.nh / a.nkv) + (int)blockIdx.x / a.nkv, sp = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tb = wave & 1, par = wave >> 1;      // par in [0, LA_ATT_PAR)
    long long* const stamp = a.dbg_times ? a.dbg_times + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (2 * LA_ATT_PAR) + wave) * 8 : nullptr;
    if (stamp && lane == 0) stamp[0] = wall_clock64();
    const int hk = h / (a.nh / a.nkv);
    const int KB = a.max_keys >> 5;
    int nkeys, tile0 = 0;
    bool mine = true;
    unsigned long long rm = a.rowmask[tb * 32 + (lane & 31)];
    if (a.seq) {
        const int slot = blockIdx.z;
        if (__ballot(a.seq[lane] == slot) == 0ull) return;          // no row of this slot in the block (uniform)
        nkeys = a.nkeys_b[slot];
        tile0 = slot * a.slot_tiles;
        mine = a.seq[tb * 32 + (lane & 31)] == slot;
        if (!mine) rm = 0ull;
    } else {
        nkeys = a.state[LA_ST_NKEYS];
    }
    // sliding window (HF Mistral mask rule, modeling_attn_mask_utils: visible iff pos_row - j <= window): tiles wholly
    // below the root's horizon are skipped for every row, the rest is masked per row
    const int NPall = (nkeys + 31) >> 5;
    const int ts = (a.window > 0 && nkeys - a.window > 0) ? ((nkeys - a.window) >> 5) : 0;
    const int NP = NPall - ts, NT = NP + 2;
    const int tbase = tile0;                  // first tile of the sequence's main-cache region
    tile0 += ts;
    const int key_lo = (a.window > 0) ? nkeys + __popcll(a.rowmask[tb * 32 + (lane & 31)]) - 1 - a.window : 0;
    const int i0 = (NT * sp) / a.nsplit;
    // a wave whose token block holds no row of the slot contributes nothing: skip its tiles (it still joins the merge)
    const int i1 = (__ballot(mine) == 0ull) ? i0 : (NT * (sp + 1)) / a.nsplit;
    const int nk_row = mine ? nkeys : 0;

    bf16x8 q[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) q[s] = *((const bf16x8*)(a.qf + ((size_t)(h * 2 + tb) * 8 + s) * 512) + lane);
    const int hh = lane >> 5;

    f32x16 o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    float m = LA_NEG, l = 0.f;

    // main-cache tile of logical (position) tile ts + it: identity, or its ring slot
    auto mtile = [&](int it) -> size_t {
        return (size_t)(a.ring_tiles > 0 ? tbase + (ts + it) % a.ring_tiles : tile0 + it);
    };
    auto kptr = [&](int it) -> const bf16x8* {
        return it >= NP ? (const bf16x8*)(a.kfresh + ((size_t)hk * 2 + (it - NP)) * 4096)
                        : (const bf16x8*)(a.kmain + ((size_t)hk * KB + mtile(it)) * 4096);
    };
    auto vptr = [&](int it) -> const bf16x8* {
        return it >= NP ? (const bf16x8*)(a.vfresh + ((size_t)hk * 2 + (it - NP)) * 4096)
                        : (const bf16x8*)(a.vmain + ((size_t)hk * KB + mtile(it)) * 4096);
    };
    // one key tile: S^T = K.Q^T, mask, online softmax, O^T += V^T.P^T.  The V fragments are requested before the
    // QK^T MFMAs and consumed after the softmax; the NEXT tile's K fragments are requested by the caller first.
    auto tile = [&](int it, const bf16x8 (&kf)[8], unsigned vlds) {
        const bool fresh = it >= NP;
        const int kb = fresh ? it - NP : it;
        bf16x8 vf[8];
        if constexpr (ST) {
            // V fragments of the staged tile: LDS reads issued before the QK^T MFMAs, awaited before the PV MFMAs
#pragma unroll
            for (int s = 0; s < 8; ++s) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vf[s]) : "v"(vlds), "n"(s * 1024) : "memory");
        } else {
            const bf16x8* vt = vptr(it);
#pragma unroll
            for (int s = 0; s < 8; ++s) vf[s] = vt[s * 64 + lane];
        }
        f32x16 sc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) sc = LA_MFMA(kf[s], q[s], sc, 0, 0, 0);
        // attn_weights = bf16(QK^T) / sqrt(head_dim) -> bf16 (modeling_llama.py:270), masked keys excluded
        float mx = LA_NEG;
        // a committed tile every row sees whole (no window, one sequence, all 32 keys below nkeys) needs no mask arithmetic:
        // the common case — all but the last committed tile and the two fresh ones (wave-uniform)
        const bool whole = !fresh && a.window <= 0 && !a.seq && (ts + kb) * 32 + 31 < nkeys;
        if (whole) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = attn_scale(sc[i], a.qk);
                sc[i] = v;
                mx = fmaxf(mx, v);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kk = (i & 3) + 8 * (i >> 2) + 4 * hh;
                // bf16(x / sqrt(128)) == bf16(x * fp32(1 / sqrt(128))) for EVERY finite bf16 x (checked exhaustively over the 65536 bit
                // patterns, tests/test_oracle_llama.py::test_attention_scale_as_multiply_is_exact): one multiply instead of an IEEE division
                float v = attn_scale(sc[i], a.qk);
                const int kidx = (ts + kb) * 32 + kk;      // committed keys: absolute index = position
                const bool ok = fresh ? ((rm >> (kb * 32 + kk)) & 1ull) != 0ull : (kidx < nk_row && kidx >= key_lo);
                v = ok ? v : LA_NEG;
                sc[i] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float alpha = __expf(m - mn);
        float ps = 0.f;
        bf16x8 pf[2];
        if (whole) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __expf(sc[i] - mn);
                ps += p;
                pf[i >> 3][i & 7] = (short)f2bf(p);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float p = (sc[i] > -1.0e29f) ? __expf(sc[i] - mn) : 0.f;
                ps += p;
                pf[i >> 3][i & 7] = (short)f2bf(p);
            }
        }
        ps += __shfl_xor(ps, 32, 64);
        l = l * alpha + ps;
        m = mn;
        if (__ballot(alpha != 1.0f) != 0ull) {          // the running maxima moved for some row: rescale (x * 1.0f is exact, so skipping is too)
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
        }
        if constexpr (ST)                                                         // the V fragments have arrived (in/out operands, see K)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]), "+v"(vf[4]), "+v"(vf[5]),
                         "+v"(vf[6]), "+v"(vf[7]) :: "memory");
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            o[db] = LA_MFMA(vf[db * 2 + 0], pf[0], o[db], 0, 0, 0);
            o[db] = LA_MFMA(vf[db * 2 + 1], pf[1], o[db], 0, 0, 0);
        }
    };
    if constexpr (ST) {
        // ---- staged K/V: stage s = the tiles i0 + 4 s + par (par = 0..3) in ring slot s & 1; wave (tb, par) copies K (tb = 0) or V
        //      (tb = 1) of its parity's tile as 8 pieces of 1 KiB.  The tile range is the workgroup's (a wave whose token block holds
        //      no row of the slot still copies for its partner); a tile past the range re-reads the first one and is never consumed,
        //      so every wave issues exactly 8 pieces per stage and the vmcnt arithmetic is static.
        char* ring = (char*)mgbuf;
        const unsigned ring0 = (unsigned)(size_t)(lptr_t)ring;
        const int i1_all = (NT * (sp + 1)) / a.nsplit;
        const int nst = (i1_all - i0 + LA_ATT_PAR - 1) / LA_ATT_PAR;
        auto issue = [&](int st) {
            int it = i0 + st * LA_ATT_PAR + par;
            it = it < i1_all ? it : i0;
            const bf16x8* src = (tb == 0 ? kptr(it) : vptr(it)) + lane;
            char* dst = ring + (st & 1) * 65536 + par * 16384 + tb * 8192;
#pragma unroll
            for (int pc = 0; pc < 8; ++pc)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + pc * 64), (lptr_t)(dst + pc * 1024), 16, 0, 0);
        };
        // Two stages are ALWAYS issued (a workgroup with fewer tiles re-reads its first one): the number of copies in flight is
        // then static, so the compiler's own wait for q below counts them instead of draining them.
        issue(0);
        issue(1);
        // q (ordinary loads, requested before the copies) becomes a consumed value here: no load of the compiler's own stays
        // pending across the stage loop
#pragma unroll
        for (int s = 0; s < 8; ++s) asm volatile("" :: "v"(q[s]));
        if (stamp && lane == 0) stamp[1] = wall_clock64();
        for (int st = 0; st < nst; ++st) {
            if (st == 0 || st + 1 < nst) vm_wait<8>(); else vm_wait<0>();   // own pieces of stage st have landed (stage st + 1 may fly)
            __builtin_amdgcn_s_barrier();                                    // ... and everybody else's
            const int it = i0 + st * LA_ATT_PAR + par;
            if (it < i1) {
                const unsigned kl = ring0 + (unsigned)((st & 1) * 65536 + par * 16384) + (unsigned)lane * 16u;
                bf16x8 kf[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[s]) : "v"(kl), "n"(s * 1024) : "memory");
                // the fragments are in/out operands of the wait: nothing that reads them can be scheduled above it
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]), "+v"(kf[5]),
                             "+v"(kf[6]), "+v"(kf[7]) :: "memory");
                tile(it, kf, kl + 8192u);
                if (stamp && lane == 0 && st == 0) stamp[2] = wall_clock64();
            }
            __builtin_amdgcn_s_barrier();                                    // slot st & 1 is free again
            if (st + 2 < nst) issue(st + 2);
        }
        vm_wait<0>();                                                        // nst < 2: the unused copies; the ring becomes the merge buffer
        __builtin_amdgcn_s_barrier();
    } else {
        bf16x8 kA[8], kB[8];
        int it = i0 + par;
        if (it < i1) {
            const bf16x8* kt = kptr(it);
#pragma unroll
            for (int s = 0; s < 8; ++s) kA[s] = kt[s * 64 + lane];
        }
        if (stamp && lane == 0) stamp[1] = wall_clock64();
        while (it < i1) {
            int nx = it + LA_ATT_PAR;
            if (nx < i1) {
                const bf16x8* kt = kptr(nx);
#pragma unroll
                for (int s = 0; s < 8; ++s) kB[s] = kt[s * 64 + lane];
            }
            tile(it, kA, 0u);
            if (stamp && lane == 0 && it < i0 + LA_ATT_PAR) stamp[2] = wall_clock64();
            it = nx;
            if (it >= i1) break;
            nx = it + LA_ATT_PAR;
            if (nx < i1) {
                const bf16x8* kt = kptr(nx);
#pragma unroll
                for (int s = 0; s < 8; ++s) kA[s] = kt[s * 64 + lane];
            }
            tile(it, kB, 0u);
            it = nx;
        }
    }

    // Merge the key-parity waves of each token block in ONE LDS round: every wave parks (O, m, l), one barrier, then wave
    // (tb, par) finishes head-dim block db = par of its token block (fixed order p = 0..LA_ATT_PAR-1 -> deterministic) and
    // stores that quarter of the split partial itself: all 8 waves take part in the merge and in the store.
    static_assert(LA_ATT_PAR == 4, "one head-dim block (of 4) per key-parity wave");
    if (stamp && lane == 0) stamp[3] = wall_clock64();
    {
        float* mg = mgbuf + (size_t)((par * 2 + tb) * 66) * 64;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) mg[(db * 16 + i) * 64 + lane] = o[db][i];
        mg[64 * 64 + lane] = m;
        mg[65 * 64 + lane] = l;
    }
    __syncthreads();
    float mp[LA_ATT_PAR], wp[LA_ATT_PAR];
    float M = LA_NEG, L = 0.f;
#pragma unroll
    for (int p = 0; p < LA_ATT_PAR; ++p) {
        mp[p] = mgbuf[(size_t)(((p * 2 + tb) * 66) + 64) * 64 + lane];
        M = fmaxf(M, mp[p]);
    }
#pragma unroll
    for (int p = 0; p < LA_ATT_PAR; ++p) {
        wp[p] = __expf(mp[p] - M);
        L += mgbuf[(size_t)(((p * 2 + tb) * 66) + 65) * 64 + lane] * wp[p];
    }
    float od[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float v = 0.f;
#pragma unroll
        for (int p = 0; p < LA_ATT_PAR; ++p) v += mgbuf[(size_t)(((p * 2 + tb) * 66) + par * 16 + i) * 64 + lane] * wp[p];
        od[i] = v;
    }
    if (stamp && lane == 0) stamp[4] = wall_clock64();
    if (!mine) return;
    const int tok = tb * 32 + (lane & 31);
    float* op = a.opart + (((size_t)h * a.nsplit + sp) * LA_TB + tok) * 128;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v = {od[4 * g], od[4 * g + 1], od[4 * g + 2], od[4 * g + 3]};
        *(f32x4*)(op + par * 32 + 8 * g + 4 * hh) = v;
    }
    if (hh == 0 && par == 0) {
        a.mpart[((size_t)h * a.nsplit + sp) * LA_TB + tok] = M;
        a.lpart[((size_t)h * a.nsplit + sp) * LA_TB + tok] = L;
    }
    if (stamp && lane == 0) stamp[5] = wall_clock64();
}

// merge key splits, normalise, round to bf16 (attn_output dtype) and emit the packed operand of o_proj
template <int NS>
__global__ __launch_bounds__(256) void k_attn_combine(const float* __restrict__ opart, const float* __restrict__ mpart,
                                                       const float* __restrict__ lpart, int nh,
                                                       const int* __restrict__ seq, bf16_t* __restrict__ attn_xp,
                                                       int n_main, PfDesc pf) {
    if ((int)blockIdx.x >= n_main) { pf_body(pf, (int)blockIdx.x - n_main); return; }   // appended idle-window prefetch workgroups
    int gid = blockIdx.x * 256 + threadIdx.x;   // (h, tok, d8)
    if (gid >= nh * LA_TB * 16) return;
    int d8 = gid & 15, tok = (gid >> 4) & 63, h = gid >> 10;
    if (seq && seq[tok] < 0) {                   // multi-sequence mode: rows no slot owns carry zeros
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        *(bf16x8*)(attn_xp + xp_offset(tok, h * 128 + d8 * 8)) = z;
        return;
    }
    float ms[NS], ls[NS];
    f32x4 o0[NS], o1[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        ms[s] = mpart[((size_t)h * NS + s) * LA_TB + tok];
        ls[s] = lpart[((size_t)h * NS + s) * LA_TB + tok];
        const float* op = opart + (((size_t)h * NS + s) * LA_TB + tok) * 128 + d8 * 8;
        o0[s] = *(const f32x4*)op;
        o1[s] = *(const f32x4*)(op + 4);
    }
    float M = LA_NEG;
#pragma unroll
    for (int s = 0; s < NS; ++s) M = fmaxf(M, ms[s]);
    float L = 0.f, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float w = __expf(ms[s] - M);
        L += w * ls[s];
        acc[0] += w * o0[s][0]; acc[1] += w * o0[s][1]; acc[2] += w * o0[s][2]; acc[3] += w * o0[s][3];
        acc[4] += w * o1[s][0]; acc[5] += w * o1[s][1]; acc[6] += w * o1[s][2]; acc[7] += w * o1[s][3];
    }
    const float inv = 1.0f / L;
    bf16x8 ov;
#pragma unroll
    for (int j = 0; j < 8; ++j) ov[j] = (short)f2bf(acc[j] * inv);
    *(bf16x8*)(attn_xp + xp_offset(tok, h * 128 + d8 * 8)) = ov;
}

// ---------------------------------------------------------------------------------------------
// Accept scan (one wavefront).  _lookahead_update_model_kwargs_for_generation, pretrained_model.py:806-880
// with an empty logits-processor list: walk from the root, at each accepted node take the child whose
// draft token equals that node's argmax (lowest row index = the reference's first surviving branch);
// emitted tokens = argmax along the path (matches + 1 bonus).  mode 1 (prefill chain): commit all T rows
// and emit argmax of the last row (pretrained_model.py:783-798).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void accept_walk(const int j, const int am, const int* __restrict__ ids,
                                            const unsigned long long* __restrict__ rowmask, int* __restrict__ state) {
    const int T = state[LA_ST_T], mode = state[LA_ST_MODE];
    const int nkeys = state[LA_ST_NKEYS];
    int n_commit;
    if (mode == 2) {            // verify only: the host walks the tree (sequential logits processors) and commits later
        if (j == 0) state[LA_ST_NOUT] = 0;
        n_commit = 0;
    } else if (mode == 1) {
        if (j < T) state[LA_ST_SRCIDX + j] = j;
        const int last = __shfl(am, T - 1, 64);   // executed by all lanes
        if (j == 0) { state[LA_ST_OUTTOK] = last; state[LA_ST_NOUT] = 1; }
        n_commit = T;
    } else {
        const unsigned long long rm = rowmask[j];
        const unsigned long long below = rm & ((1ull << j) - 1ull);
        const int parent = (j == 0 || below == 0ull) ? -1 : 63 - __clzll(below);
        const int myid = ids[j];
        // live = the rows whose root path spells the tokens accepted so far.  A hier tree has one such row per depth (trie
        // children carry distinct tokens), i.e. "the child of cur"; a par layout (lookahead_cache.py:441-488) repeats a shared
        // prefix in every chain, all copies stay live, and the lowest row — the reference's first surviving leaf branch,
        // mask_indices[0] at pretrained_model.py:831 — supplies the next logits row and the kept K/V row.
        int cur = 0, depth = 0;
        unsigned long long live = 1ull;
        if (j == 0) { state[LA_ST_SRCIDX] = 0; }
        while (true) {
            const int want = __shfl(am, cur, 64);
            if (j == 0) state[LA_ST_OUTTOK + depth] = want;
            const unsigned long long cand = __ballot(j < T && j > 0 && parent >= 0 && ((live >> parent) & 1ull) != 0ull && myid == want);
            if (cand == 0ull) break;
            cur = __ffsll((long long)cand) - 1;
            live = cand;
            ++depth;
            if (j == 0) state[LA_ST_SRCIDX + depth] = cur;
        }
        if (j == 0) state[LA_ST_NOUT] = depth + 1;
        n_commit = depth + 1;
    }
    if (j == 0) {
        state[LA_ST_DSTBASE] = nkeys;
        state[LA_ST_NCOMMIT] = n_commit;
        state[LA_ST_NKEYS] = nkeys + n_commit;
    }
}

// Step tail (single-sequence step): k_argmax_finalize + k_accept_scan + k_publish in ONE launch of one workgroup.
// 1024 threads = 64 tokens x 16 candidate strides; the per-token winners meet in LDS (lowest vocabulary index wins ties, as
// torch.argmax), wave 0 walks the tree (accept_walk: the body of k_accept_scan) and the result block goes to the caller's
// pinned host block BEFORE the KV commit kernel runs: the host's trie update / next query overlap the commit.
__device__ __forceinline__ void accept_walk(const int j, const int am, const int* __restrict__ ids,
                                            const unsigned long long* __restrict__ rowmask, int* __restrict__ state);
__global__ __launch_bounds__(1024) void k_step_tail(const float* __restrict__ cv, const int* __restrict__ ci, int n_tiles,
                                                     const int* __restrict__ ids, const unsigned long long* __restrict__ rowmask,
                                                     int* __restrict__ state, int* __restrict__ host_out) {
    __shared__ float sv[16][64];
    __shared__ int si[16][64];
    const int tok = threadIdx.x & 63, part = threadIdx.x >> 6;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    // 16 candidates per thread and round, ALL requested before the first compare: one memory round trip for the 256 lm_head
    // workgroups' candidates instead of one per candidate (a rolled loop waits for each pair: 16 dependent round trips, ~10 us)
    for (int base = 0; base < n_tiles; base += 256) {
        float vv[16];
        int ii[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = base + part + 16 * k;
            const int ic = i < n_tiles ? i : n_tiles - 1;                 // clamped re-read of the last tile: never wins a strict compare
            vv[k] = cv[(size_t)ic * LA_TB + tok];
            ii[k] = ci[(size_t)ic * LA_TB + tok];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (vv[k] > best || (vv[k] == best && ii[k] < bidx)) { best = vv[k]; bidx = ii[k]; }
    }
    sv[part][tok] = best; si[part][tok] = bidx;
    __syncthreads();
    if (part == 0) {
#pragma unroll
        for (int p = 1; p < 16; ++p) {
            const float v = sv[p][tok]; const int idx = si[p][tok];
            if (v > best || (v == best && idx < bidx)) { best = v; bidx = idx; }
        }
        state[LA_ST_ARGMAX + tok] = bidx;
        accept_walk(tok, bidx, ids, rowmask, state);
    }
    if (!host_out) return;
    __threadfence();                    // wave 0's state words are visible to the other waves of the block after the barrier
    __syncthreads();
    const int j = threadIdx.x;
    int seq = 0;
    if (j == 0) { seq = state[LA_ST_SEQ] + 1; state[LA_ST_SEQ] = seq; }
    if (j < LA_ST_OUTTOK + 64 && j != LA_ST_SEQ) host_out[j] = __hip_atomic_load(state + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __syncthreads();
    if (j == 0) __hip_atomic_store(host_out + LA_ST_SEQ, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_accept_scan(const int* __restrict__ ids, const unsigned long long* __restrict__ rowmask,
                              int* __restrict__ state) {
    accept_walk((int)threadIdx.x, state[LA_ST_ARGMAX + threadIdx.x], ids, rowmask, state);      // 64 threads
}

// ---------------------------------------------------------------------------------------------
// KV commit: accepted fresh rows -> main cache rows DSTBASE.. for every layer and kv head.
// Replaces _update_cache_with_axis_2 (pretrained_model.py:894-907); moves only the accepted rows
// (<= branch_length+1) instead of re-materialising the cache.  grid = layers*nkv, 256 threads.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_kv_commit(const bf16_t* __restrict__ kfresh, const bf16_t* __restrict__ vfresh,
                                                    bf16_t* __restrict__ kmain, bf16_t* __restrict__ vmain,
                                                    const int* __restrict__ state, int nkv, int max_keys, int ring) {
    const int lh = blockIdx.x;               // layer * nkv + kv head
    const int n = state[LA_ST_NCOMMIT], base = state[LA_ST_DSTBASE];
    const size_t KB = (size_t)(max_keys >> 5);
    const bf16_t* kf = kfresh + (size_t)lh * 2 * 4096;
    const bf16_t* vf = vfresh + (size_t)lh * 2 * 4096;
    bf16_t* km = kmain + (size_t)lh * KB * 4096;
    bf16_t* vm = vmain + (size_t)lh * KB * 4096;
    for (int i = threadIdx.x; i < n * 16; i += 256) {
        int r = i >> 4, p = i & 15;
        int src = state[LA_ST_SRCIDX + r], dst = ring ? (base + r) % ring : base + r;
        if (dst >= max_keys) continue;
        *(bf16x8*)(km + rf_offset(dst, p * 8)) = *(const bf16x8*)(kf + rf_offset(src, p * 8));
    }
    for (int i = threadIdx.x; i < n * 128; i += 256) {
        int r = i >> 7, d = i & 127;
        int src = state[LA_ST_SRCIDX + r], dst = ring ? (base + r) % ring : base + r;
        if (dst >= max_keys) continue;
        vm[vf_offset(dst, d)] = vf[vf_offset(src, d)];
    }
}

// Zero-copy result hand-over: the last kernel of the captured step copies the first LA_ST_OUTTOK+64 state words into the
// caller's PINNED host block and then bumps the sequence word (LA_ST_SEQ) there; the host polls that word instead of
// queueing a D2H copy kernel and sleeping on the stream (measured: 22 us graph-end -> copy gap + ~50 us wake-up).
__global__ void k_publish(int* __restrict__ state, int* __restrict__ host_out) {
    const int j = threadIdx.x;      // 128 threads
    int seq = 0;
    if (j == 0) { seq = state[LA_ST_SEQ] + 1; state[LA_ST_SEQ] = seq; }
    if (j < LA_ST_OUTTOK + 64 && j != LA_ST_SEQ) host_out[j] = state[j];
    __threadfence_system();
    __syncthreads();
    if (j == 0) {
        __hip_atomic_store(host_out + LA_ST_SEQ, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------------------------
// Multi-sequence accept scan + commit plan (pretrained_model_batch.py:814-905).  One wavefront per slot: the same
// walk as k_accept_scan over the rows the slot owns, at most LIMIT[slot] emitted tokens (the loop bound
// min(max_branch_length, input_length-cur-2)+1, :862); DST[row] = absolute main-cache key row of every row that is
// kept (root + accepted drafts, or all rows of a prefill chain), -1 otherwise.
// ---------------------------------------------------------------------------------------------
__global__ void k_accept_scan_b(const int* __restrict__ in, const int* __restrict__ ids,
                                const unsigned long long* __restrict__ rowmask, int* __restrict__ bstate, int slot_keys, int ring) {
    const int s = blockIdx.x, j = threadIdx.x;   // 64 threads
    const int sq = bstate[LA_BST_SEQ + j];
    const bool mine = sq == s;
    const unsigned long long own = __ballot(mine);
    if (s == 0 && sq < 0) bstate[LA_BST_DST + j] = -1;
    if (own == 0ull) { if (j == 0) bstate[LA_BST_NOUT + s] = 0; return; }
    const int root = __ffsll((long long)own) - 1;
    const int mode = in[LA_BIN_MODE + s];
    int limit = in[LA_BIN_LIMIT + s];
    limit = limit < 1 ? 1 : (limit > 16 ? 16 : limit);
    const int nkeys = bstate[LA_BST_NKEYS + s];
    const int sbase = s * slot_keys;
    auto row_of = [&](int k) { return sbase + (ring ? (nkeys + k) % slot_keys : nkeys + k); };      // main-cache row of the k-th kept key
    const int am = bstate[LA_BST_ARGMAX + j];
    int dst = -1, n_commit;
    if (mode == 2) {
        // forward only (sequential accept path, pretrained_model_batch.py:814-875 with a non-empty processor list): the host
        // walks the tree over the logits rows and decides the commit (la_llama_bcommit); nothing moves here
        if (j == 0) bstate[LA_BST_NOUT + s] = 0;
        n_commit = 0;
    } else if (mode == 1) {
        const int last = 63 - __clzll((long long)own);
        const int tok = __shfl(am, last, 64);
        if (mine) dst = row_of(__popcll(own & ((1ull << j) - 1ull)));
        if (j == 0) { bstate[LA_BST_OUTTOK + s * 16] = tok; bstate[LA_BST_NOUT + s] = 1; }
        n_commit = __popcll(own);
    } else {
        const unsigned long long below = rowmask[j] & ((1ull << j) - 1ull);
        const int parent = (!mine || j == root || below == 0ull) ? -1 : 63 - __clzll((long long)below);
        const int myid = ids[j];
        int cur = root, depth = 0;
        if (j == root) dst = row_of(0);
        while (true) {
            const int want = __shfl(am, cur, 64);
            if (j == 0) bstate[LA_BST_OUTTOK + s * 16 + depth] = want;
            if (depth + 1 >= limit) break;
            const unsigned long long cand = __ballot(mine && j != root && parent == cur && myid == want);
            if (cand == 0ull) break;
            cur = __ffsll((long long)cand) - 1;
            ++depth;
            if (j == cur) dst = row_of(depth);
        }
        if (j == 0) bstate[LA_BST_NOUT + s] = depth + 1;
        n_commit = depth + 1;
    }
    if (mine) bstate[LA_BST_DST + j] = dst;
    if (j == 0) bstate[LA_BST_NKEYS + s] = nkeys + n_commit;
}

// fresh rows -> their DST rows of the main cache (in-place cursor writes of modeling_llama_batch.py:384-400 and the
// row moves of pretrained_model_batch.py:986-989 in one pass).  grid = layers*nkv, 256 threads.
__global__ __launch_bounds__(256) void k_kv_commit_b(const bf16_t* __restrict__ kfresh, const bf16_t* __restrict__ vfresh,
                                                      bf16_t* __restrict__ kmain, bf16_t* __restrict__ vmain,
                                                      const int* __restrict__ bstate, int total_keys) {
    const int lh = blockIdx.x;
    const size_t KB = (size_t)(total_keys >> 5);
    const bf16_t* kf = kfresh + (size_t)lh * 2 * 4096;
    const bf16_t* vf = vfresh + (size_t)lh * 2 * 4096;
    bf16_t* km = kmain + (size_t)lh * KB * 4096;
    bf16_t* vm = vmain + (size_t)lh * KB * 4096;
    for (int i = threadIdx.x; i < LA_TB * 16; i += 256) {
        const int r = i >> 4, p = i & 15;
        const int dst = bstate[LA_BST_DST + r];
        if (dst < 0 || dst >= total_keys) continue;
        *(bf16x8*)(km + rf_offset(dst, p * 8)) = *(const bf16x8*)(kf + rf_offset(r, p * 8));
    }
    for (int i = threadIdx.x; i < LA_TB * 128; i += 256) {
        const int r = i >> 7, d = i & 127;
        const int dst = bstate[LA_BST_DST + r];
        if (dst < 0 || dst >= total_keys) continue;
        vm[vf_offset(dst, d)] = vf[vf_offset(r, d)];
    }
}

// all experts of a layer in one launch: same arithmetic and the same expert order as k_moe_accum called E times
template <int NS>
__global__ __launch_bounds__(256) void k_moe_accum_all(const float* __restrict__ slabs, long ex_slab, const float* __restrict__ route_w,
                                                        int n_experts, int hidden, bf16_t* __restrict__ acc) {
    const int t = blockIdx.x;
    // the (<= 4) experts this row is routed to, in index order (block-uniform)
    int sel[4];
    float wsel[4];
    int ns = 0;
    for (int e = 0; e < n_experts && ns < 4; ++e) {
        const float w = route_w[t * LA_MOE_MAX_E + e];
        if (w != 0.f) { sel[ns] = e; wsel[ns] = w; ++ns; }
    }
    for (int k = ns; k < 4; ++k) { sel[k] = 0; wsel[k] = 0.f; }
    for (int c = threadIdx.x; c < (hidden >> 3); c += 256) {
        f32x4 v[4][NS][2];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < ns) {
#pragma unroll
                for (int s2 = 0; s2 < NS; ++s2) {
                    const float* sp = slabs + (size_t)sel[k] * ex_slab + ((size_t)s2 * LA_TB + t) * hidden + c * 8;
                    v[k][s2][0] = *(const f32x4*)sp;
                    v[k][s2][1] = *(const f32x4*)(sp + 4);
                }
            }
        float cur[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < ns) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float add = 0.f;
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2) add += v[k][s2][j >> 2][j & 3];
                    const float contrib = bfr(bfr(add) * wsel[k]);
                    cur[j] = k ? bfr(cur[j] + contrib) : contrib;
                }
            }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(cur[j]);
        *(bf16x8*)(acc + (size_t)t * hidden + c * 8) = o;
    }
}

// =============================================================================================
// launchers
// =============================================================================================
// measurement knobs (la_debug_set, scripts/gpu_ab.py); 0 in production
int g_la_graph_epoch = 0;     // bumped by la_debug_set when a capture-time knob changes: la_llama_step captures its graph again
// forked weight prefetch (round 6; la_lab_set keys 26-30, KiB per consumer workgroup, 0 = off): [0] o_proj under the attention launch,
// [1] gate/up queued behind it (runs under attention / o_proj / norm), [2] gate/up forked after o_proj (under the post-attention norm),
// [3] next QKV / lm_head forked after down_proj (under the input norm), [4] down_proj forked after o_proj

#define LA_CAND_LDS (8 * LA_TB * 8)      // lm_head: [8 waves][64 tokens] (value, index) candidates behind the reduction buffer
// leading scalar kernel arguments of the GEMM kernels (kernarg preload, see k_gemm64 / k_gemm64r)
#define G64_HEAD(a) (a).wp, (a).xp, (a).route_col, (a).K16, (((a).ex_on ? 1 : 0) | (((a).kskew & 127) << 1) | ((a).prio_hi << 8))
#define G64R_HEAD(ra) G64_HEAD((ra).g), (ra).wg_chunks, \
    ((unsigned)(ra).nvl[0] | ((unsigned)(ra).nvl[1] << 8) | ((unsigned)(ra).nvl[2] << 16) | ((unsigned)(ra).nvl[3] << 24)), \
    (ra).boff[0], (ra).boff[1], (ra).boff[2], (ra).boff[3]
#define LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

// appended prefetch workgroups of a launch whose own grid has n_main workgroups (n_main % 8 == 0 keeps the XCD residue)
static inline int pf_extra(const PfDesc* pf) { return (pf && pf->base && pf->n_consumers > 0) ? pf->n_consumers : 0; }
static inline PfDesc pf_or_none(const PfDesc* pf) { PfDesc d{}; if (pf_extra(pf)) d = *pf; return d; }


#if LA_LAB
#include "lab/lk_pf_only.inc"
#endif

int lk_pack_weight(hipStream_t st, const void* w, const void* w2, int N, int K, int il, void* out) {
    size_t total = (size_t)(il ? 2 * N : N) / 32 * (K / 16) * 64;
    k_pack_weight<<<(unsigned)((total + 255) / 256), 256, 0, st>>>((const bf16_t*)w, (const bf16_t*)w2, N, K, il, (bf16_t*)out);
    LAUNCH_CHECK(); return 0;
}
int lk_pack_x(hipStream_t st, const void* x, int K, void* out) {
    int total = LA_TB * (K / 8);
    k_pack_x<<<(total + 255) / 256, 256, 0, st>>>((const bf16_t*)x, K, (bf16_t*)out);
    LAUNCH_CHECK(); return 0;
}

// variant encoding (the `rb` argument of the public entry points): low byte = row-blocks per workgroup (1|2),
// bits 8.. = pipeline variant: 0 = default (4 waves, 8 tile-sets in flight), 1 = 8 waves x 8 (rb=1 only), 2 = 4 waves x 6
template <int RB, int EPI>
static int launch_gemm(hipStream_t st, const GemmArgs& a, int nblocks, int ksplit, int variant) {
    dim3 g(nblocks, ksplit);
    if constexpr (RB == 1) {
        if (variant == 1) { k_gemm64<RB, EPI, 8, 8><<<g, 512, 0, st>>>(G64_HEAD(a), a); LAUNCH_CHECK(); return 0; }
    }
    switch (variant) {
        case 2: k_gemm64<RB, EPI, 6, 4><<<g, 256, 0, st>>>(G64_HEAD(a), a); break;
        case 3: if constexpr (EPI == EPI_SLAB) { k_gemm64<RB, EPI, 8, 8><<<g, 512, 0, st>>>(G64_HEAD(a), a); break; }     // 8 waves x 8 tile-sets
        case 4: if constexpr (EPI == EPI_SLAB) { k_gemm64<RB, EPI, 4, 8><<<g, 512, 0, st>>>(G64_HEAD(a), a); break; }     // 8 waves x 4 tile-sets
        default: k_gemm64<RB, EPI, 8, 4><<<g, 256, 0, st>>>(G64_HEAD(a), a); break;
    }
    LAUNCH_CHECK(); return 0;
}

int lk_gemm64_slab(hipStream_t st, const void* wp, const void* xp, int N, int K, int rbv, int ksplit, float* slabs,
                   const float* route_col) {
    const int rb = rbv & 0xff, variant = rbv >> 8;
    GemmArgs a{}; a.dbg_noepi = g_la_dbg_noepi; a.kskew = g_la_kskew; a.prio_hi = g_la_prio_hi; a.dbg_times = g_la_dbg_times; a.wp = (const bf16_t*)wp; a.xp = (const bf16_t*)xp; a.K16 = K / 16; a.N = N; a.slabs = slabs;
    a.route_col = route_col; a.slab_wt = g_la_slab_wt;
    if (rb == 2 && (N % 64) == 0) return launch_gemm<2, EPI_SLAB>(st, a, N / 64, ksplit, variant);
    return launch_gemm<1, EPI_SLAB>(st, a, N / 32, ksplit, variant);
}
int lk_gemm64_swiglu(hipStream_t st, const void* wp, const void* xp, int F, int K, void* act_xp, int variant,
                     const float* route_col) {
    GemmArgs a{}; a.dbg_noepi = g_la_dbg_noepi; a.kskew = g_la_kskew; a.prio_hi = g_la_prio_hi; a.dbg_times = g_la_dbg_times; a.wp = (const bf16_t*)wp; a.xp = (const bf16_t*)xp; a.K16 = K / 16; a.N = F; a.act_xp = (bf16_t*)act_xp;
    a.route_col = route_col;
    return launch_gemm<2, EPI_SWIGLU>(st, a, F / 32, 1, variant);
}
int lk_gemm64_logits(hipStream_t st, const void* wp, const void* xp, int V, int K, int rbv, void* logits,
                     float* cv, int* ci) {
    const int rb = rbv & 0xff, variant = rbv >> 8;
    GemmArgs a{}; a.dbg_noepi = g_la_dbg_noepi; a.kskew = g_la_kskew; a.prio_hi = g_la_prio_hi; a.dbg_times = g_la_dbg_times; a.wp = (const bf16_t*)wp; a.xp = (const bf16_t*)xp; a.K16 = K / 16; a.N = V;
    a.logits = (bf16_t*)logits; a.cand_val = cv; a.cand_idx = ci;
    if (rb == 2 && (V % 64) == 0) return launch_gemm<2, EPI_LOGITS>(st, a, V / 64, 1, variant);
    return launch_gemm<1, EPI_LOGITS>(st, a, V / 32, 1, variant);
}
// QKV projection with the RoPE / fragment epilogue.  wp must be packed from the row-permuted [Wq;Wk;Wv] (lk_qkv_row_perm).
int lk_gemm64_qkv(hipStream_t st, const void* wp, const void* xp, int nh, int nkv, int K, const int* pos, const void* rcos,
                  const void* rsin, void* qf, void* kfresh, void* vfresh, int variant) {
    GemmArgs a{}; a.dbg_noepi = g_la_dbg_noepi; a.kskew = g_la_kskew; a.prio_hi = g_la_prio_hi; a.dbg_times = g_la_dbg_times; a.wp = (const bf16_t*)wp; a.xp = (const bf16_t*)xp; a.K16 = K / 16; a.N = (nh + 2 * nkv) * 128;
    a.pos = pos; a.rcos = (const bf16_t*)rcos; a.rsin = (const bf16_t*)rsin;
    a.qf = (bf16_t*)qf; a.kfresh = (bf16_t*)kfresh; a.vfresh = (bf16_t*)vfresh; a.nh = nh; a.nkv = nkv;
    return launch_gemm<2, EPI_QKV>(st, a, a.N / 64, 1, variant);
}
// packed row r of the permuted QKV matrix <- original row perm[r]: workgroup b = r/64 holds head slot b>>1, dims
// {32u + f} (row-block 0) and {64 + 32u + f} (row-block 1), u = b&1.
void lk_qkv_row_perm(int nh, int nkv, int* perm) {
    const int N = (nh + 2 * nkv) * 128;
    for (int r = 0; r < N; ++r) {
        const int nb = r >> 5, f = r & 31, b = nb >> 1, rb = nb & 1, slot = b >> 1, u = b & 1;
        perm[r] = slot * 128 + 64 * rb + 32 * u + f;
    }
}
// ---- balanced variants (one workgroup per CU): weights packed from the row plan of lk_rowplan ----
static void fill_nv(GemmRArgs& ra, int R, int blocks_per_matrix, int matrices) {
    for (int m = 0; m < matrices; ++m)
        for (int b = 0; b < blocks_per_matrix; ++b) {
            int v = R - 32 * b; if (v > 32) v = 32;
            ra.nv[m * blocks_per_matrix + b] = v;
        }
    int off = 0;
    for (int i = 0; i < matrices * blocks_per_matrix; ++i) {
        ra.nvl[i] = (ra.nv[i] + 3) & ~3;
        ra.boff[i] = off; off += ra.nvl[i] * 2 * ra.g.K16;
    }
    ra.wg_chunks = off;
}
// elements of the planned image (stored rows x K)
long lk_planned_elems(int kind, int n_rows, int K, int n_wg) {
    GemmRArgs ra{}; ra.g.K16 = K / 16;
    if (kind == 2) { const int R = (n_rows / 2) / n_wg; return (long)n_wg * 2 * ((R + 3) & ~3) * K; }
    if (kind == 1) { ra.R = n_rows / n_wg; fill_nv(ra, ra.R, 2, 2); }
    else { ra.R = n_rows / n_wg; fill_nv(ra, ra.R, 4, 1); }
    return (long)n_wg * ra.wg_chunks * 8;
}
// Prefetch descriptors (PfDesc) of the first `kib` KiB each workgroup of a GEMM launch will stream.
// planned images (k_gemm64r): kind / n_rows as lk_rowplan; 8 waves split K evenly, wave w starts at k-tile w * K16 / 8.
void lk_pf_planned(PfDesc* d, const void* wp, int kind, int n_rows, int K, int n_wg, int kib, int delay, int* sink) {
    *d = PfDesc{};
    if (!wp || kib <= 0 || n_wg <= 0) return;
    GemmRArgs ra{}; ra.g.K16 = K / 16;
    int RB;
    if (kind == 2) { const int pairs = n_rows / 2; ra.R = pairs / n_wg; RB = 2;
        ra.nvl[0] = ra.nvl[1] = (ra.R + 3) & ~3;
        ra.boff[0] = 0; ra.boff[1] = ra.nvl[0] * 2 * ra.g.K16; ra.wg_chunks = 2 * ra.boff[1]; }
    else if (kind == 1) { ra.R = n_rows / n_wg; fill_nv(ra, ra.R, 2, 2); RB = 4; }
    else { ra.R = n_rows / n_wg; fill_nv(ra, ra.R, 4, 1); RB = 4; }
    const int NW = 8, q = ra.g.K16 / NW;
    int tile_all = 0;                                   // bytes of one k-tile over all row-blocks
    for (int rb = 0; rb < RB; ++rb) tile_all += 2 * ra.nvl[rb] * 16;
    int P = (kib * 1024) / (NW * tile_all);             // k-tiles per wave
    if (P > q) P = q;
    if (P < 1) return;
    d->base = (const char*)wp; d->n_consumers = n_wg; d->nbx = n_wg; d->A = (unsigned)ra.wg_chunks * 16u; d->A2 = 0u;
    for (int rb = 0; rb < RB; ++rb) {
        const unsigned wstr = 2u * (unsigned)ra.nvl[rb] * 16u;      // bytes per k-tile of this row-block
        d->boff[rb] = (unsigned)ra.boff[rb] * 16u; d->C[rb] = (unsigned)q * wstr; d->L[rb] = (unsigned)P * wstr;
    }
    d->RB = RB; d->NW = NW; d->delay = delay; d->magic = 0x5bd1e995u; d->sink = sink;
}
// classic images (k_gemm64 over la_pack_weight, grid (N / (32 RB), ksplit)): rbv as lk_gemm64_slab
void lk_pf_classic(PfDesc* d, const void* wp, int N, int K, int rbv, int ksplit, int kib, int delay, int* sink) {
    *d = PfDesc{};
    if (!wp || kib <= 0 || ksplit <= 0) return;
    const int rb0 = rbv & 0xff, variant = rbv >> 8, K16 = K / 16;
    const int RB = (rb0 == 2 && (N % 64) == 0) ? 2 : 1;
    const int NW = (variant == 3 || variant == 4 || (RB == 1 && variant == 1)) ? 8 : 4;
    const int nbx = N / (32 * RB), per = K16 / ksplit, q = per / NW;
    if (q < 1) return;
    int P = (kib * 1024) / (NW * RB * 1024);
    if (P > q) P = q;
    if (P < 1) return;
    d->base = (const char*)wp; d->n_consumers = nbx * ksplit; d->nbx = nbx;
    d->A = (unsigned)RB * (unsigned)K16 * 1024u; d->A2 = (unsigned)per * 1024u;
    for (int rb = 0; rb < RB; ++rb) { d->boff[rb] = (unsigned)rb * (unsigned)K16 * 1024u; d->C[rb] = (unsigned)q * 1024u; d->L[rb] = (unsigned)P * 1024u; }
    d->RB = RB; d->NW = NW; d->delay = delay; d->magic = 0x5bd1e995u; d->sink = sink;
}
// planned packing (see k_pack_planned): kind as lk_rowplan
int lk_pack_planned(hipStream_t st, const void* w, const void* w2, const int* d_plan, int kind, int n_rows, int K, int n_wg, void* out) {
    GemmRArgs ra{}; ra.g.K16 = K / 16;
    PackPlanArgs pa{}; pa.n_wg = n_wg; pa.K16 = K / 16; pa.n_rows = n_rows;
    if (kind == 2) { const int pairs = n_rows / 2; ra.R = pairs / n_wg; ra.nv[0] = ra.nv[1] = ra.R; pa.RB = 2;
        ra.nvl[0] = ra.nvl[1] = (ra.R + 3) & ~3;
        ra.boff[0] = 0; ra.boff[1] = ra.nvl[0] * 2 * ra.g.K16; ra.wg_chunks = 2 * ra.boff[1]; }
    else if (kind == 1) { ra.R = n_rows / n_wg; fill_nv(ra, ra.R, 2, 2); pa.RB = 4; }
    else { ra.R = n_rows / n_wg; fill_nv(ra, ra.R, 4, 1); pa.RB = 4; }
    for (int i = 0; i < 4; ++i) { pa.nv[i] = ra.nvl[i]; pa.boff[i] = ra.boff[i]; }
    pa.wg_chunks = ra.wg_chunks;
    size_t total = (size_t)n_wg * pa.wg_chunks;
    k_pack_planned<<<(unsigned)((total + 255) / 256), 256, 0, st>>>((const bf16_t*)w, (const bf16_t*)w2, d_plan, pa, (bf16_t*)out);
    LAUNCH_CHECK(); return 0;
}
static bool set_fused_norm(GemmRArgs& ra, const FusedNorm* fn, int n_wg) {
    if (!fn || !fn->counter) return false;
    if (n_wg < LA_TB || fn->hidden > 8192 || (fn->hidden & 7)) return false;
    ra.fn_slabs = fn->slabs; ra.fn_h = (bf16_t*)fn->h; ra.fn_nw = (const bf16_t*)fn->nw; ra.fn_hidden = fn->hidden;
    ra.fn_cast = fn->cast_first; ra.fn_eps = fn->eps; ra.fn_counter = fn->counter; ra.fn_wt = fn->write_through;
    return true;
}
int lk_gemm64r_swiglu(hipStream_t st, const void* wp, const void* xp, int F, int K, int n_wg, void* act_xp,
                      const float* route_col, const FusedNorm* fn, const PfDesc* pf) {
    GemmRArgs ra{}; ra.g.dbg_noepi = g_la_dbg_noepi; ra.g.kskew = g_la_kskew; ra.g.prio_hi = g_la_prio_hi; ra.g.dbg_times = g_la_dbg_times; ra.g.wp = (const bf16_t*)wp; ra.g.xp = (const bf16_t*)xp; ra.g.K16 = K / 16; ra.g.N = F; ra.g.act_xp = (bf16_t*)act_xp;
    ra.g.route_col = route_col;
    ra.R = F / n_wg; if (F % n_wg || ra.R > 64 || ra.R <= 32) return -1;
    fill_nv(ra, ra.R, 2, 2);
    if (pf_extra(pf)) ra.pf = *pf;
    if (set_fused_norm(ra, fn, n_wg)) {
        if (fn->n_slabs != 4 || route_col) return -1;
        k_gemm64r<4, EPI_SWIGLU, 4, 8, 4><<<n_wg, 512, 8 * 4 * 4096, st>>>(G64R_HEAD(ra), ra);
    } else if (g_la_gemm_4w & 1) {
        // measurement (la_debug_set key 15 bit 0): ONE wave per SIMD, 8 tile-sets in flight per wave — no SIMD partner to lose the
        // VMEM issue arbitration to (the 8-wave form's younger waves finish their K range 10 us after the older ones)
        if (lk_gemm64r_init() != 0) return -1;
        k_gemm64r<4, EPI_SWIGLU, 8, 4><<<n_wg, 256, 4 * 4 * 4096, st>>>(G64R_HEAD(ra), ra);
    } else {
        k_gemm64r<4, EPI_SWIGLU, 4, 8><<<n_wg, 512, 8 * 4 * 4096, st>>>(G64R_HEAD(ra), ra);
    }
    LAUNCH_CHECK(); return 0;
}
int lk_gemm64r_logits(hipStream_t st, const void* wp, const void* xp, int V, int K, int n_wg, void* logits, float* cv, int* ci) {
    GemmRArgs ra{}; ra.g.dbg_noepi = g_la_dbg_noepi; ra.g.kskew = g_la_kskew; ra.g.prio_hi = g_la_prio_hi; ra.g.dbg_times = g_la_dbg_times; ra.g.wp = (const bf16_t*)wp; ra.g.xp = (const bf16_t*)xp; ra.g.K16 = K / 16; ra.g.N = V;
    ra.g.logits = (bf16_t*)logits; ra.g.cand_val = cv; ra.g.cand_idx = ci;
    ra.R = V / n_wg; if (V % n_wg || ra.R > 128 || ra.R <= 96) return -1;
    fill_nv(ra, ra.R, 4, 1);
    k_gemm64r<4, EPI_LOGITS, 4, 8><<<n_wg, 512, 8 * 4 * 4096 + LA_CAND_LDS, st>>>(G64R_HEAD(ra), ra);
    LAUNCH_CHECK(); return 0;
}
int lk_gemm64r_qkv(hipStream_t st, const void* wp, const void* xp, int nh, int nkv, int K, int n_wg, const int* pos,
                   const void* rcos, const void* rsin, void* qf, void* kfresh, void* vfresh, const FusedNorm* fn) {
    GemmRArgs ra{}; ra.g.dbg_noepi = g_la_dbg_noepi; ra.g.kskew = g_la_kskew; ra.g.prio_hi = g_la_prio_hi; ra.g.dbg_times = g_la_dbg_times; ra.g.wp = (const bf16_t*)wp; ra.g.xp = (const bf16_t*)xp; ra.g.K16 = K / 16; ra.g.N = (nh + 2 * nkv) * 128;
    ra.g.pos = pos; ra.g.rcos = (const bf16_t*)rcos; ra.g.rsin = (const bf16_t*)rsin;
    ra.g.qf = (bf16_t*)qf; ra.g.kfresh = (bf16_t*)kfresh; ra.g.vfresh = (bf16_t*)vfresh; ra.g.nh = nh; ra.g.nkv = nkv;
    const int pairs = (nh + 2 * nkv) * 64;
    ra.R = pairs / n_wg; if (pairs % n_wg || ra.R > 32) return -1;
    ra.nv[0] = ra.nv[1] = ra.R;
    ra.nvl[0] = ra.nvl[1] = (ra.R + 3) & ~3;
    ra.boff[0] = 0; ra.boff[1] = ra.nvl[0] * 2 * ra.g.K16; ra.wg_chunks = 2 * ra.boff[1];
    if (set_fused_norm(ra, fn, n_wg)) {
        if (fn->n_slabs != 4) return -1;
        k_gemm64r<2, EPI_QKV, 8, 8, 4><<<n_wg, 512, 2 * 8 * 2 * 4096, st>>>(G64R_HEAD(ra), ra);
    } else {
        k_gemm64r<2, EPI_QKV, 8, 8><<<n_wg, 512, 2 * 8 * 2 * 4096, st>>>(G64R_HEAD(ra), ra);
    }
    LAUNCH_CHECK(); return 0;
}
// Role-fused gate/up -> down_proj launch (k_gateup_down): the balanced gate/up image over n_wg workgroups + the classic
// down_proj image as N/64 row-groups x ksplit.  counter: a zeroed device int per launch and step.
int lk_gateup_down(hipStream_t st, const void* wgu, const void* xp, int F, int K, int n_wg, void* act_xp, const void* wdown, int N,
                   int ksplit, float* slabs, int* counter, int dd, const FusedNorm* fn) {
    GemmRArgs ra{}; ra.g.kskew = 0; ra.g.wp = (const bf16_t*)wgu; ra.g.xp = (const bf16_t*)xp; ra.g.K16 = K / 16; ra.g.N = F; ra.g.act_xp = (bf16_t*)act_xp;
    ra.R = F / n_wg; if (F % n_wg || ra.R > 64 || ra.R <= 32 || (N % 64) || (F % 16) || !counter) return -1;
    fill_nv(ra, ra.R, 2, 2);
    GemmArgs d{}; d.wp = (const bf16_t*)wdown; d.xp = (const bf16_t*)act_xp; d.K16 = F / 16; d.N = N; d.slabs = slabs;
    const int gx = N / 64, n_dn = gx * ksplit;
    if (lk_gemm64r_init() != 0) return -1;
#define GD(DD) k_gateup_down<2, DD><<<n_wg + n_dn, 512, 8 * 4 * 4096, st>>>(ra.g.wp, ra.g.xp, ra.g.K16, 0, ra.wg_chunks, \
        ((unsigned)ra.nvl[0] | ((unsigned)ra.nvl[1] << 8) | ((unsigned)ra.nvl[2] << 16) | ((unsigned)ra.nvl[3] << 24)), \
        ra.boff[0], ra.boff[1], ra.boff[2], ra.boff[3], n_wg, ra, d, gx, ksplit, counter)
#define GDN(DD) k_gateup_down<2, DD, 4><<<n_wg + n_dn, 512, 8 * 4 * 4096, st>>>(ra.g.wp, ra.g.xp, ra.g.K16, 0, ra.wg_chunks, \
        ((unsigned)ra.nvl[0] | ((unsigned)ra.nvl[1] << 8) | ((unsigned)ra.nvl[2] << 16) | ((unsigned)ra.nvl[3] << 24)), \
        ra.boff[0], ra.boff[1], ra.boff[2], ra.boff[3], n_wg, ra, d, gx, ksplit, counter)
    if (set_fused_norm(ra, fn, n_wg)) {
        if (fn->n_slabs != 4) return -1;
        if (dd == 8) GDN(8); else GDN(4);
    } else if (dd == 8) GD(8); else GD(4);
#undef GDN
#undef GD
    LAUNCH_CHECK(); return 0;
}
static bool g_attr_done = false;
int lk_gemm64r_init() {
    if (g_attr_done) return 0;
    if (hipFuncSetAttribute((const void*)k_tree_attn<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * LA_ATT_PAR * 66 * 64 * sizeof(float)) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void*)k_tree_attn<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * LA_ATT_PAR * 66 * 64 * sizeof(float)) != hipSuccess) return -1;
    hipError_t e = hipFuncSetAttribute((const void*)k_gemm64r<4, EPI_SWIGLU, 4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 4 * 4096);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm64r<4, EPI_LOGITS, 4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 4 * 4096 + LA_CAND_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm64r<2, EPI_QKV, 8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 2 * 4096);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm64r<4, EPI_SWIGLU, 4, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 4 * 4096);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm64r<2, EPI_QKV, 8, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 2 * 4096);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm64r<4, EPI_SWIGLU, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 4 * 4096);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gateup_down<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 4 * 4096);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gateup_down<2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 4 * 4096);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gateup_down<2, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 4 * 4096);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gateup_down<2, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 4 * 4096);
    if (e != hipSuccess) return (int)e;
    if (lk_attn1_init() != 0) return -1;
    g_attr_done = true;
    return 0;
}
// Row plan of the balanced packing: out[i] = source row of packed row i (for kind 1 rows >= n_rows address the second
// matrix: row - n_rows of `up`), or -1 for a zero pad row.  Returns the number of packed rows (n_wg * RB * 32).
//   kind 0: one matrix [n_rows][K] (lm_head), R = n_rows/n_wg, blocks of 32 per workgroup
//   kind 1: gate/up pair, R = n_rows/n_wg: blocks {G0, G1, U0, U1}
//   kind 2: qkv, n_rows = (nh+2nkv)*128: R = pairs/n_wg RoPE pairs per workgroup: blocks {lo, hi}
int lk_rowplan(int kind, int n_rows, int n_wg, int* out) {
    if (kind == 2) {
        const int pairs = n_rows / 2, R = pairs / n_wg;
        if (pairs % n_wg || R > 32) return -1;
        if (out)
            for (int w = 0; w < n_wg; ++w)
                for (int rb = 0; rb < 2; ++rb)
                    for (int f = 0; f < 32; ++f) {
                        int v = -1;
                        if (f < R) { const int pr = R * w + f, slot = pr >> 6, dlo = pr & 63; v = slot * 128 + dlo + 64 * rb; }
                        out[(w * 2 + rb) * 32 + f] = v;
                    }
        return n_wg * 64;
    }
    const int R = n_rows / n_wg;
    if (n_rows % n_wg) return -1;
    const int bpm = kind == 1 ? 2 : 4, mats = kind == 1 ? 2 : 1;
    if (R > 32 * bpm || R <= 32 * (bpm - 1)) return -1;
    if (out)
        for (int w = 0; w < n_wg; ++w)
            for (int m = 0; m < mats; ++m)
                for (int b = 0; b < bpm; ++b)
                    for (int f = 0; f < 32; ++f) {
                        const int rr = 32 * b + f;
                        out[((w * mats + m) * bpm + b) * 32 + f] = rr < R ? m * n_rows + R * w + rr : -1;
                    }
    return n_wg * mats * bpm * 32;
}
// number of [64]-token candidate slots la_gemm64_logits writes (input of lk_argmax_finalize)
int lk_logits_cand_slots(int V, int rbv) {
    const int rb = ((rbv & 0xff) == 2 && (V % 64) == 0) ? 2 : 1, variant = rbv >> 8;
    const int nw = (rb == 1 && variant == 1) ? 8 : 4;
    return V / (32 * rb) * (nw / 2);
}
int lk_argmax_finalize(hipStream_t st, const float* cv, const int* ci, int n_tiles, int* out_rows) {
    k_argmax_finalize<<<LA_TB, 256, 0, st>>>(cv, ci, n_tiles, out_rows);
    LAUNCH_CHECK(); return 0;
}
int lk_embed_norm(hipStream_t st, const void* embed, const int* ids, const void* nw, int hidden, float eps, void* h, void* xp,
                  int cast_first, const PfDesc* pf) {
    if (hidden > 8192 || (hidden & 7)) return -1;
    k_row_norm<0, false><<<LA_TB + pf_extra(pf), 512, 0, st>>>((const bf16_t*)embed, ids, (bf16_t*)h, nullptr, (const bf16_t*)nw, hidden, eps,
                                                (bf16_t*)xp, nullptr, nullptr, 0, 0, nullptr, nullptr, cast_first, pf_or_none(pf));
    LAUNCH_CHECK(); return 0;
}
int lk_resid_norm(hipStream_t st, void* h, const float* slabs, int n_slabs, const void* nw, int hidden, float eps, void* xp,
                  int cast_first, const PfDesc* pf) {
    if (hidden > 8192 || (hidden & 7)) return -1;
    const PfDesc pfd = pf_or_none(pf);
#define RN(NS) k_row_norm<NS, false><<<LA_TB + pf_extra(pf), 512, 0, st>>>(nullptr, nullptr, (bf16_t*)h, slabs, (const bf16_t*)nw, hidden, eps, \
                                                            (bf16_t*)xp, nullptr, nullptr, 0, 0, nullptr, nullptr, cast_first, pfd)
    switch (n_slabs) {
        case 0: RN(0); break; case 1: RN(1); break; case 2: RN(2); break; case 3: RN(3); break;
        case 4: RN(4); break; case 6: RN(6); break; case 8: RN(8); break;
        default: return -1;
    }
#undef RN
    LAUNCH_CHECK(); return 0;
}
// residual add + norm + fused MoE router (post-attention norm of a Mixtral layer)
int lk_resid_norm_router(hipStream_t st, void* h, const float* slabs, int n_slabs, const void* nw, int hidden, float eps,
                         void* xp, const void* wrouter, int n_experts, int top_k, float* route_w, const int* n_rows,
                         int cast_first) {
    if (hidden > 8192 || (hidden & 7) || n_experts < 1 || n_experts > LA_MOE_MAX_E || top_k < 1 || top_k > n_experts) return -1;
#define RN(NS) k_row_norm<NS, true><<<LA_TB, 512, 0, st>>>(nullptr, nullptr, (bf16_t*)h, slabs, (const bf16_t*)nw, hidden, eps, \
                                                           (bf16_t*)xp, nullptr, (const bf16_t*)wrouter, n_experts, top_k, route_w, n_rows, cast_first, PfDesc{})
    switch (n_slabs) {
        case 1: RN(1); break; case 2: RN(2); break; case 3: RN(3); break; case 4: RN(4); break;
        case 6: RN(6); break; case 8: RN(8); break;
        default: return -1;
    }
#undef RN
    LAUNCH_CHECK(); return 0;
}
// residual + accumulated expert outputs (bf16) + next norm
int lk_resid_norm_addend(hipStream_t st, void* h, const void* addend, const void* nw, int hidden, float eps, void* xp,
                         int cast_first) {
    if (hidden > 8192 || (hidden & 7) || !addend) return -1;
    k_row_norm<0, false><<<LA_TB, 512, 0, st>>>(nullptr, nullptr, (bf16_t*)h, nullptr, (const bf16_t*)nw, hidden, eps,
                                                (bf16_t*)xp, (const bf16_t*)addend, nullptr, 0, 0, nullptr, nullptr, cast_first, PfDesc{});
    LAUNCH_CHECK(); return 0;
}
int lk_moe_accum(hipStream_t st, const float* slabs, int n_slabs, const float* route_col, int hidden, void* acc, int first) {
    if (hidden & 7) return -1;
#define MA(NS) do { if (first) k_moe_accum<NS, true><<<LA_TB, 256, 0, st>>>(slabs, route_col, hidden, (bf16_t*)acc); \
                    else k_moe_accum<NS, false><<<LA_TB, 256, 0, st>>>(slabs, route_col, hidden, (bf16_t*)acc); } while (0)
    switch (n_slabs) {
        case 1: MA(1); break; case 2: MA(2); break; case 3: MA(3); break; case 4: MA(4); break;
        case 6: MA(6); break; case 8: MA(8); break;
        default: return -1;
    }
#undef MA
    LAUNCH_CHECK(); return 0;
}
int lk_build_tree_inputs(hipStream_t st, const int* in, int* state, int* pos, uint64_t* rowmask, int* ids) {
    k_build_tree_inputs<<<1, 64, 0, st>>>(in, state, pos, (unsigned long long*)rowmask, ids);
    LAUNCH_CHECK(); return 0;
}
int lk_qkv_post(hipStream_t st, const float* slabs, int n_slabs, int nh, int nkv, const int* pos, const void* rcos,
                const void* rsin, void* qf, void* kfresh, void* vfresh) {
#define QP(NS) k_qkv_post<NS><<<nh + 2 * nkv, 256, 0, st>>>(slabs, nh, nkv, pos, (const bf16_t*)rcos, (const bf16_t*)rsin, \
                                                          (bf16_t*)qf, (bf16_t*)kfresh, (bf16_t*)vfresh)
    switch (n_slabs) {
        case 1: QP(1); break; case 2: QP(2); break; case 3: QP(3); break; case 4: QP(4); break; case 8: QP(8); break;
        default: return -1;
    }
#undef QP
    LAUNCH_CHECK(); return 0;
}
static int tree_attn_launch(hipStream_t st, AttnArgs a, int n_slots, void* attn_xp, const PfDesc* pf);

int lk_qk_scale_check(int head_dim) {
    if (head_dim < 2 || head_dim > 128 || (head_dim & 1)) return -1;
    if (LA_DTYPE == 1) return 0;                               // the fp16 build divides
    auto rne = [](float f) -> uint32_t {                       // fp32 -> bf16 bits, round-to-nearest-even (finite inputs)
        uint32_t u; memcpy(&u, &f, 4);
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    };
    const float mul = (float)(1.0 / sqrt((double)head_dim)), div = (float)sqrt((double)head_dim);
    for (uint32_t b = 0; b < 65536u; ++b) {
        if (((b >> 7) & 0xffu) == 0xffu) continue;             // inf / nan
        const uint32_t u = b << 16;
        float x; memcpy(&x, &u, 4);
        if (rne(x * mul) != rne(x / div)) return 1;
    }
    return 0;
}

int lk_tree_attn_b(hipStream_t st, const void* qf, const void* kmain, const void* vmain, const void* kfresh,
                   const void* vfresh, const uint64_t* rowmask, const int* bstate, int nh, int nkv, int slot_keys,
                   int n_slots, int nsplit, float* opart, float* mpart, float* lpart, void* attn_xp, int window, int ring_keys,
                   const PfDesc* pf, int head_dim) {
    if (n_slots < 1 || n_slots > LA_MAX_SEQ || (slot_keys & 31)) return -1;
    AttnArgs a{};
    a.qk = la_qk_scale(head_dim);
    a.dbg_times = g_la_dbg_times;
    a.window = window; a.ring_tiles = ring_keys >> 5;
    a.qf = (const bf16_t*)qf; a.kmain = (const bf16_t*)kmain; a.vmain = (const bf16_t*)vmain;
    a.kfresh = (const bf16_t*)kfresh; a.vfresh = (const bf16_t*)vfresh;
    a.rowmask = (const unsigned long long*)rowmask; a.state = nullptr;
    a.nh = nh; a.nkv = nkv; a.max_keys = slot_keys * n_slots; a.nsplit = nsplit;
    a.opart = opart; a.mpart = mpart; a.lpart = lpart;
    a.seq = bstate + LA_BST_SEQ; a.nkeys_b = bstate + LA_BST_NKEYS; a.slot_tiles = slot_keys >> 5;
    return tree_attn_launch(st, a, n_slots, attn_xp, pf);
}

int lk_tree_attn(hipStream_t st, const void* qf, const void* kmain, const void* vmain, const void* kfresh,
                 const void* vfresh, const uint64_t* rowmask, const int* state, int nh, int nkv, int max_keys,
                 int nsplit, float* opart, float* mpart, float* lpart, void* attn_xp, int window, int ring_keys,
                 const PfDesc* pf, int form, const PfDesc* ride, int head_dim) {
    AttnArgs a{};
    a.qk = la_qk_scale(head_dim);
    a.dbg_times = g_la_dbg_times;
    a.window = window; a.ring_tiles = ring_keys >> 5;
    a.qf = (const bf16_t*)qf; a.kmain = (const bf16_t*)kmain; a.vmain = (const bf16_t*)vmain;
    a.kfresh = (const bf16_t*)kfresh; a.vfresh = (const bf16_t*)vfresh;
    a.rowmask = (const unsigned long long*)rowmask; a.state = state;
    a.nh = nh; a.nkv = nkv; a.max_keys = max_keys; a.nsplit = nsplit;
    a.opart = opart; a.mpart = mpart; a.lpart = lpart;
    a.seq = nullptr; a.nkeys_b = nullptr; a.slot_tiles = 0;
    // default: the single-launch form (la_attn1.hip); la_debug_set(17, 0) = key splits + combine (the A/B switch, and the carrier of
    // the idle-window prefetch workgroups)
    if (g_la_attn_one && form != 0 && !pf_extra(pf) && !g_la_attn_staged)
        return lk_tree_attn1(st, qf, kmain, vmain, kfresh, vfresh, rowmask, state, nh, nkv, max_keys, attn_xp, window, ring_keys, ride, head_dim);
    return tree_attn_launch(st, a, 1, attn_xp, pf);
}

static int tree_attn_launch(hipStream_t st, AttnArgs a, int n_slots, void* attn_xp, const PfDesc* pf) {
    const int nh = a.nh, nsplit = a.nsplit;
    float *opart = a.opart, *mpart = a.mpart, *lpart = a.lpart;
    if (lk_gemm64r_init() != 0) return -1;
    static_assert(2 * LA_ATT_PAR * 66 * 64 * sizeof(float) >= 2 * LA_ATT_PAR * 16384, "the K/V ring of the staged form aliases the merge buffer");
#define ATT_HEAD(a) (a).qf, (a).rowmask, (a).state, (a).seq, (a).kmain, (a).vmain, (a).max_keys, (a).nsplit, (((a).nh << 16) | (a).nkv), (a).window
    if (g_la_attn_staged) k_tree_attn<true><<<dim3(nh, nsplit, n_slots), 2 * LA_ATT_PAR * 64, 2 * LA_ATT_PAR * 66 * 64 * sizeof(float), st>>>(ATT_HEAD(a), a);
    else k_tree_attn<false><<<dim3(nh, nsplit, n_slots), 2 * LA_ATT_PAR * 64, 2 * LA_ATT_PAR * 66 * 64 * sizeof(float), st>>>(ATT_HEAD(a), a);
#undef ATT_HEAD
    LAUNCH_CHECK();
    if (!attn_xp) return 0;                          // lab knob 33: the consumer (k_oproj_merge) reads the partials itself
    int total = nh * LA_TB * 16;
    const int n_main = (total + 255) / 256;
    if (n_main % 8) pf = nullptr;                    // appended ids would land on other XCDs than their consumers
    const PfDesc pfd = pf_or_none(pf);
#define AC(NS) k_attn_combine<NS><<<n_main + pf_extra(pf), 256, 0, st>>>(opart, mpart, lpart, nh, a.seq, (bf16_t*)attn_xp, n_main, pfd)
    switch (nsplit) {
        case 1: AC(1); break; case 2: AC(2); break; case 3: AC(3); break; case 4: AC(4); break; case 6: AC(6); break;
        case 8: AC(8); break; case 12: AC(12); break; case 16: AC(16); break;
        default: return -1;
    }
#undef AC
    LAUNCH_CHECK(); return 0;
}
// ---- merged MoE launches: E experts in one grid ----
int lk_gemm64r_swiglu_ex(hipStream_t st, const void* wp0, long w_stride, const void* xp, int F, int K, int n_wg, void* act0,
                         long act_stride, const float* route_w, int E) {
    GemmRArgs ra{}; ra.g.dbg_noepi = g_la_dbg_noepi; ra.g.kskew = g_la_kskew; ra.g.prio_hi = g_la_prio_hi; ra.g.dbg_times = g_la_dbg_times; ra.g.wp = (const bf16_t*)wp0; ra.g.xp = (const bf16_t*)xp; ra.g.K16 = K / 16; ra.g.N = F; ra.g.act_xp = (bf16_t*)act0;
    ra.g.route_col = route_w; ra.g.ex_on = 1; ra.g.ex_w = w_stride; ra.g.ex_x = 0; ra.g.ex_act = act_stride;
    ra.R = F / n_wg; if (F % n_wg || ra.R > 64 || ra.R <= 32 || E < 1 || E > LA_MOE_MAX_E) return -1;
    fill_nv(ra, ra.R, 2, 2);
    k_gemm64r<4, EPI_SWIGLU, 4, 8><<<dim3(n_wg, E), 512, 8 * 4 * 4096, st>>>(G64R_HEAD(ra), ra);
    LAUNCH_CHECK(); return 0;
}
int lk_gemm64_swiglu_ex(hipStream_t st, const void* wp0, long w_stride, const void* xp, int F, int K, void* act0, long act_stride,
                        const float* route_w, int E) {
    GemmArgs a{}; a.dbg_noepi = g_la_dbg_noepi; a.kskew = g_la_kskew; a.prio_hi = g_la_prio_hi; a.dbg_times = g_la_dbg_times; a.wp = (const bf16_t*)wp0; a.xp = (const bf16_t*)xp; a.K16 = K / 16; a.N = F; a.act_xp = (bf16_t*)act0;
    a.route_col = route_w; a.ex_on = 1; a.ex_w = w_stride; a.ex_act = act_stride;
    if (E < 1 || E > LA_MOE_MAX_E) return -1;
    k_gemm64<2, EPI_SWIGLU, 8, 4><<<dim3(F / 32, 1, E), 256, 0, st>>>(G64_HEAD(a), a);
    LAUNCH_CHECK(); return 0;
}
int lk_gemm64_slab_ex(hipStream_t st, const void* wp0, long w_stride, const void* xp0, long x_stride, int N, int K, int rbv, int ksplit,
                      float* slabs0, long slab_stride, const float* route_w, int E) {
    const int rb = rbv & 0xff;
    GemmArgs a{}; a.dbg_noepi = g_la_dbg_noepi; a.kskew = g_la_kskew; a.prio_hi = g_la_prio_hi; a.dbg_times = g_la_dbg_times; a.wp = (const bf16_t*)wp0; a.xp = (const bf16_t*)xp0; a.K16 = K / 16; a.N = N; a.slabs = slabs0;
    a.route_col = route_w; a.ex_on = 1; a.ex_w = w_stride; a.ex_x = x_stride; a.ex_slab = slab_stride;
    if (E < 1 || E > LA_MOE_MAX_E) return -1;
    if (rb == 2 && (N % 64) == 0) k_gemm64<2, EPI_SLAB, 8, 4><<<dim3(N / 64, ksplit, E), 256, 0, st>>>(G64_HEAD(a), a);
    else k_gemm64<1, EPI_SLAB, 8, 4><<<dim3(N / 32, ksplit, E), 256, 0, st>>>(G64_HEAD(a), a);
    LAUNCH_CHECK(); return 0;
}
int lk_moe_accum_all(hipStream_t st, const float* slabs0, long slab_stride, int n_slabs, const float* route_w, int E, int hidden, void* acc) {
    if (hidden & 7) return -1;
#define MA(NS) k_moe_accum_all<NS><<<LA_TB, 256, 0, st>>>(slabs0, slab_stride, route_w, E, hidden, (bf16_t*)acc)
    switch (n_slabs) {
        case 1: MA(1); break; case 2: MA(2); break; case 3: MA(3); break; case 4: MA(4); break;
        case 6: MA(6); break; case 8: MA(8); break;
        default: return -1;
    }
#undef MA
    LAUNCH_CHECK(); return 0;
}
int lk_step_head(hipStream_t st, const int* in, int* state, int* pos, uint64_t* rowmask, int* ids, const void* embed, const void* nw,
                 int hidden, float eps, void* h, void* xp, int cast_first, const PfDesc* pf, uint64_t* gran, int n_gran) {
    if (hidden > 8192 || (hidden & 7)) return -1;
    k_step_head<<<LA_TB + pf_extra(pf), 512, 0, st>>>(in, state, pos, (unsigned long long*)rowmask, ids, (const bf16_t*)embed, (bf16_t*)h,
                                                      (const bf16_t*)nw, hidden, eps, (bf16_t*)xp, cast_first, pf_or_none(pf),
                                                      (unsigned long long*)gran, gran ? n_gran : 0);
    LAUNCH_CHECK(); return 0;
}
#if LA_LAB
#include "lab/lk_resid_norm4.inc"
#endif
// cv / ci: [n_tiles][64] candidates (one per lm_head workgroup and token); host_out: pinned result block or null
int lk_step_tail(hipStream_t st, const float* cv, const int* ci, int n_tiles, const int* ids, const uint64_t* rowmask, int* state,
                 int* host_out) {
    k_step_tail<<<1, 1024, 0, st>>>(cv, ci, n_tiles, ids, (const unsigned long long*)rowmask, state, host_out);
    LAUNCH_CHECK(); return 0;
}
int lk_publish(hipStream_t st, int* state, int* host_out) {
    k_publish<<<1, 128, 0, st>>>(state, host_out);
    LAUNCH_CHECK(); return 0;
}
int lk_accept_scan(hipStream_t st, const int* ids, const uint64_t* rowmask, int* state) {
    k_accept_scan<<<1, 64, 0, st>>>(ids, (const unsigned long long*)rowmask, state);
    LAUNCH_CHECK(); return 0;
}
int lk_build_tree_inputs_b(hipStream_t st, const int* in, int* bstate, int* pos, uint64_t* rowmask, int* ids) {
    k_build_tree_inputs_b<<<1, 64, 0, st>>>(in, bstate, pos, (unsigned long long*)rowmask, ids);
    LAUNCH_CHECK(); return 0;
}
int lk_accept_scan_b(hipStream_t st, const int* in, const int* ids, const uint64_t* rowmask, int* bstate, int n_slots,
                     int slot_keys, int ring) {
    if (n_slots < 1 || n_slots > LA_MAX_SEQ) return -1;
    k_accept_scan_b<<<n_slots, 64, 0, st>>>(in, ids, (const unsigned long long*)rowmask, bstate, slot_keys, ring);
    LAUNCH_CHECK(); return 0;
}
int lk_kv_commit_b(hipStream_t st, const void* kfresh, const void* vfresh, void* kmain, void* vmain, const int* bstate,
                   int n_layers, int nkv, int total_keys) {
    k_kv_commit_b<<<n_layers * nkv, 256, 0, st>>>((const bf16_t*)kfresh, (const bf16_t*)vfresh, (bf16_t*)kmain,
                                                  (bf16_t*)vmain, bstate, total_keys);
    LAUNCH_CHECK(); return 0;
}
int lk_kv_commit(hipStream_t st, const void* kfresh, const void* vfresh, void* kmain, void* vmain, const int* state,
                 int n_layers, int nkv, int max_keys, int ring) {
    k_kv_commit<<<n_layers * nkv, 256, 0, st>>>((const bf16_t*)kfresh, (const bf16_t*)vfresh, (bf16_t*)kmain,
                                                (bf16_t*)vmain, state, nkv, max_keys, ring);
    LAUNCH_CHECK(); return 0;
}
