// la_attn1.hip — tree attention of the single-sequence verify step as ONE launch (round 4).
//
// Reference: LlamaAttention.forward (lookahead/lookahead/models/llama/modeling_llama.py:270-296) under the rank-4 tree mask of
// lookahead_prepare_inputs_for_generation (common/pretrained_model.py:725-734); GQA: models/mistral/modeling_mistral.py:236-318.
//
// What it replaces: k_tree_attn (heads x 8 key splits, each writing fp32 partials: 10 MB per layer) + k_attn_combine (a second,
// dependent launch that reads them back).  The key split existed to spread the K/V stream over all CUs; its price was 2x the
// algorithmic traffic and a launch whose only work is a merge.
//
// Here a workgroup owns (head, token block of 32 rows, token slice of 32 / SL rows) and sees ALL keys of its head:
//   * its 8 waves are the 8 key-tile parities (tile = 32 keys); each wave runs the online softmax over its tiles, the waves meet
//     ONCE in LDS, and the workgroup normalises and stores its slice of the o_proj operand (bf16, packed XP layout) itself —
//     no partials in HBM, no second launch, nothing another workgroup has to wait for;
//   * the nh/nkv * 2 * SL workgroups that read the same K/V (one kv head) sit on one XCD (block id mod 8): the head's K/V is
//     fetched from HBM once per XCD (concurrent misses on a line merge in L2) and re-read from that XCD's L2, where a CU reads ~6x
//     faster than from HBM (per-CU rate = bytes in flight / latency).  A performance assumption only: a tile nobody has fetched
//     yet is simply a miss.  (Round 4 rotated the sharers' start offsets — worth 0.01 ms per step; round 5 trades that for the
//     first K tile requested before the cursor arrives, see `spec`.)
//   * MFMA tiles are 32 tokens wide whatever the slice, so the 32 / SL-row slice costs SL x redundant matrix work — free here
//     (the launch is bound by memory latency; the matrix pipe is idle), and it buys SL x more workgroups in flight.
// Same arithmetic per (row, key) as k_tree_attn (bf16(QK^T) * 1/sqrt(d) -> bf16, fp32 softmax, bf16 P, fp32 PV accumulation,
// bf16 output); the summation order over keys differs (tile order per wave, one merge level instead of two).
#include <type_traits>
#include "la_common.h"
#include "la_kernels.h"
#include "la_knobs.h"

#define LA_NEG (-1.0e30f)

struct Attn1Args {
    const bf16_t* kfresh;
    const bf16_t* vfresh;
    bf16_t* attn_xp;
    long long* dbg_times;     // measurement aid: [workgroup][wave][8] wall-clock stamps, null in production
    float qk;                 // la_qk_scale(head_dim) (attn_scale, la_common.h)
    int n_main;               // workgroups of the attention proper; block ids past it are weight-prefetch riders (round 6, see pf)
    PfDesc pf;                // riders: the first KiB every o_proj workgroup will stream, pulled into its XCD's L2 by the CUs this launch leaves idle
};

__device__ __forceinline__ void attn1_rider(const PfDesc& p, int b);

// Leading scalars (kernel-argument preload, build.sh): everything between dispatch and the first K-tile request.
// RIDE = the instantiation whose grid carries weight-prefetch rider workgroups behind the attention's own (lab knob 31).
template <bool RIDE>
__global__ __launch_bounds__(512) void k_tree_attn1(const bf16_t* __restrict__ qf, const unsigned long long* __restrict__ rowmask,
                                                     const int* __restrict__ state, const bf16_t* __restrict__ kmain,
                                                     const bf16_t* __restrict__ vmain, int max_keys, int nh_nkv, int window,
                                                     int sl_ring, Attn1Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds1[];      // [Q fragments: 8 KiB][merge buffer]
    if constexpr (RIDE) { if ((int)blockIdx.x >= a.n_main) { attn1_rider(a.pf, (int)blockIdx.x - a.n_main); return; } }
    const int nh = nh_nkv >> 16, nkv = nh_nkv & 0xffff, G = nh / nkv;
    const int SL = sl_ring & 63, ring_tiles = sl_ring >> 8;
    const int W = 32 / SL;                                            // token rows this workgroup stores
    const int NS = G * 2 * SL;                                        // workgroups that read the same kv head
    const int b = blockIdx.x;
    int hk, r;
    if ((nkv & 7) == 0) { const int x = b & 7, q = b >> 3; hk = x * (nkv >> 3) + q / NS; r = q % NS; }     // sharers on one XCD
    else { hk = b / NS; r = b % NS; }
    const int tb = (r / SL) & 1, sl = r % SL;
    const int h = hk * G + r / (2 * SL);
    const int lane = threadIdx.x & 63;
    const int par = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hh = lane >> 5;
    long long* const stamp = a.dbg_times ? a.dbg_times + ((size_t)b * 8 + par) * 8 : nullptr;
    if (stamp && lane == 0) stamp[0] = wall_clock64();

    const int KB = max_keys >> 5;
    // First K tile BEFORE the cursor is known (round 5).  The tile list hangs on `nkeys`, a scalar load of a line another XCD wrote:
    // ~1.8 us behind dispatch (stamps, profiles/r04_attention_one_launch.txt).  Wave `par` starts at tile `par`, and with a linear,
    // unwindowed cache that tile's address depends on kernel arguments only (preloaded scalars), so its 8 fragments are requested
    // at once and KEPT — unlike round 4's speculative touch, which dropped its data and queued the real request behind it.  The
    // tile is committed (valid) whenever par < NP, i.e. from 225 + 1 committed keys on for every wave; below that, or with a
    // window / ring, the wave simply requests its real first tile once nkeys has arrived (the early data is never used then; the
    // addresses read are inside the layer's cache: KB >= 8).  Price: the start rotation of the sharers (worth 0.01 ms per step).
    // (bit 7 of sl_ring = la_lab_set key 18 bit 0: the early request off — the A/B switch; results are bit-identical either way)
    const bool spec = window <= 0 && ring_tiles == 0 && KB >= 8 && (sl_ring & 128) == 0;
    bf16x8 kA[8], kB[8], vE[8];
    if (spec) {
        const bf16x8* kt = (const bf16x8*)(kmain + ((size_t)hk * KB + par) * 4096);
#pragma unroll
        for (int s = 0; s < 8; ++s) kA[s] = kt[s * 64 + lane];
        // ... and its V tile behind it: with K early, the first tile's V (requested inside tile(), i.e. after the cursor) became the
        // load the first PV MFMA waits for (GPU call 1 of round 5: the K-only form gained 0.4 of the 1.8 us per layer)
        const bf16x8* vt0 = (const bf16x8*)(vmain + ((size_t)hk * KB + par) * 4096);
#pragma unroll
        for (int s = 0; s < 8; ++s) vE[s] = vt0[s * 64 + lane];
    }
    // Q fragments of (h, tb): wave `par` brings fragment `par` (1 KiB); all waves read the 8 fragments back per tile
    bf16x8* const qs = (bf16x8*)lds1;
    const bf16x8 qmine = *((const bf16x8*)(qf + ((size_t)(h * 2 + tb) * 8 + par) * 512) + lane);
    const unsigned long long rm = rowmask[tb * 32 + (lane & 31)];
    const int nkeys = state[LA_ST_NKEYS];
    // sliding window (HF Mistral mask rule: visible iff pos_row - j <= window): tiles wholly below the root's horizon are skipped
    const int NPall = (nkeys + 31) >> 5;
    const int ts = (window > 0 && nkeys - window > 0) ? ((nkeys - window) >> 5) : 0;
    const int NP = NPall - ts, NT = NP + 2;                           // committed tiles + the 2 fresh tiles of the tree
    const int key_lo = (window > 0) ? nkeys + __popcll(rm) - 1 - window : 0;
    // this wave's tiles: par, par + 8, ... in list order (every sharer of the head starts at the head of its list: see `spec` above)
    const int cnt = NT > par ? (NT - par + 7) >> 3 : 0;
    int idx = 0;

    auto mtile = [&](int it) -> size_t { return (size_t)(ring_tiles > 0 ? (ts + it) % ring_tiles : ts + it); };
    auto kptr = [&](int it) -> const bf16x8* {
        return it >= NP ? (const bf16x8*)(a.kfresh + ((size_t)hk * 2 + (it - NP)) * 4096)
                        : (const bf16x8*)(kmain + ((size_t)hk * KB + mtile(it)) * 4096);
    };
    auto vptr = [&](int it) -> const bf16x8* {
        return it >= NP ? (const bf16x8*)(a.vfresh + ((size_t)hk * 2 + (it - NP)) * 4096)
                        : (const bf16x8*)(vmain + ((size_t)hk * KB + mtile(it)) * 4096);
    };

    f32x16 o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    float m = LA_NEG, l = 0.f;

    // one key tile: S^T = K.Q^T (a lane owns one token column), mask, online softmax, O^T += V^T.P^T; V is requested before the
    // QK^T MFMAs and consumed after the softmax, the NEXT tile's K fragments are requested by the caller first
    auto tile = [&](int it, const bf16x8 (&kf)[8], auto v_early) {
        const bool fresh = it >= NP;
        const int kb = fresh ? it - NP : it;
        bf16x8 vf[8];
        if constexpr (decltype(v_early)::value) {       // the wave's FIRST tile: its V is in flight already (vE)
#pragma unroll
            for (int s = 0; s < 8; ++s) vf[s] = vE[s];
        } else {
            const bf16x8* vt = vptr(it);
#pragma unroll
            for (int s = 0; s < 8; ++s) vf[s] = vt[s * 64 + lane];
        }
        f32x16 sc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) sc = LA_MFMA(kf[s], qs[s * 64 + lane], sc, 0, 0, 0);
        float mx = LA_NEG;
        // committed tile every row sees whole (no window, all 32 keys below nkeys): no mask arithmetic (wave-uniform)
        const bool whole = !fresh && window <= 0 && (ts + kb) * 32 + 31 < nkeys;
        if (whole) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                // attn_weights = bf16(QK^T) / sqrt(head_dim) -> bf16 (modeling_llama.py:270); bf16(x / sqrt(128)) == bf16(x * fp32(1 / sqrt(128)))
                // for every finite bf16 x (tests/test_oracle_llama.py::test_attention_scale_as_multiply_is_exact)
                const float v = attn_scale(sc[i], a.qk);
                sc[i] = v;
                mx = fmaxf(mx, v);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kk = (i & 3) + 8 * (i >> 2) + 4 * hh;
                float v = attn_scale(sc[i], a.qk);
                const int kidx = (ts + kb) * 32 + kk;                  // committed keys: absolute index = position
                const bool ok = fresh ? ((rm >> (kb * 32 + kk)) & 1ull) != 0ull : (kidx < nkeys && kidx >= key_lo);
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
                const float p = (sc[i] > -1.0e29f) ? __expf(sc[i] - mn) : 0.f;
                ps += p;
                pf[i >> 3][i & 7] = (short)f2bf(p);
            }
        }
        ps += __shfl_xor(ps, 32, 64);
        l = l * alpha + ps;
        m = mn;
        if (stamp && lane == 0 && stamp[5] == 0) stamp[5] = wall_clock64();      // first tile: scores done, V not yet awaited
        if (__ballot(alpha != 1.0f) != 0ull) {          // a running maximum moved for some row: rescale (x * 1.0f is exact, so skipping is too)
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            o[db] = LA_MFMA(vf[db * 2 + 0], pf[0], o[db], 0, 0, 0);
            o[db] = LA_MFMA(vf[db * 2 + 1], pf[1], o[db], 0, 0, 0);
        }
    };

    int it = par;
    const bool early_ok = spec && par < NP;
    if (cnt > 0 && !early_ok) {                     // wave-uniform: the early request did not fetch this wave's first tile (K and V)
        const bf16x8* kt = kptr(it);
#pragma unroll
        for (int s = 0; s < 8; ++s) kA[s] = kt[s * 64 + lane];
        const bf16x8* vt0 = vptr(it);
#pragma unroll
        for (int s = 0; s < 8; ++s) vE[s] = vt0[s * 64 + lane];
    }
    qs[par * 64 + lane] = qmine;
    __syncthreads();
    if (stamp && lane == 0) stamp[1] = wall_clock64();
    auto next_tile = [&]() { idx = idx + 1 == cnt ? 0 : idx + 1; return par + 8 * idx; };
    if (cnt > 0) {
        // the first tile is peeled: its V comes from vE (a compile-time choice inside tile(): no merge of two V sources)
        int nx = 0;
        if (1 < cnt) {
            nx = next_tile();
            const bf16x8* kt = kptr(nx);
#pragma unroll
            for (int s = 0; s < 8; ++s) kB[s] = kt[s * 64 + lane];
        }
        tile(it, kA, std::true_type{});
        if (stamp && lane == 0) stamp[2] = wall_clock64();
        it = nx;
        for (int k = 1; k < cnt; k += 2) {
            if (k + 1 < cnt) {
                nx = next_tile();
                const bf16x8* kt = kptr(nx);
#pragma unroll
                for (int s = 0; s < 8; ++s) kA[s] = kt[s * 64 + lane];
            }
            tile(it, kB, std::false_type{});
            if (k + 1 >= cnt) break;
            it = nx;
            if (k + 2 < cnt) {
                nx = next_tile();
                const bf16x8* kt = kptr(nx);
#pragma unroll
                for (int s = 0; s < 8; ++s) kB[s] = kt[s * 64 + lane];
            }
            tile(it, kA, std::false_type{});
            it = nx;
        }
    }
    if (stamp && lane == 0) stamp[3] = wall_clock64();

    // ---- the 8 key parities meet once: the lanes of the slice's token columns park (O, m, l); then every (head-dim group of 4,
    //      token, half) item is merged in the fixed order p = 0..7 (deterministic), normalised, rounded to bf16 (attn_output
    //      dtype) and stored as 8 bytes of o_proj's packed operand
    f32x4* const mg4 = (f32x4*)(lds1 + 2048);                         // [par][db * 4 + i4][2 W] float4
    float* const ml = lds1 + 2048 + 8 * 16 * 2 * W * 4;               // [par][W] {m, l}
    {
        const int col = lane & 31;
        if (col / W == sl) {
            const int j = (col % W) + W * hh;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const f32x4 v = {o[db][4 * i4], o[db][4 * i4 + 1], o[db][4 * i4 + 2], o[db][4 * i4 + 3]};
                    mg4[((par * 16) + db * 4 + i4) * (2 * W) + j] = v;
                }
            if (hh == 0) { ml[(par * W + col % W) * 2] = m; ml[(par * W + col % W) * 2 + 1] = l; }
        }
    }
    __syncthreads();
    for (int item = threadIdx.x; item < 16 * 2 * W; item += 512) {
        const int jj = item % (2 * W), q16 = item / (2 * W);
        const int tl = jj % W, h2 = jj / W;
        float wp[8];
        float M = LA_NEG, L = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) { wp[p] = ml[(p * W + tl) * 2]; M = fmaxf(M, wp[p]); }
#pragma unroll
        for (int p = 0; p < 8; ++p) { wp[p] = __expf(wp[p] - M); L += ml[(p * W + tl) * 2 + 1] * wp[p]; }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 8; ++p) acc += mg4[(p * 16 + q16) * (2 * W) + jj] * wp[p];
        const float inv = 1.0f / L;
        const int tok = tb * 32 + sl * W + tl;
        const int d = (q16 >> 2) * 32 + 8 * (q16 & 3) + 4 * h2;
        const bf16x4 ov = {(short)f2bf(acc[0] * inv), (short)f2bf(acc[1] * inv), (short)f2bf(acc[2] * inv), (short)f2bf(acc[3] * inv)};
        *(bf16x4*)(a.attn_xp + xp_offset(tok, h * 128 + d)) = ov;
    }
    if (stamp && lane == 0) stamp[4] = wall_clock64();
}

// Rider workgroup b: 16 x 16-byte loads per thread over the first bytes consumer workgroup b of the NEXT launch (o_proj) streams
// (PfDesc, la_kernels.h); default cache policy, data dropped; the wave retires when they have landed in this XCD's L2.
__device__ __forceinline__ void attn1_rider(const PfDesc& p, int b) {
    if (b >= p.n_consumers) return;
    for (int i = 0; i < p.delay; ++i) __builtin_amdgcn_s_sleep(32);        // lab knob 32: let the attention's own first (HBM) tiles go first
    const int bx = b % p.nbx, ks = b / p.nbx;
    const char* start = p.base + (size_t)bx * p.A + (size_t)ks * p.A2;
    const unsigned n0 = p.L[0] >> 4, n1 = p.RB > 1 ? p.L[1] >> 4 : 0u;          // classic images: RB <= 2
    const unsigned cps = n0 + n1, total = cps * (unsigned)p.NW;
    f32x4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        unsigned i = threadIdx.x + (unsigned)j * 512u;
        i = i < total ? i : total - 1u;
        const unsigned w = i / cps, r = i - w * cps;
        const bool second = r >= n0;
        const char* addr = start + (second ? p.boff[1] : p.boff[0]) + (size_t)w * (second ? p.C[1] : p.C[0]) + (size_t)(second ? r - n0 : r) * 16;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[j]) : "v"(addr));
    }
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
                 "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]) : "memory");
}


static int g_attn1_cus = 0;
// one-off set-up, called from lk_gemm64r_init (never inside a stream capture): dynamic-LDS limit of the SL = 1 form, CU count
int lk_attn1_init() {
    if (g_attn1_cus) return 0;
    if (hipFuncSetAttribute((const void*)k_tree_attn1<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 + 8 * 16 * 64 * 16 + 8 * 32 * 8) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void*)k_tree_attn1<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 + 8 * 16 * 64 * 16 + 8 * 32 * 8) != hipSuccess) return -1;
    hipDeviceProp_t p; int dev = 0;
    g_attn1_cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    return 0;
}
int lk_tree_attn1(hipStream_t st, const void* qf, const void* kmain, const void* vmain, const void* kfresh, const void* vfresh,
                  const uint64_t* rowmask, const int* state, int nh, int nkv, int max_keys, void* attn_xp, int window, int ring_keys,
                  const PfDesc* pf, int head_dim) {
    if (nh <= 0 || nkv <= 0 || nh % nkv || nh > 0x7fff || (ring_keys >> 5) >= (1 << 22)) return -1;
    if (lk_gemm64r_init() != 0) return -1;
    // token slices per 32-row block: 2 (measured: every slice count streams the same bytes per CU — all of the head's K/V — and
    // two slices leave the least redundant L2 traffic at equal or better time, profiles/r04_attention_one_launch.txt); fewer when
    // nh * 2 * SL workgroups would not fit one per CU
    int SL = 2;
    while (SL > 1 && nh * 2 * SL > g_attn1_cus) SL >>= 1;
    if ((g_la_attn1_var >> 1) & 3) SL = 1 << (((g_la_attn1_var >> 1) & 3) - 1);
    const int W = 32 / SL;
    Attn1Args a{};
    a.kfresh = (const bf16_t*)kfresh; a.vfresh = (const bf16_t*)vfresh; a.attn_xp = (bf16_t*)attn_xp; a.dbg_times = g_la_dbg_times;
    a.qk = la_qk_scale(head_dim);
    const size_t lds = 8192 + (size_t)8 * 16 * 2 * W * 16 + (size_t)8 * W * 8;
    a.n_main = nh * 2 * SL;
    int riders = 0;
    if (pf && pf->base && pf->n_consumers > 0 && pf->RB <= 2 && (a.n_main & 7) == 0) { a.pf = *pf; riders = pf->n_consumers; }
    if (riders)
        k_tree_attn1<true><<<a.n_main + riders, 512, lds, st>>>((const bf16_t*)qf, (const unsigned long long*)rowmask, state, (const bf16_t*)kmain,
                                                (const bf16_t*)vmain, max_keys, (nh << 16) | nkv, window, SL | ((g_la_attn1_var & 1) << 7) | ((ring_keys >> 5) << 8), a);
    else
        k_tree_attn1<false><<<a.n_main, 512, lds, st>>>((const bf16_t*)qf, (const unsigned long long*)rowmask, state, (const bf16_t*)kmain,
                                                (const bf16_t*)vmain, max_keys, (nh << 16) | nkv, window, SL | ((g_la_attn1_var & 1) << 7) | ((ring_keys >> 5) << 8), a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
