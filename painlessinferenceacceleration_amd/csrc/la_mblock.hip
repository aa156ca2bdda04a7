// la_mblock.hip — multi-block verify step for gfx950: M = nblk x 64 rows (nblk <= 8) through the SAME packed weights.
//
// Why more GEMM families.  At M = 64 (la_kernels.hip) a weight byte is used 64 times: the launch is a pure HBM stream,
// waves split K and every wave keeps its own x fragments in registers.  At M = 128..512 (configs 3-5 of BASELINE.json:
// bs 4..8 x 64-token trees; prompt prefill) the same bytes feed 2-8x the MFMAs and x no longer fits a wave's registers.
//   k_gemm_mb   (M <= 128, and the gathered experts of Mixtral): the 8 waves own different 32-row weight blocks x K parts, the x
//               tile of a stage is staged once per workgroup in LDS (register-staged double buffer), weight fragments stream
//               HBM -> VGPR in MFMA operand order; K parts reduced through LDS.
//   k_gemm_wide (M = 192..512): one pass over the weights for ALL rows; both operands by LDS-DMA through one ring of stages,
//               every wave two row-blocks x TW token blocks over the whole K range, epilogues from the accumulators.
// Mixtral at M >= 128 gathers the rows of every expert into their own blocks (k_moe_plan_mb / k_moe_gather_mb) instead of
// running every expert over all rows.
// Reference semantics: modeling_llama_batch.py:340-420 (batched forward), pretrained_model_batch.py:706-931 (per-sample
// draft, accept, in-place KV) with the per-sample 64-token tree SURVEY H2 / BASELINE configs 3-5 ask for.
#include <type_traits>
#include "la_common.h"
#include "la_mblock.h"
#include "la_knobs.h"

#define LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

enum { MB_SLAB = 0, MB_SWIGLU = 1, MB_QKV = 2, MB_LOGITS = 3 };

// k-tile range [t0, t1) of K split ks.  An even K16 is cut on EVEN k-tiles (round 5): every multi-block GEMM family uses the same
// boundaries (so their split-K slabs stay bit-identical to each other), and the fat-wave kernels, which consume whole 2-k-tile stages,
// take every shape with an even K16 — e.g. Llama-2-13B's o_proj, 320 k-tiles in 3 splits: 106 / 108 / 106 instead of 106 / 107 / 107.
__device__ __forceinline__ void mb_k_range(int K16, int ks, int ksplit, int& t0, int& t1) {
    if ((K16 & 1) == 0) {
        const long h = K16 >> 1;
        t0 = 2 * (int)((h * ks) / ksplit); t1 = 2 * (int)((h * (ks + 1)) / ksplit);
    } else {
        t0 = (int)(((long)K16 * ks) / ksplit); t1 = (int)(((long)K16 * (ks + 1)) / ksplit);
    }
}

struct MbArgs {
    const bf16_t* wp;
    const bf16_t* xp;
    int K16;                 // k-tiles of the whole reduction dimension (x image: K16*1024 elements per 64-row block)
    int N;
    int M;                   // rows of the slab buffer per K split (slab stride)
    int nblk;                // 64-row blocks of the step that carry data
    // weight addressing: planned (workgroup-major virtual blocks, la_pack_planned) or classic (la_pack_weight)
    int planned, R;
    int gu_interleaved;      // MB_SWIGLU on a classic image: row-blocks alternate gate/up (la_pack_weight interleave2)
    int nv[8], nvl[8], boff[8], wg_chunks;       // entries 4..7: second region of a paired gate/up launch (RBV = 8)
    float* slabs;            // MB_SLAB: [ksplit][M][N]
    bf16_t* act_xp;          // MB_SWIGLU: [blk][64 x N packed]
    bf16_t* logits;          // MB_LOGITS: [M][N] row-major bf16 (may be null)
    float* cand_val;         // MB_LOGITS: [blk][gridDim.x * 4][64]
    int* cand_idx;
    const int* pos; const bf16_t* rcos; const bf16_t* rsin;      // MB_QKV
    bf16_t* qf; bf16_t* kfresh; bf16_t* vfresh; int nh, nkv;
    // mixture-of-experts: routing weights of THIS expert, one float per row at stride LA_MOE_MAX_E; when no row of the step routes
    // to the expert the launch returns at once and its weights are never read
    const float* route_col; int route_rows;
    // gathered MoE (k_moe_plan_mb): the number of 64-row blocks this expert received, decided on the device
    const int* nblk_dev;
    // ... for ALL experts of a stage in one launch (ex_n > 1): grid.z = expert x pass; operands of expert e at e * stride
    int ex_n; long ex_w_stride, ex_x_stride, ex_o_stride;
    int ex_pad;              // measurement (la_debug_set(16, 2)): keep the padded two-block pass for an expert's last single block
    // paired form of the wide kernel (launch_mb): the weight rows of a workgroup are read by a second workgroup working on the other
    // token blocks (same XCD): stream them with the default cache policy so that the second reader finds them in L2
    int w_keep;
    // k_gemm_fat slab launches with 2 / 4 / 8 K splits (round 5, launch_mb): workgroup -> tile mapping that gives every XCD ONE K split (8 / ksplit
    // XCDs per split), so an XCD's L2 holds a 1 / ksplit slice of x instead of all of it
    int xcd_map;
    long long* dbg_times;    // measurement build DBG = 6: [workgroup][wave][8] accumulated shader cycles per loop segment
};

// ---------------------------------------------------------------------------------------------------------------
// Epilogue of one 64-row block: the workgroup's RBV x 2 accumulator tiles sit in LDS as KP partial sums
// red4[(((p * RBV + rb) * 2 + tb) * 4 + i/4) * 64 + lane] (16-byte units); each of the 8 waves finishes a fixed set of
// (row-block, token block, register group) slices, parts summed in the order p = 0..KP-1 (deterministic).
// ---------------------------------------------------------------------------------------------------------------
template <int RBV, int EPI, int KP>
__device__ __forceinline__ void mb_epilogue_block(const MbArgs& a, const f32x4* red4, int blk, int ks, int wave, int lane, size_t o_off = 0) {
    const int tl = lane & 31, hh = lane >> 5;
    auto total4 = [&](int rbq, int tb, int gi) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < KP; ++p) v += red4[(((p * RBV + rbq) * 2 + tb) * 4 + gi) * 64 + lane];
        return v;
    };
    if constexpr (EPI == MB_SLAB) {
        static_assert(EPI != MB_SLAB || RBV == 2 || RBV == 4, "slab layout");
        // slices (rb, tb, gi): 8 RBV -> RBV per wave (RBV = 4: the paired form of the merged-expert launch, two 64-row regions per workgroup)
#pragma unroll
        for (int i = 0; i < RBV; ++i) {
            const int sl = wave * RBV + i, rq = sl >> 3, tb = (sl >> 2) & 1, gi = sl & 3;
            const f32x4 v = total4(rq, tb, gi);
            const int tok = blk * 64 + tb * 32 + tl;
            float* o = a.slabs + o_off + ((size_t)ks * a.M + tok) * a.N + (blockIdx.x * RBV + rq) * 32 + 8 * gi + 4 * hh;
            *(f32x4*)o = v;
        }
    } else if constexpr (EPI == MB_SWIGLU) {
        static_assert(EPI != MB_SWIGLU || RBV == 4 || RBV == 8, "swiglu layout {G0,G1,U0,U1} (x 2 regions: the paired merged-expert launch)");
        // pair slices (region, q, tb, gi): 4 RBV -> RBV / 2 per wave;  act = bf16(silu(bf16(g)) * bf16(u)) (LlamaMLP.forward, :185-186)
        constexpr int NREG = RBV / 4;
#pragma unroll
        for (int i = 0; i < RBV / 2; ++i) {
            const int sl = wave * (RBV / 2) + i, qa = sl >> 3, tb = (sl >> 2) & 1, gi = sl & 3;
            const int region = qa >> 1, qq = NREG == 1 ? qa : (qa & 1);
            const int rg = (NREG == 1 && a.gu_interleaved) ? 2 * qq : 4 * region + qq, ru = (NREG == 1 && a.gu_interleaved) ? 2 * qq + 1 : rg + 2;
            if (8 * gi + 4 * hh < a.nv[rg]) {
                const f32x4 g4 = total4(rg, tb, gi), u4 = total4(ru, tb, gi);
                const int tok = tb * 32 + tl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = 8 * gi + 4 * hh + j;
                    if (f < a.nv[rg]) {
                        const float gv = bfr(g4[j]), uv = bfr(u4[j]);
                        const float sv = bfr(gv / (1.0f + expf(-gv)));
                        const int feat = (NREG == 1 && a.gu_interleaved) ? (2 * blockIdx.x + qq) * 32 + f : a.R * (NREG * blockIdx.x + region) + 32 * qq + f;
                        a.act_xp[o_off + (size_t)blk * 64 * a.N + xp_offset(tok, feat)] = f2bf(sv * uv);
                    }
                }
            }
        }
    } else if constexpr (EPI == MB_QKV) {
        static_assert(EPI != MB_QKV || RBV == 2, "qkv layout {lo, hi}");
        // pair slices (tb, gi): 8 -> 1 per wave; RoPE in bf16 arithmetic (apply_rotary_pos_emb, modeling_llama.py:154-169)
        const int tb = wave >> 2, gi = wave & 3;
        const int tok = tb * 32 + tl;
        if (8 * gi + 4 * hh < a.nv[0]) {
            const f32x4 xl4 = total4(0, tb, gi), xh4 = total4(1, tb, gi);
            const int ps = a.pos[blk * 64 + tok];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = 8 * gi + 4 * hh + j;
                if (f < a.nv[0]) {
                    const int prr = a.R * blockIdx.x + f, slot = prr >> 6, dlo = prr & 63, dhi = dlo + 64;
                    const float xl = xl4[j], xh = xh4[j];
                    if (slot < a.nh + a.nkv) {
                        bf16_t* dst = slot < a.nh ? a.qf + ((size_t)blk * a.nh + slot) * 8192
                                                  : a.kfresh + ((size_t)blk * a.nkv + (slot - a.nh)) * 8192;
                        const float cc = bf2f(a.rcos[(size_t)ps * 64 + dlo]), sn = bf2f(a.rsin[(size_t)ps * 64 + dlo]);
                        const float bl = bfr(xl), bh = bfr(xh);
                        dst[rf_offset(tok, dlo)] = f2bf(bfr(bl * cc) + bfr(-bh * sn));
                        dst[rf_offset(tok, dhi)] = f2bf(bfr(bh * cc) + bfr(bl * sn));
                    } else {
                        bf16_t* dst = a.vfresh + ((size_t)blk * a.nkv + (slot - a.nh - a.nkv)) * 8192;
                        dst[vf_offset(tok, dlo)] = f2bf(xl);
                        dst[vf_offset(tok, dhi)] = f2bf(xh);
                    }
                }
            }
        }
    } else {
        static_assert(EPI != MB_LOGITS || RBV == 4, "logits layout");
        // wave -> token block w&1, row-block w>>1, all four register groups; one argmax candidate per (wave, token)
        const int tb = wave & 1, rq = wave >> 1;
        const int tok = tb * 32 + tl;
        float best = -INFINITY;
        int bidx = 0x7fffffff;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            if (8 * gi + 4 * hh < a.nv[rq]) {
                const f32x4 t4 = total4(rq, tb, gi);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = 8 * gi + 4 * hh + j;
                    if (f < a.nv[rq]) {
                        const bf16_t hv = f2bf(t4[j]);
                        const int idx = a.R * blockIdx.x + 32 * rq + f;
                        if (a.logits) a.logits[((size_t)blk * 64 + tok) * a.N + idx] = hv;
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
            const size_t slot = ((size_t)blk * gridDim.x + blockIdx.x) * 4 + rq;
            a.cand_val[slot * 64 + tok] = best;
            a.cand_idx[slot * 64 + tok] = bidx;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// out[M][N] = x[M][K] . W[N][K]^T for NT/2 64-row blocks per pass (grid.z = pass).
//   workgroup = 8 waves = RBV weight row-blocks x KP = 8/RBV K-parts; wave (rb, kp) streams row-block rb over part kp
//   (weight tiles D deep in registers, nontemporal), a stage = 4 k-tiles (KT = 4/KP per part) whose x fragments
//   (4*NT KiB) sit in an LDS double buffer; per stage and wave: KT x NT MFMAs.  K-parts are summed through LDS in a fixed
//   order (deterministic) per 64-row block, then the same epilogues as the single-block kernels.
// ---------------------------------------------------------------------------------------------------------------
// body of k_gemm_mb for the NT / 2 blocks blk0 .. of a pass (nblk = the blocks that exist; operands offset for the merged-expert form)
// D = weight tiles in flight per wave: 8 for the dense launches (one workgroup per CU); 4 for the merged-expert launches, whose
// register footprint then admits TWO workgroups per CU (<= 128 VGPRs): the same 64 KiB of weights in flight per CU, and the ramp / drain
// of one workgroup (x sets + first tiles; K-part reduction + epilogue: ~3 of its ~23 us) runs under the other one's stream.
template <int RBV, int NT, int EPI, int D = 8>
__device__ __forceinline__ void gemm_mb_body(const MbArgs& a, const int blk0, const int nblk, const size_t w_off, const size_t x_off,
                                             const size_t o_off) {
    constexpr int KP = 8 / RBV, KT = 4 / KP;
    constexpr int FR = 4 * NT;                  // x fragments (1 KiB) per stage
    constexpr int FPW = FR / 8;                 // fragments staged by one wave
    constexpr int SPG = D / KT;                 // stages per unrolled group (weight ring slots are compile-time inside it)
    static_assert(FPW >= 1 && (FR % 8) == 0, "stage split");
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    bf16x8* xs = (bf16x8*)lds_raw;              // [2 stages][FR][64 lanes]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rb = wave % RBV, kp = wave / RBV;
    const int ksplit = gridDim.y, ks = blockIdx.y;
    int t0, t1;
    mb_k_range(a.K16, ks, ksplit, t0, t1);
    const int twg = t1 - t0, pq = twg / KP, pr = twg - pq * KP;
    const int my_start = t0 + kp * pq + (kp < pr ? kp : pr), my_cnt = pq + (kp < pr ? 1 : 0);
    const int nstages = (pq + (pr > 0 ? 1 : 0) + KT - 1) / KT;

    // ---- weight fragment addressing (16-byte chunks)
    const bf16x8* __restrict__ wbase = (const bf16x8*)(a.wp + w_off);
    unsigned woff, wstr;
    int nvalid = 32;
    if (a.planned) {
        const int nvb = a.nvl[rb];
        const int rr = (lane & 31) < nvb ? (lane & 31) : nvb - 1;
        wstr = (unsigned)(2 * nvb);
        woff = (unsigned)blockIdx.x * (unsigned)a.wg_chunks + (unsigned)a.boff[rb] + (unsigned)my_start * wstr + (unsigned)((lane >> 5) * nvb + rr);
        nvalid = a.nv[rb];
    } else {
        wstr = 64u;
        woff = (unsigned)(((blockIdx.x * RBV + rb) * a.K16 + my_start) * 64 + lane);
    }
    // ---- x fragments staged by this wave: f = wave + 8*i -> (part, k-tile in stage, token block)
    const bf16x8* __restrict__ xbase = (const bf16x8*)(a.xp + x_off);
    int xpart_start[FPW], xpart_cnt[FPW], xj[FPW];
    unsigned xconst[FPW];
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
        const int f = wave + 8 * i, part = f / (KT * NT), j = (f / NT) % KT, tbk = f % NT;
        xpart_start[i] = t0 + part * pq + (part < pr ? part : pr);
        xpart_cnt[i] = pq + (part < pr ? 1 : 0);
        xj[i] = j;
        int xb = blk0 + (tbk >> 1);
        xb = xb < nblk ? xb : nblk - 1;             // surplus blocks of a padded pass re-read the last real block (results dropped)
        xconst[i] = (unsigned)(((xb * a.K16) * 2 + (tbk & 1)) * 64 + lane);
    }
    auto xload = [&](int s, bf16x8 (&xr)[FPW]) {
#pragma unroll
        for (int i = 0; i < FPW; ++i) {
            const int kl = s * KT + xj[i];
            const int kk = kl < xpart_cnt[i] ? kl : 0;                    // past the end: any valid tile (its MFMAs are skipped)
            xr[i] = xbase[xconst[i] + (unsigned)(xpart_start[i] + kk) * 128u];
        }
    };
    auto xstore = [&](int slot, const bf16x8 (&xr)[FPW]) {
#pragma unroll
        for (int i = 0; i < FPW; ++i) xs[(slot * FR + wave + 8 * i) * 64 + lane] = xr[i];
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    if (nstages > 0) {
        // x fragments are prefetched XD stages ahead in registers.  vmcnt retires in order, so a wait on an x set also waits for
        // every OLDER load: with the sets issued XD = D/KT stages before their ds_write, the loads younger than the awaited set
        // are XD-1 x sets and exactly D weight tiles — the weight ring stays D deep across every stage boundary (with a 1-stage
        // x prefetch each stage end drained the weight queue to KT tiles: 3x slower, profiles/r02_mblock_kernel_stats_v1.txt).
        // (D = 4 with one tile per stage and part — the two-per-CU down launch — prefetches 2 sets: 4 would put it past 128 VGPRs)
        constexpr int XD = (D == 4 && KT == 1) ? 2 : (D / KT >= 4 ? 4 : D / KT);
        static_assert(SPG % XD == 0 && (SPG % 2) == 0, "static ring indices inside an unrolled group");
        bf16x8 fa[D];
        bf16x8 xr[XD][FPW];
#pragma unroll
        for (int x = 0; x < XD; ++x) xload(x, xr[x]);                      // stages 0..XD-1 (clamped past the end)
#pragma unroll
        for (int d = 0; d < D; ++d) fa[d] = __builtin_nontemporal_load(wbase + woff + (unsigned)(d < my_cnt ? d : 0) * wstr);
        xstore(0, xr[0]);
        xload(XD, xr[0]);
        __syncthreads();
        // Full groups: every part has all D tiles of the group -> branch-free body (tile indices past a part's end only occur in
        // refills and x prefetches, where they are clamped to a valid tile), order pinned with sched_barrier so that hipcc
        // keeps the ring slots in fixed registers and emits counted s_waitcnt vmcnt(n) instead of draining the queue.
        const int gfull = pq / D;
        for (int g = 0; g < gfull; ++g) {
#pragma unroll
            for (int u = 0; u < SPG; ++u) {
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    const int slotw = u * KT + j;
                    const bf16x8* xt = xs + (((u & 1) * FR) + (kp * KT + j) * NT) * 64 + lane;
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[t] = LA_MFMA(fa[slotw], xt[t * 64], acc[t], 0, 0, 0);
                    const int nx = g * D + slotw + D;
                    fa[slotw] = __builtin_nontemporal_load(wbase + woff + (unsigned)(nx < my_cnt ? nx : 0) * wstr);
                    __builtin_amdgcn_sched_barrier(0);
                }
                xstore((u + 1) & 1, xr[(u + 1) % XD]);
                xload(g * SPG + u + 1 + XD, xr[(u + 1) % XD]);
                __syncthreads();
            }
        }
        // Tail group (< D tiles per part, parts may differ by one tile): same ring positions, every step guarded.
        {
            const int sg = gfull * SPG;
#pragma unroll
            for (int u = 0; u < SPG; ++u) {
                const int s = sg + u;
                if (s < nstages) {                                         // workgroup-uniform
#pragma unroll
                    for (int j = 0; j < KT; ++j) {
                        const int kl = s * KT + j;
                        const int slotw = u * KT + j;
                        if (kl < my_cnt) {                                 // wave-uniform
                            const bf16x8* xt = xs + (((u & 1) * FR) + (kp * KT + j) * NT) * 64 + lane;
#pragma unroll
                            for (int t = 0; t < NT; ++t)
                                acc[t] = LA_MFMA(fa[slotw], xt[t * 64], acc[t], 0, 0, 0);
                        }
                    }
                    xstore((u + 1) & 1, xr[(u + 1) % XD]);
                    xload(s + 1 + XD, xr[(u + 1) % XD]);
                    __syncthreads();
                }
            }
        }
    }

    // ---- epilogue, one 64-row block at a time: K-parts parked in LDS [kp][rb][tb][i/4][lane] (16-byte units), one barrier,
    //      fixed summation order p = 0..KP-1, every wave finishes a fixed set of (row-block, token block, register group) slices
    f32x4* red4 = (f32x4*)lds_raw;
    (void)nvalid;
#pragma unroll
    for (int c = 0; c < NT / 2; ++c) {
        const int blk = blk0 + c;
        if (blk >= nblk) break;                     // surplus blocks of a padded pass (workgroup-uniform)
        if (c) __syncthreads();
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 w4 = {acc[2 * c + tb][4 * i4], acc[2 * c + tb][4 * i4 + 1], acc[2 * c + tb][4 * i4 + 2], acc[2 * c + tb][4 * i4 + 3]};
                red4[(((kp * RBV + rb) * 2 + tb) * 4 + i4) * 64 + lane] = w4;
            }
        __syncthreads();
        mb_epilogue_block<RBV, EPI, KP>(a, red4, blk, ks, wave, lane, o_off);
    }
}

// DW = weight tiles in flight per wave (see gemm_mb_body): 4 = the two-workgroups-per-CU form of the merged-expert launches
template <int RBV, int NT, int EPI, bool EX = false, int DW = 8>
__global__ __launch_bounds__(512, (DW == 4 ? 4 : 1)) void k_gemm_mb(MbArgs a) {
    // EX: every expert of a gathered MoE stage in ONE launch (round 3; one launch per expert before — 16 launches per layer, each
    // with its own ramp and drain): grid.z = expert x pass, the operands of expert e sit e strides further, and the workgroups of
    // expert e + 1 start while the last ones of expert e finish.  The argument block stays untouched (a mutable copy of it cost
    // the dense instantiations their scalar-register residency: Mixtral bs=4 24 -> 35 ms).
    int zpass = blockIdx.z;
    size_t w_off = 0, x_off = 0, o_off = 0;
    const int* nbd = a.nblk_dev;
    if constexpr (EX) {
        const int npass = (int)gridDim.z / a.ex_n, e = (int)blockIdx.z / npass;
        zpass = (int)blockIdx.z - e * npass;
        w_off = (size_t)e * a.ex_w_stride; x_off = (size_t)e * a.ex_x_stride; o_off = (size_t)e * a.ex_o_stride; nbd += e;
    }
    if (a.route_col) {                          // every wave scans all rows: a uniform decision without a barrier
        const int lane = threadIdx.x & 63;
        bool any = false;
        for (int t = lane; t < a.route_rows; t += 64) any |= a.route_col[(size_t)t * LA_MOE_MAX_E] != 0.f;
        if (__ballot(any) == 0ull) return;
    }
    const int blk0 = zpass * (NT / 2);
    const int nblk = nbd ? __builtin_amdgcn_readfirstlane(*nbd) : a.nblk;
    if (blk0 >= nblk) return;                   // a pass with no block of this expert: its weights are never read
    if constexpr (EX && NT == 4) {
        // the LAST block of an expert alone in a two-block pass (about half the experts of a 256-row Mixtral step hold <= 64 rows):
        // the one-block body — half the x traffic, LDS stores and MFMAs of the padded pass
        if (nblk - blk0 == 1 && !a.ex_pad) { gemm_mb_body<RBV, 2, EPI, DW>(a, blk0, nblk, w_off, x_off, o_off); return; }
    }
    gemm_mb_body<RBV, NT, EPI, DW>(a, blk0, nblk, w_off, x_off, o_off);
}

// ---------------------------------------------------------------------------------------------------------------
// The wide form for nblk >= 3 (192..512 rows): ALL token blocks of the step in one weight pass, both operands through LDS,
// no K parts, no reduction, epilogues straight from the accumulators.
//   workgroup = 8 waves = RG row groups x TQ token groups (RBV = 4: 2 x 4, RBV = 2: 1 x 8); every wave owns TWO weight
//   row-blocks x TW token blocks (2 TW accumulator tiles) over the workgroup's whole K range.  The two row-blocks of a wave
//   are the ones its epilogue combines — gate and up rows of the same features, the lo and hi halves of a RoPE pair — so no
//   epilogue needs another wave's sums.
//   Operands arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no staging registers).  A stage =
//   KS = 2 k-tiles = A_STAGE weight pieces + B_STAGE x pieces in the order the fragments are read back; wave w copies the
//   pieces w + 8 i.  vmcnt retires in order per wave, so both operands ride ONE 3-stage ring and a wave's outstanding DMA
//   is always "the pieces of the younger stages": s_waitcnt vmcnt(pieces per stage), never 0 inside the loop.
//   Stage s is consumed as two k-tiles out of two register sets; the barrier that publishes stage s+1 sits BETWEEN them:
//     read set1 <- (s, k-tile 1) | MFMAs of set0 | lgkmcnt(0): every read of stage s is complete | own DMA of s+1 landed |
//     s_barrier | read set0 <- (s+1, k-tile 0) | MFMAs of set1, one DMA issue of stage s+3 (into the slot the barrier has
//     just freed) after each — every ds_read latency runs under the MFMAs of the other set, every DMA issue between MFMAs.
//   Fragment reads are ds_read_b128, lane-linear (conflict-free): (2 + TW) per k-tile for 2 TW MFMAs.
// Measured anatomy (gate/up, 512 rows, profiles/r02_mblock_wide_parts.txt): MFMA issue floor 68 us at the 2.0 PFLOP/s the
// chip sustains, DMA-only loop 67 us (x from L2 and weights from HBM share the CU's in-order vector-memory path).
// ---------------------------------------------------------------------------------------------------------------
// float -> bf16 on the gfx950 converter (v_cvt_pk_bf16_f32: round-to-nearest-even like f2bf, one instruction instead of six)
__device__ __forceinline__ bf16_t f2bf_hw(float f) { return f2bf(f); }
__device__ __forceinline__ float bfr_hw(float f) { return bf2f(f2bf_hw(f)); }

template <int N> __device__ __forceinline__ void vm_wait() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// one LDS-DMA piece through a buffer resource: 16 B per lane at (voffset + soffset) of the buffer -> LDS at d + 16 lane
template <int AUX>
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rs, char* d, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)d, 16, voff, soff, 0, AUX);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dma_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7ffffff0, 0x00020000);
}

template <int RBV, int TW, int NRG = 4> struct WideGeom {
    static constexpr int KS = 2, RG = RBV / 2, TQ = 8 / RG, NTBP = TQ * TW;  // token blocks (32 rows) per workgroup
    // RBV = 8: the paired gate/up launch — two adjacent planned regions {G0,G1,U0,U1} x 2 per workgroup, 4 row groups x 2 token groups
    static constexpr int A_STAGE = KS * RBV, B_STAGE = KS * NTBP;           // 1 KiB pieces per stage
    static constexpr int STAGE = A_STAGE + B_STAGE;
    static constexpr int NR = NRG;                                            // ring slots (stages): NR - 1 stages in flight ahead of the one consumed
    static constexpr int NP_HI = (STAGE + 7) / 8, NP_LO = STAGE / 8;          // pieces per wave and stage
    static constexpr int N_HI = STAGE - 8 * NP_LO;                            // waves 0..N_HI-1 carry NP_HI pieces
    static constexpr int H = (NP_HI + 1) / 2;                                 // pieces issued in the first half of a stage
    static constexpr int LDS = NR * STAGE * 1024;
    static constexpr int BLOCKS = NTBP / 2;                                   // 64-row blocks per workgroup
    static_assert(A_STAGE <= 16 && LDS <= 160 * 1024 && (RBV == 2 || RBV == 4 || RBV == 8), "ring geometry");
};

// SCH = schedule of a stage (bit 0: reads interleaved, bit 1: buffer-addressed pieces).  0: the six fragment reads of a k-tile are issued together, then its MFMAs; the pieces are addressed
// by 64-bit per-lane pointers (global_load_lds).  Bit 0: ONE fragment read after each MFMA (the matrix pipe restarts right behind every
// barrier instead of behind six ds_read issues and their address adds).  Bit 1: the pieces are addressed through two buffer resources
// (weights of this workgroup, x) with a per-lane 32-bit offset and a scalar k-tile offset: buffer_load_dwordx4 ... offen lds, ~5 SALU
// and no VALU per piece instead of ~9 SALU + a 64-bit VALU add.  Same MFMAs on the same operands in the same order: bit-identical.
// (Measured and dropped: four loader waves — one per SIMD — issuing all pieces of a stage: 5-12 % SLOWER,
// profiles/r04_wide_gemm_schedule.txt: a wave's own issue chain is the cost, not the queueing of the eight waves.)
template <int RBV, int TW, int EPI, int DBG = 0, int SCH = 0, int NRG = 4>
__global__ __launch_bounds__(512) void k_gemm_wide(MbArgs a) {
    using GEO = WideGeom<RBV, TW, NRG>;
    static_assert(NRG == 4 || NRG == 3, "ring depth");
    constexpr int KS = GEO::KS, TQ = GEO::TQ, NTBP = GEO::NTBP, A_STAGE = GEO::A_STAGE, STAGE = GEO::STAGE, NR = GEO::NR;
    constexpr int NP_HI = GEO::NP_HI, NP_LO = GEO::NP_LO, N_HI = GEO::N_HI;
    static_assert(SCH == 0 || ((SCH == 2 || SCH == 3) && (DBG == 0 || DBG == 4 || DBG == 6)), "measurement builds 1, 2, 3, 5 exist for schedule 0 only");
    constexpr bool BUF = (SCH & 2) != 0;            // pieces addressed through buffer resources
    constexpr bool ILV = (SCH & 1) != 0;            // one fragment read after each MFMA
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int lane = threadIdx.x & 63;
    if (a.route_col) {
        bool any = false;
        for (int t = lane; t < a.route_rows; t += 64) any |= a.route_col[(size_t)t * LA_MOE_MAX_E] != 0.f;
        if (__ballot(any) == 0ull) return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // row group = wave / 4: a workgroup's waves are dealt to the SIMDs cyclically, so each SIMD gets one wave of either row group
    // (the row groups' epilogues differ in weight: a planned gate/up image has 32 valid rows in one pair of blocks, R - 32 in the other)
    const int rg = wave / TQ, tq = wave % TQ;        // RBV = 2: 1 x 8, RBV = 4: 2 x 4, RBV = 8: 4 row groups x 2 token groups
    static_assert(TQ == (RBV == 8 ? 2 : RBV == 4 ? 4 : 8) && (EPI == MB_SWIGLU || EPI == MB_QKV || RBV != 8), "wave grid");
    // the wave's two row-blocks: planned gate/up images are {G0, G1, U0, U1} (pair = rb, rb + 2), classic interleaved images
    // {G, U, G, U} (pair = 2 rg, 2 rg + 1); lm_head rows are independent; RBV = 2 images are {lo, hi} / two plain blocks
    // RBV = 8 (planned gate/up only): row groups 0, 1 work on region 2 x, row groups 2, 3 on region 2 x + 1 (blocks 4..7)
    // RBV = 8 with MB_QKV ("quad" form): FOUR adjacent {lo, hi} regions per workgroup, row group rg = region 4 x + rg, a quarter of
    // the token blocks per workgroup (grid.z = 4)
    const int rb0 = (RBV == 8 && EPI == MB_QKV) ? 2 * rg : RBV == 8 ? 4 * (rg >> 1) + (rg & 1) : RBV == 4 ? ((EPI == MB_SWIGLU && !a.gu_interleaved) ? rg : 2 * rg) : 0;
    const int rb1 = (RBV == 8 && EPI == MB_QKV) ? 2 * rg + 1 : RBV == 8 ? rb0 + 2 : RBV == 4 ? ((EPI == MB_SWIGLU && !a.gu_interleaved) ? rg + 2 : 2 * rg + 1) : 1;
    const int ksplit = gridDim.y, ks = blockIdx.y;
    int t0, t1;
    mb_k_range(a.K16, ks, ksplit, t0, t1);
    const int nst = (t1 - t0 + KS - 1) / KS;
    const int zb0 = blockIdx.z * GEO::BLOCKS;       // first 64-row block of this workgroup (grid.z splits the token blocks)

    const bf16x8* __restrict__ wbase = (const bf16x8*)a.wp;
    const bf16x8* __restrict__ xbase = (const bf16x8*)a.xp;
    const bool hi = wave < N_HI;                    // this wave carries NP_HI pieces
    // piece p = wave + 8 i of a stage is a weight piece iff p < A_STAGE (i = 0 for the waves below A_STAGE; i = 0 and 1 at RBV = 8)
    const bf16x8* src[NP_HI];                       // SCH 0: per-lane source of the piece at k-tile 0
    unsigned gstr[NP_HI];                           // its k-tile stride (16 B units); SCH 1: in bytes, wave-uniform
    unsigned voff[NP_HI];                           // SCH 1: per-lane byte offset of the piece inside its buffer (k-tile 0)
    int pkk[NP_HI], pdst[NP_HI];                    // wave-uniform: k-tile inside the stage, byte offset inside the stage buffer
    // SCH 1 buffers: the weights of this workgroup (planned: its chunk range; classic: its RBV row-blocks) and the x image
    const bf16x8* wg_w = a.planned ? wbase + (size_t)blockIdx.x * (unsigned)a.wg_chunks : wbase + (size_t)blockIdx.x * RBV * a.K16 * 64;
    const __amdgpu_buffer_rsrc_t rs_w = dma_rsrc(wg_w), rs_x = dma_rsrc(xbase);
#pragma unroll
    for (int i = 0; i < NP_HI; ++i) {
        const int p = wave + 8 * ((i < NP_LO || hi) ? i : 0);
        if (p < A_STAGE) {
            const int kk = p / RBV, rb = p % RBV;
            if (a.planned) {
                const int nvb = a.nvl[rb];
                const int rr = (lane & 31) < nvb ? (lane & 31) : nvb - 1;
                gstr[i] = BUF ? (unsigned)(32 * nvb) : (unsigned)(2 * nvb);
                voff[i] = ((unsigned)a.boff[rb] + (unsigned)((lane >> 5) * nvb + rr)) * 16u;
                src[i] = wbase + ((size_t)blockIdx.x * (unsigned)a.wg_chunks + (unsigned)a.boff[rb] + (unsigned)((lane >> 5) * nvb + rr));
            } else {
                gstr[i] = BUF ? 1024u : 64u;
                voff[i] = ((unsigned)(rb * a.K16 * 64) + (unsigned)lane) * 16u;
                src[i] = wbase + ((size_t)(blockIdx.x * RBV + rb) * a.K16 * 64 + lane);
            }
            pkk[i] = kk;
            pdst[i] = (kk * RBV + rb) * 1024;
        } else {
            const int q = p - A_STAGE, kk = q / NTBP, tb = q % NTBP;
            int xb = zb0 + (tb >> 1);
            xb = xb < a.nblk ? xb : a.nblk - 1;     // token blocks past the step re-read the last real block (results dropped)
            gstr[i] = BUF ? 2048u : 128u;
            voff[i] = ((unsigned)(((xb * a.K16) * 2 + (tb & 1)) * 64) + (unsigned)lane) * 16u;
            src[i] = xbase + ((size_t)((xb * a.K16) * 2 + (tb & 1)) * 64 + lane);
            pkk[i] = kk;
            pdst[i] = (A_STAGE + kk * NTBP + tb) * 1024;
        }
    }
    auto issue_one = [&](int sidx, int i) {         // piece i of this wave's share of stage sidx (k-tile clamped into the range)
        int kt = t0 + sidx * KS + pkk[i];
        kt = kt < t1 ? kt : t1 - 1;
        char* d = lds_raw + (sidx % NR) * (STAGE * 1024) + pdst[i];
        if (i < NP_LO || hi) {
            if constexpr (BUF) {
                const unsigned so = (unsigned)kt * gstr[i];
                if (8 * i < A_STAGE && wave + 8 * i < A_STAGE) {        // (the first test is a compile-time one: later pieces are x pieces)
                    if (!a.w_keep) dma_piece<2>(rs_w, d, voff[i], so);   // streamed weights: nt
                    else dma_piece<0>(rs_w, d, voff[i], so);
                } else dma_piece<0>(rs_x, d, voff[i], so);
            } else {
                const bf16x8* g = src[i] + (size_t)kt * gstr[i];
                if (wave + 8 * i < A_STAGE && !a.w_keep) __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)d, 16, 0, 2);   // streamed weights: nt
                else __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)d, 16, 0, 0);
            }
        }
    };
    auto issue = [&](int sidx) {
#pragma unroll
        for (int i = 0; i < NP_HI; ++i) issue_one(sidx, i);
    };

    f32x16 acc[2][TW];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][t][i] = 0.f;

    // Fragment reads are inline asm so that the LDS queue is counted by hand (lgkmcnt retires ds_reads in order): hipcc's own
    // scoreboard gives up across the loop back-edge and waits lgkmcnt(0) before the first MFMA of a set, i.e. for the reads it
    // has just issued for the OTHER set.
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds_raw + (unsigned)lane * 16u;
    // fragment j of k-tile (sidx, kk) in the order the MFMAs want them: weights of row-block 0, the TW x tiles, weights of row-block 1
    auto read_one = [&](int sidx, int kk, int j, bf16x8 (&fa)[2], bf16x8 (&fb)[TW]) {
        const unsigned S = lds0 + (unsigned)((sidx % NR) * (STAGE * 1024));
        const unsigned A = S + (unsigned)(kk * RBV * 1024);
        const unsigned B = S + (unsigned)((A_STAGE + kk * NTBP + tq * TW) * 1024);
        if (j == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[0]) : "v"(A + rb0 * 1024u) : "memory");
        else if (j == TW + 1) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[1]) : "v"(A + rb1 * 1024u) : "memory");
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[j - 1]) : "v"(B), "n"((j - 1) * 1024) : "memory");
    };
    auto read_frags = [&](int sidx, int kk, bf16x8 (&fa)[2], bf16x8 (&fb)[TW]) {
        const unsigned S = lds0 + (unsigned)((sidx % NR) * (STAGE * 1024));
        const unsigned A = S + (unsigned)(kk * RBV * 1024);
        const unsigned B = S + (unsigned)((A_STAGE + kk * NTBP + tq * TW) * 1024);
        asm volatile("ds_read_b128 %0, %1" : "=v"(fa[0]) : "v"(A + rb0 * 1024u) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(fa[1]) : "v"(A + rb1 * 1024u) : "memory");
#pragma unroll
        for (int t = 0; t < TW; ++t) asm volatile("ds_read_b128 %0, %1" : "=v"(fb[t]) : "v"(B + t * 1024u) : "memory");
    };
    auto mma = [&](int r, int t, const bf16x8 (&fa)[2], const bf16x8 (&fb)[TW]) {
        if constexpr (DBG == 1 || DBG == 3) acc[r][t][0] += __builtin_bit_cast(float, (int)(fa[r][0] ^ fb[t][0]));
        else acc[r][t] = LA_MFMA(fa[r], fb[t], acc[r][t], 0, 0, 0);
    };

    constexpr int H = GEO::H;
#pragma unroll
    for (int i = 0; i < NR - 1; ++i) issue(i);
    if (hi) vm_wait<(NR - 2) * NP_HI>(); else vm_wait<(NR - 2) * NP_LO>();
    __builtin_amdgcn_s_barrier();
    bf16x8 fa0[2], fb0[TW], fa1[2], fb1[TW];
    read_frags(0, 0, fa0, fb0);
    constexpr int NMMA = 2 * TW, NRD = 2 + TW;
    if constexpr (ILV) {
        constexpr int NG1 = NMMA > H ? (NMMA > NRD ? NMMA : NRD) : (H > NRD ? H : NRD);
        constexpr int H2 = NP_HI - H;
        constexpr int NG2 = NMMA > H2 ? (NMMA > NRD ? NMMA : NRD) : (H2 > NRD ? H2 : NRD);
        // DBG = 6: shader cycles per wave summed over the stages: [0] first half incl. the wait for its reads, [1] wait for the own DMA
        // pieces, [2] barrier, [3] second half incl. the wait for its reads (each s_memtime costs its own round trip: read proportions)
        // (Measured with this build, profiles/r04_wide_gemm_schedule.txt part 4: issue arbitration between the two waves of a SIMD — w and
        // w + 4 — is oldest-first: waves 0..3 run a stage in ~940 cycles and wait ~470 at the barrier for waves 4..7.  s_setprio schedules
        // that make the two classes take turns remove the barrier wait and lengthen both halves by as much: the SIMD's combined issue
        // stream, not the sharing, sets the stage time.  Not kept.)
        unsigned long long tacc[4] = {0ull, 0ull, 0ull, 0ull}, tprev = 0ull;
        if constexpr (DBG == 6) tprev = __builtin_readcyclecounter();
        for (int s = 0; s < nst; ++s) {
            const bool full = (t0 + s * KS + 1) < t1;   // an odd K range ends on half a stage (workgroup-uniform)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // set0 (read during the previous half) is complete
            if constexpr (DBG == 6) { const unsigned long long t = __builtin_readcyclecounter(); if (s > 0) tacc[3] += t - tprev; tprev = t; }
            __builtin_amdgcn_sched_barrier(0);
            // first half: MFMA of set0 | one fragment read of set1 (stage s, k-tile 1) | one DMA piece of stage s+3, group after group
#pragma unroll
            for (int m = 0; m < NG1; ++m) {
                if (m < NMMA) mma(m / TW, m % TW, fa0, fb0);
                if (m < NRD) read_one(s, 1, m, fa1, fb1);
                if (m < H) issue_one(s + NR - 1, m);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // every read of stage s is complete
            if constexpr (DBG == 6) { const unsigned long long t = __builtin_readcyclecounter(); tacc[0] += t - tprev; tprev = t; __builtin_amdgcn_sched_barrier(0); }
            if (hi) vm_wait<(NR - 3) * NP_HI + H>(); else vm_wait<(NR - 3) * NP_LO + H>();   // own pieces of stage s+1 landed
            if constexpr (DBG == 6) { const unsigned long long t = __builtin_readcyclecounter(); tacc[1] += t - tprev; tprev = t; __builtin_amdgcn_sched_barrier(0); }
            __builtin_amdgcn_s_barrier();                                // stage s+1 complete for everyone; the slot of stage s is free
            if constexpr (DBG == 6) { const unsigned long long t = __builtin_readcyclecounter(); tacc[2] += t - tprev; tprev = t; }
            __builtin_amdgcn_sched_barrier(0);
            // second half: MFMA of set1 | one fragment read of set0 (stage s+1, k-tile 0) | the remaining pieces of stage s+3
            if (full) {
#pragma unroll
                for (int m = 0; m < NG2; ++m) {
                        if (m < NMMA) mma(m / TW, m % TW, fa1, fb1);
                    if (m < NRD) read_one(s + 1, 0, m, fa0, fb0);
                    if (m < H2) issue_one(s + NR - 1, H + m);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int m = 0; m < (NRD > H2 ? NRD : H2); ++m) {
                    if (m < NRD) read_one(s + 1, 0, m, fa0, fb0);
                    if (m < H2) issue_one(s + NR - 1, H + m);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (DBG == 6) {
            if (a.dbg_times && lane == 0) {
                long long* o = a.dbg_times + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + wave) * 8;
                o[0] = (long long)tacc[0]; o[1] = (long long)tacc[1]; o[2] = (long long)tacc[2]; o[3] = (long long)tacc[3]; o[4] = nst;
            }
        }
    } else {
    for (int s = 0; s < nst; ++s) {
        const bool full = (t0 + s * KS + 1) < t1;   // an odd K range ends on half a stage (workgroup-uniform)
        if constexpr (DBG != 3) read_frags(s, 1, fa1, fb1);
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(DBG != 3 ? NRD : 0) : "memory");      // set0 complete, set1 in flight
        __builtin_amdgcn_sched_barrier(0);
        // first half: MFMAs of set0, the first H DMA pieces of stage s+3 between them (its slot was freed one barrier ago)
#pragma unroll
        for (int m = 0; m < (NMMA > H ? NMMA : H); ++m) {
            if (m < NMMA) mma(m / TW, m % TW, fa0, fb0);
            if constexpr (DBG != 2 && DBG != 5) { if (m < H) issue_one(s + NR - 1, m); }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // every read of stage s is complete
        // own pieces of stage s+1 landed: younger are all of stage s+2 and the H pieces of s+3
        if constexpr (DBG != 2 && DBG != 5) { if (hi) vm_wait<(NR - 3) * NP_HI + H>(); else vm_wait<(NR - 3) * NP_LO + H>(); }
        else vm_wait<0>();
        if constexpr (DBG != 5) __builtin_amdgcn_s_barrier();               // stage s+1 complete for everyone; the slot of stage s is free
        if constexpr (DBG != 3) read_frags(s + 1, 0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        // second half: MFMAs of set1 and the remaining pieces of stage s+3
        if (full) {
#pragma unroll
            for (int m = 0; m < (NMMA > NP_HI - H ? NMMA : NP_HI - H); ++m) {
                if (m < NMMA) mma(m / TW, m % TW, fa1, fb1);
                if constexpr (DBG != 2 && DBG != 5) { if (m < NP_HI - H) issue_one(s + NR - 1, H + m); }
            }
        } else {
            if constexpr (DBG != 2 && DBG != 5) {
#pragma unroll
                for (int m = H; m < NP_HI; ++m) issue_one(s + NR - 1, m);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    vm_wait<0>();
    if constexpr (DBG == 4) {                       // measurement: the launch without its epilogue
        float sacc = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int t = 0; t < TW; ++t) sacc += acc[r][t][0] + acc[r][t][15];
        if (sacc == 1.2345e-30f) a.act_xp[0] = 0;
        return;
    }

    // ---- epilogues from the accumulators: lane (tl, hh) of tile (row-block, token block) holds rows f = 8 (i/4) + 4 hh + i%4
    //      of token tl; same arithmetic and rounding points as mb_epilogue_block / the single-block kernels
    const int tl = lane & 31, hh = lane >> 5;
    // MB_SWIGLU staging geometry: features [sw_lo, sw_lo + sw_r) of this workgroup, columns shifted by sw_sh = sw_lo % 8 so that
    // the 16-byte chunks of the activation image are 16-byte aligned in the tile; row stride in elements, never a multiple of 64
    constexpr int REG = RBV == 8 ? 2 : 1;           // planned regions per workgroup (their features are adjacent)
    const int sw_lo = a.gu_interleaved ? 64 * blockIdx.x : a.R * REG * blockIdx.x, sw_r = a.gu_interleaved ? 64 : REG * a.R, sw_sh = sw_lo & 7;
    const int sw_nch = (sw_sh + sw_r + 7) >> 3;
    const int sw_stride = (sw_nch * 8) % 64 == 0 ? sw_nch * 8 + 8 : sw_nch * 8;
    if constexpr (EPI == MB_SWIGLU) __syncthreads();            // every wave is done with the ring
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        const int tbg = tq * TW + t, blk = zb0 + (tbg >> 1), tok = (tbg & 1) * 32 + tl;
        if (blk >= a.nblk) continue;                // wave-uniform
        if constexpr (EPI == MB_SLAB) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    const f32x4 v = {acc[r][t][4 * gi], acc[r][t][4 * gi + 1], acc[r][t][4 * gi + 2], acc[r][t][4 * gi + 3]};
                    float* o = a.slabs + ((size_t)ks * a.M + blk * 64 + tok) * a.N + (blockIdx.x * RBV + (r == 0 ? rb0 : rb1)) * 32 + 8 * gi + 4 * hh;
                    *(f32x4*)o = v;
                }
        } else if constexpr (EPI == MB_SWIGLU) {
            // act = bf16(silu(bf16(g)) * bf16(u)) (LlamaMLP.forward, :185-186), parked as tile[token][sh + feature - lo] in the
            // (drained) ring: the activation image keeps 8 consecutive features of a token in one 16-byte chunk, but a balanced
            // plan starts a workgroup's features anywhere (R = 43 at the 7B shape) — 2-byte stores straight from the MFMA layout
            // cost ~27 us per launch at 512 rows (L2 write REQUESTS, not bytes); see the store pass below
            const int nvg = a.nv[rb0];
            bf16_t* tile = (bf16_t*)lds_raw;
            const int c0 = sw_sh + (RBV == 8 ? (rg >> 1) * a.R + 32 * (rg & 1) : 32 * rg);
            const int trow = (tbg >> 1) * 64 + tok;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (8 * (i >> 2) >= nvg) break;                        // wave-uniform: no valid row in this register group
                const int f = 8 * (i >> 2) + 4 * hh + (i & 3);
                const float gv = bfr_hw(acc[0][t][i]), uv = bfr_hw(acc[1][t][i]);
                // silu in fast fp32 (v_exp_f32 + v_rcp_f32, ~1 ulp each): the result is rounded to bf16 right away, and the
                // libm / IEEE-division forms cost ~60 VALU instructions per element — 64 elements per lane at 512 rows
                const float sv = bfr_hw(gv * __builtin_amdgcn_rcpf(1.0f + __expf(-gv)));
                const bf16_t o = f2bf_hw(sv * uv);
                if (f < nvg) tile[trow * sw_stride + c0 + f] = o;
            }
        } else if constexpr (EPI == MB_QKV) {
            // RoPE in bf16 arithmetic (apply_rotary_pos_emb, modeling_llama.py:154-169); tile 0 = lo halves, tile 1 = hi halves
            const int ps = a.pos[blk * 64 + tok];
            const int region = blockIdx.x * (RBV / 2) + rg;      // RBV = 4 (paired form): row group rg holds the {lo, hi} blocks of region 2 x + rg
            if ((a.R & 3) == 0) {
                // groups of FOUR consecutive rows = one register group of the lane (R % 4 == 0 keeps a group inside one head slot and
                // one 16-byte chunk of the fragment: 7B / Mistral / Mixtral shapes): 8-byte cos / sin loads and fragment stores —
                // half the store instructions of the pair form below (the epilogue of this launch is store-issue-bound)
#pragma unroll
                for (int i = 0; i < 16; i += 4) {
                    if (8 * (i >> 2) >= a.nv[0]) break;                // wave-uniform
                    const int f = 8 * (i >> 2) + 4 * hh;
                    if (f < a.nv[0]) {
                        const int prr = a.R * region + f, slot = prr >> 6, dlo = prr & 63, dhi = dlo + 64;
                        if (slot < a.nh + a.nkv) {
                            bf16_t* dst = slot < a.nh ? a.qf + ((size_t)blk * a.nh + slot) * 8192
                                                      : a.kfresh + ((size_t)blk * a.nkv + (slot - a.nh)) * 8192;
                            const uint2 c4 = *(const uint2*)(a.rcos + (size_t)ps * 64 + dlo), s4 = *(const uint2*)(a.rsin + (size_t)ps * 64 + dlo);
                            uint2 olo = {0u, 0u}, ohi = {0u, 0u};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const uint32_t cw = e < 2 ? c4.x : c4.y, sw = e < 2 ? s4.x : s4.y;
                                const float cc = bf2f((bf16_t)(cw >> (16 * (e & 1)))), sn = bf2f((bf16_t)(sw >> (16 * (e & 1))));
                                const float bl = bfr_hw(acc[0][t][i + e]), bh = bfr_hw(acc[1][t][i + e]);
                                const uint32_t lo16 = (uint32_t)f2bf_hw(bfr_hw(bl * cc) + bfr_hw(-bh * sn)) << (16 * (e & 1));
                                const uint32_t hi16 = (uint32_t)f2bf_hw(bfr_hw(bh * cc) + bfr_hw(bl * sn)) << (16 * (e & 1));
                                if (e < 2) { olo.x |= lo16; ohi.x |= hi16; } else { olo.y |= lo16; ohi.y |= hi16; }
                            }
                            *(uint2*)(dst + rf_offset(tok, dlo)) = olo;
                            *(uint2*)(dst + rf_offset(tok, dhi)) = ohi;
                        } else {
                            bf16_t* dst = a.vfresh + ((size_t)blk * a.nkv + (slot - a.nh - a.nkv)) * 8192;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                dst[vf_offset(tok, dlo + e)] = f2bf_hw(acc[0][t][i + e]);
                                dst[vf_offset(tok, dhi + e)] = f2bf_hw(acc[1][t][i + e]);
                            }
                        }
                    }
                }
            } else if ((a.R & 1) == 0) {
                // pairs of consecutive rows (an even R keeps them in one head and one 16-byte chunk): 4-byte table loads and stores
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    if (8 * (i >> 2) >= a.nv[0]) break;                // wave-uniform
                    const int f = 8 * (i >> 2) + 4 * hh + (i & 3);
                    if (f < a.nv[0]) {
                        const int prr = a.R * region + f, slot = prr >> 6, dlo = prr & 63, dhi = dlo + 64;
                        if (slot < a.nh + a.nkv) {
                            bf16_t* dst = slot < a.nh ? a.qf + ((size_t)blk * a.nh + slot) * 8192
                                                      : a.kfresh + ((size_t)blk * a.nkv + (slot - a.nh)) * 8192;
                            const uint32_t c2 = *(const uint32_t*)(a.rcos + (size_t)ps * 64 + dlo), s2 = *(const uint32_t*)(a.rsin + (size_t)ps * 64 + dlo);
                            uint32_t olo = 0, ohi = 0;
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const float cc = bf2f((bf16_t)(c2 >> (16 * e))), sn = bf2f((bf16_t)(s2 >> (16 * e)));
                                const float bl = bfr_hw(acc[0][t][i + e]), bh = bfr_hw(acc[1][t][i + e]);
                                olo |= (uint32_t)f2bf_hw(bfr_hw(bl * cc) + bfr_hw(-bh * sn)) << (16 * e);
                                ohi |= (uint32_t)f2bf_hw(bfr_hw(bh * cc) + bfr_hw(bl * sn)) << (16 * e);
                            }
                            *(uint32_t*)(dst + rf_offset(tok, dlo)) = olo;
                            *(uint32_t*)(dst + rf_offset(tok, dhi)) = ohi;
                        } else {
                            bf16_t* dst = a.vfresh + ((size_t)blk * a.nkv + (slot - a.nh - a.nkv)) * 8192;
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                dst[vf_offset(tok, dlo + e)] = f2bf_hw(acc[0][t][i + e]);
                                dst[vf_offset(tok, dhi + e)] = f2bf_hw(acc[1][t][i + e]);
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int f = 8 * (i >> 2) + 4 * hh + (i & 3);
                    if (f < a.nv[0]) {
                        const int prr = a.R * region + f, slot = prr >> 6, dlo = prr & 63, dhi = dlo + 64;
                        const float xl = acc[0][t][i], xh = acc[1][t][i];
                        if (slot < a.nh + a.nkv) {
                            bf16_t* dst = slot < a.nh ? a.qf + ((size_t)blk * a.nh + slot) * 8192
                                                      : a.kfresh + ((size_t)blk * a.nkv + (slot - a.nh)) * 8192;
                            const float cc = bf2f(a.rcos[(size_t)ps * 64 + dlo]), sn = bf2f(a.rsin[(size_t)ps * 64 + dlo]);
                            const float bl = bfr_hw(xl), bh = bfr_hw(xh);
                            dst[rf_offset(tok, dlo)] = f2bf_hw(bfr_hw(bl * cc) + bfr_hw(-bh * sn));
                            dst[rf_offset(tok, dhi)] = f2bf_hw(bfr_hw(bh * cc) + bfr_hw(bl * sn));
                        } else {
                            bf16_t* dst = a.vfresh + ((size_t)blk * a.nkv + (slot - a.nh - a.nkv)) * 8192;
                            dst[vf_offset(tok, dlo)] = f2bf_hw(xl);
                            dst[vf_offset(tok, dhi)] = f2bf_hw(xh);
                        }
                    }
                }
            }
        } else {
            // lm_head: bf16 logits + one argmax candidate per (row-block, token)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int rq = r == 0 ? rb0 : rb1;
                float best = -INFINITY;
                int bidx = 0x7fffffff;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int f = 8 * (i >> 2) + 4 * hh + (i & 3);
                    if (f < a.nv[rq]) {
                        const bf16_t hv = f2bf(acc[r][t][i]);
                        const int idx = a.R * blockIdx.x + 32 * rq + f;
                        if (a.logits) a.logits[((size_t)blk * 64 + tok) * a.N + idx] = hv;
                        const float v = bf2f(hv);
                        if (v > best || (v == best && idx < bidx)) { best = v; bidx = idx; }
                    }
                }
                float ob = __shfl_xor(best, 32, 64);
                int oi = __shfl_xor(bidx, 32, 64);
                if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
                if (hh == 0) {
                    const size_t slot = ((size_t)blk * gridDim.x + blockIdx.x) * 4 + rq;
                    a.cand_val[slot * 64 + tok] = best;
                    a.cand_idx[slot * 64 + tok] = bidx;
                }
            }
        }
    }
    if constexpr (EPI == MB_SWIGLU) {
        // store pass: one (token, 16-byte chunk) item per lane and step — 32 consecutive tokens of a lane group are 512
        // contiguous bytes of the activation image; the first / last chunk of a token may be shared with the neighbour workgroups
        __syncthreads();
        const bf16_t* tile = (const bf16_t*)lds_raw;
        const int ntok = GEO::BLOCKS * 64;
        for (int it = threadIdx.x; it < ntok * sw_nch; it += 512) {
            const int trow = it % ntok, c = it / ntok, blk = zb0 + (trow >> 6);
            if (blk >= a.nblk) continue;
            const int fa_ = (sw_lo & ~7) + 8 * c;                        // first feature of the chunk
            bf16_t* dst = a.act_xp + (size_t)blk * 64 * a.N + xp_offset(trow & 63, fa_);
            const bf16_t* srcp = tile + trow * sw_stride + 8 * c;
            if (fa_ >= sw_lo && fa_ + 8 <= sw_lo + sw_r) {
                *(bf16x8*)dst = *(const bf16x8*)srcp;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (fa_ + e >= sw_lo && fa_ + e < sw_lo + sw_r) dst[e] = srcp[e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_gemm_fat (round 5): the PAIRED wide launches with FOUR waves of a 4 x TW tile each instead of eight waves of 2 x TW.
//   gate/up (RBV = 8): two adjacent planned regions {G0,G1,U0,U1} x 2, wave (rg, tq) = region rg x token group tq (2 x 2), TW = 2..4;
//   slab / QKV (RBV = 4): the two paired regions' {lo, hi} or plain row-blocks 0..3, wave tq = token group (1 x 4), TW = 1..2.
//   A wave owns ALL four row-blocks of its row group x TW token blocks: 4 TW accumulator tiles (256 registers at TW = 4: one wave
//   per SIMD on the 512-entry unified file, accumulators in AGPRs).
// Why: the stage time of k_gemm_wide is the SIMD's in-order instruction stream, not its data (profiles/r04_wide_gemm_schedule.txt:
// 32 MFMAs + 10 LDS-DMA issues + 24 fragment reads + ~60 SALU per SIMD and stage, matrix pipe busy 62 %).  Per MFMA the fat wave
// issues (4 + TW) / (4 TW) fragment reads (0.5 at TW = 4, was 0.75) and the same DMA pieces with half the address bookkeeping
// (one wave's worth instead of two), and its 4 TW MFMAs per k-tile leave every read a dozen gaps to land in.  What it gives up: the
// second wave of a SIMD that covers a wait (MI355X_MICROARCH.md prices that at <= 5 hidden issues per MFMA gap for one wave per
// SIMD — this stream has ~1.5).  Same ring, same stage anatomy and the same MFMA chain per output element as k_gemm_wide (k-tiles
// in ascending order into one accumulator): bit-identical outputs.  Measured per launch (profiles/r05_fat_waves.txt): gate/up
// 7B 512 rows 113.8 -> 102.9 us, 256 rows 72.0 -> 65.8; Mistral 119.4 -> 113.3 / 77.5 -> 71.3; 13B 145.9 -> 138.7 / 96.1 -> 84.6.
// The K range of a workgroup must hold an even number of k-tiles (a branch around half a stage's MFMAs makes every accumulator
// tile a loop-carried phi with two sources, and hipcc then spills accumulators INSIDE the loop): the launcher sends other shapes
// to k_gemm_wide.
// ---------------------------------------------------------------------------------------------------------------
template <int RBV, int TW, int RPW_ = 4> struct FatGeom {
    static constexpr int KS = 2, RPW = RPW_, NW = 4, RG = RBV / RPW, TQ = NW / RG, NTBP = TQ * TW;
    static constexpr int A_STAGE = KS * RBV, B_STAGE = KS * NTBP, STAGE = A_STAGE + B_STAGE;
    static constexpr int NR = 4 * STAGE * 1024 <= 144 * 1024 ? 4 : 3;          // ring slots (one region x 512 rows: 40 KiB per stage -> 3)
    static constexpr int NP = STAGE / NW;                                    // pieces per wave and stage
    static constexpr int NPA = A_STAGE / NW;                                 // ... of which weight pieces (the first NPA)
    static constexpr int H = (NP + 1) / 2;
    static constexpr int LDS = NR * STAGE * 1024;
    static constexpr int BLOCKS = NTBP / 2;
    static_assert((RBV == 8 || RBV == 4 || RBV == 2) && RBV % RPW == 0 && STAGE % NW == 0 && A_STAGE % NW == 0 && NTBP % 2 == 0 && LDS <= 160 * 1024, "ring geometry");
};

// RoPE + fragment stores of one {lo, hi} accumulator pair (the QKV epilogue of k_gemm_wide, same arithmetic and rounding points:
// apply_rotary_pos_emb, modeling_llama.py:154-169, in bf16 arithmetic; V rows pass through)
// RV = rows a lane handles per store: 4 (R % 4 == 0: 8-byte table loads and fragment stores), 2 (R even: 4-byte), 1 (any R)
template <int RV>
__device__ __forceinline__ void fat_qkv_tile(const MbArgs& a, const f32x16& lo, const f32x16& hi, int region, int blk, int tok, int hh) {
    const int ps = a.pos[blk * 64 + tok];
    if constexpr (RV == 4) {
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            if (8 * (i >> 2) >= a.nv[0]) break;
            const int f = 8 * (i >> 2) + 4 * hh;
            if (f < a.nv[0]) {
                const int prr = a.R * region + f, slot = prr >> 6, dlo = prr & 63, dhi = dlo + 64;
                if (slot < a.nh + a.nkv) {
                    bf16_t* dst = slot < a.nh ? a.qf + ((size_t)blk * a.nh + slot) * 8192
                                              : a.kfresh + ((size_t)blk * a.nkv + (slot - a.nh)) * 8192;
                    const uint2 c4 = *(const uint2*)(a.rcos + (size_t)ps * 64 + dlo), s4 = *(const uint2*)(a.rsin + (size_t)ps * 64 + dlo);
                    uint2 olo = {0u, 0u}, ohi = {0u, 0u};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t cw = e < 2 ? c4.x : c4.y, sw = e < 2 ? s4.x : s4.y;
                        const float cc = bf2f((bf16_t)(cw >> (16 * (e & 1)))), sn = bf2f((bf16_t)(sw >> (16 * (e & 1))));
                        const float bl = bfr_hw(lo[i + e]), bh = bfr_hw(hi[i + e]);
                        const uint32_t lo16 = (uint32_t)f2bf_hw(bfr_hw(bl * cc) + bfr_hw(-bh * sn)) << (16 * (e & 1));
                        const uint32_t hi16 = (uint32_t)f2bf_hw(bfr_hw(bh * cc) + bfr_hw(bl * sn)) << (16 * (e & 1));
                        if (e < 2) { olo.x |= lo16; ohi.x |= hi16; } else { olo.y |= lo16; ohi.y |= hi16; }
                    }
                    *(uint2*)(dst + rf_offset(tok, dlo)) = olo;
                    *(uint2*)(dst + rf_offset(tok, dhi)) = ohi;
                } else {
                    bf16_t* dst = a.vfresh + ((size_t)blk * a.nkv + (slot - a.nh - a.nkv)) * 8192;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dst[vf_offset(tok, dlo + e)] = f2bf_hw(lo[i + e]);
                        dst[vf_offset(tok, dhi + e)] = f2bf_hw(hi[i + e]);
                    }
                }
            }
        }
    } else if constexpr (RV == 2) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            if (8 * (i >> 2) >= a.nv[0]) break;
            const int f = 8 * (i >> 2) + 4 * hh + (i & 3);
            if (f < a.nv[0]) {
                const int prr = a.R * region + f, slot = prr >> 6, dlo = prr & 63, dhi = dlo + 64;
                if (slot < a.nh + a.nkv) {
                    bf16_t* dst = slot < a.nh ? a.qf + ((size_t)blk * a.nh + slot) * 8192
                                              : a.kfresh + ((size_t)blk * a.nkv + (slot - a.nh)) * 8192;
                    const uint32_t c2 = *(const uint32_t*)(a.rcos + (size_t)ps * 64 + dlo), s2 = *(const uint32_t*)(a.rsin + (size_t)ps * 64 + dlo);
                    uint32_t olo = 0, ohi = 0;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float cc = bf2f((bf16_t)(c2 >> (16 * e))), sn = bf2f((bf16_t)(s2 >> (16 * e)));
                        const float bl = bfr_hw(lo[i + e]), bh = bfr_hw(hi[i + e]);
                        olo |= (uint32_t)f2bf_hw(bfr_hw(bl * cc) + bfr_hw(-bh * sn)) << (16 * e);
                        ohi |= (uint32_t)f2bf_hw(bfr_hw(bh * cc) + bfr_hw(bl * sn)) << (16 * e);
                    }
                    *(uint32_t*)(dst + rf_offset(tok, dlo)) = olo;
                    *(uint32_t*)(dst + rf_offset(tok, dhi)) = ohi;
                } else {
                    bf16_t* dst = a.vfresh + ((size_t)blk * a.nkv + (slot - a.nh - a.nkv)) * 8192;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        dst[vf_offset(tok, dlo + e)] = f2bf_hw(lo[i + e]);
                        dst[vf_offset(tok, dhi + e)] = f2bf_hw(hi[i + e]);
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = 8 * (i >> 2) + 4 * hh + (i & 3);
            if (f < a.nv[0]) {
                const int prr = a.R * region + f, slot = prr >> 6, dlo = prr & 63, dhi = dlo + 64;
                const float xl = lo[i], xh = hi[i];
                if (slot < a.nh + a.nkv) {
                    bf16_t* dst = slot < a.nh ? a.qf + ((size_t)blk * a.nh + slot) * 8192
                                              : a.kfresh + ((size_t)blk * a.nkv + (slot - a.nh)) * 8192;
                    const float cc = bf2f(a.rcos[(size_t)ps * 64 + dlo]), sn = bf2f(a.rsin[(size_t)ps * 64 + dlo]);
                    const float bl = bfr_hw(xl), bh = bfr_hw(xh);
                    dst[rf_offset(tok, dlo)] = f2bf_hw(bfr_hw(bl * cc) + bfr_hw(-bh * sn));
                    dst[rf_offset(tok, dhi)] = f2bf_hw(bfr_hw(bh * cc) + bfr_hw(bl * sn));
                } else {
                    bf16_t* dst = a.vfresh + ((size_t)blk * a.nkv + (slot - a.nh - a.nkv)) * 8192;
                    dst[vf_offset(tok, dlo)] = f2bf_hw(xl);
                    dst[vf_offset(tok, dhi)] = f2bf_hw(xh);
                }
            }
        }
    }
}

// STG = 1 (round 5, second form): the operands reach LDS through REGISTERS (buffer_load_dwordx4 -> VGPR -> ds_write_b128) instead of LDS-DMA.
//   What the DMA form pays: an LDS-DMA piece blocks its wave's issue for ~60-185 cycles (MI355X_MICROARCH.md price list), and with one wave per
//   SIMD nothing else issues meanwhile — 8 pieces per wave and stage idle the matrix pipe for roughly half a stage (the 512-row gate/up launch:
//   128 stages x ~1560 cycles at 2.03 GHz = the measured 103-113 us, where 1024 cycles per stage would be the MFMA time).  A plain buffer load
//   issues in a few cycles and the fat kernels have the registers to hold two stages in flight (gate/up: 135 + 256 of 512).  Pipeline: the
//   pieces of stage s + 3 are requested in the second half of stage s (one per MFMA gap) into the register set of stage s + 1's parity — which
//   the first half of stage s has just emptied into LDS slot (s + 1) % 2 (one ds_write per gap) — and the barrier in the middle of stage s
//   publishes stage s + 1 as before.  Two LDS slots instead of four.  Same MFMAs on the same operands in the same order: bit-identical.
template <int RBV, int TW, int EPI, int RV = 4, int WPOL = 0, int RPW_ = 4, int STG = 0>
__global__ __launch_bounds__(256) void k_gemm_fat(MbArgs a) {
    using GEO = FatGeom<RBV, TW, RPW_>;
    constexpr int KS = GEO::KS, RPW = GEO::RPW, NW = GEO::NW, TQ = GEO::TQ, NTBP = GEO::NTBP;
    constexpr int A_STAGE = GEO::A_STAGE, STAGE = GEO::STAGE, NR = STG == 1 ? 2 : GEO::NR, NP = GEO::NP, NPA = GEO::NPA, H = GEO::H;
    static_assert((EPI == MB_SWIGLU && (RBV == 8 || RBV == 4) && RPW_ == 4) || ((EPI == MB_SLAB || EPI == MB_QKV) && RBV == 4 && RPW_ == 4) || ((EPI == MB_QKV || EPI == MB_SLAB) && RBV == 2 && RPW_ == 2), "gate/up: one planned region {G0,G1,U0,U1} x all token blocks, or two regions x half of them; slab / QKV: two {lo, hi} regions");
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rg = wave / TQ, tq = wave % TQ;          // row group (4 row-blocks), token group
    // workgroup -> (weight regions bx, K split ks, token group bz).  Slab launches may re-map (MbArgs.xcd_map): workgroups go to the XCDs round-robin
    // by their linear id, and with the plain grid order every XCD sees every K split — all of x (512 rows x K) lands in each of the 8 L2s: for
    // down at the Mistral shape that is 8 x 14.7 MB = as many fabric bytes as the weights (counters: 268 MB per launch for 117 MB of W).  Re-mapped,
    // XCD c works on K split c / (8 / ksplit) only: x per L2 = 1 / ksplit of it, the weights as before (each region's K slice is read in one XCD).
    int bx = blockIdx.x, ks = blockIdx.y, bz = blockIdx.z;
    const int ksplit = gridDim.y;
    if constexpr (EPI == MB_SLAB) {
        if (a.xcd_map) {
            const int gx = gridDim.x, gz = gridDim.z;
            const int lin = bx + gx * (ks + ksplit * bz);
            const int per = (gx * ksplit * gz) >> 3;           // workgroups per XCD (launcher: a multiple of 8 in total)
            const int xcd = lin & 7, slot = lin >> 3, g = 8 / ksplit;
            const int j = (xcd % g) * per + slot;              // 0 .. gx * gz: the tiles of this K split, both token groups of a region in one XCD
            ks = xcd / g;
            bx = j / gz;
            bz = j - bx * gz;
        }
    }
    int t0, t1;
    mb_k_range(a.K16, ks, ksplit, t0, t1);
    const int nst = (t1 - t0) / KS;                    // an even range (launcher)
    const int zb0 = bz * GEO::BLOCKS;

    // ---- DMA pieces of this wave: p = wave + NW i; i < NPA: weight piece (k-tile p / RBV, row-block p % RBV), else an x piece
    const bf16x8* wg_w = a.planned ? (const bf16x8*)a.wp + (size_t)bx * (unsigned)a.wg_chunks
                                   : (const bf16x8*)a.wp + (size_t)bx * RBV * a.K16 * 64;
    const __amdgpu_buffer_rsrc_t rs_w = dma_rsrc(wg_w), rs_x = dma_rsrc(a.xp);
    unsigned gstr[NP], voff[NP];
    int pkk[NP], pdst[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = wave + NW * i;
        if (i < NPA) {
            const int kk = p / RBV, rb = p % RBV;
            if (a.planned) {
                const int nvb = a.nvl[rb];
                const int rr = (lane & 31) < nvb ? (lane & 31) : nvb - 1;
                gstr[i] = (unsigned)(32 * nvb);
                voff[i] = ((unsigned)a.boff[rb] + (unsigned)((lane >> 5) * nvb + rr)) * 16u;
            } else {
                gstr[i] = 1024u;
                voff[i] = ((unsigned)(rb * a.K16 * 64) + (unsigned)lane) * 16u;
            }
            pkk[i] = kk;
            pdst[i] = (kk * RBV + rb) * 1024;
        } else {
            const int q = p - A_STAGE, kk = q / NTBP, tb = q % NTBP;
            int xb = zb0 + (tb >> 1);
            xb = xb < a.nblk ? xb : a.nblk - 1;
            gstr[i] = 2048u;
            voff[i] = ((unsigned)(((xb * a.K16) * 2 + (tb & 1)) * 64) + (unsigned)lane) * 16u;
            pkk[i] = kk;
            pdst[i] = (A_STAGE + kk * NTBP + tb) * 1024;
        }
    }
    // the per-piece scalars are wave-uniform by construction; say so, or hipcc keeps the k-tile stride of the planned / classic select in
    // a VGPR and wraps every weight piece in a v_readfirstlane waterfall loop (GPU call 3 of round 5: +8 us on the 512-row gate/up launch)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        gstr[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)gstr[i]);
        pkk[i] = __builtin_amdgcn_readfirstlane(pkk[i]);
        pdst[i] = __builtin_amdgcn_readfirstlane(pdst[i]);
    }
    auto issue_one = [&](int sidx, int i) {
        int kt = t0 + sidx * KS + pkk[i];
        kt = kt < t1 ? kt : t1 - 1;
        char* d = lds_raw + (sidx % NR) * (STAGE * 1024) + pdst[i];
        const unsigned so = (unsigned)kt * gstr[i];
        // WPOL (compile-time: no branch in the loop): 0 = default cache policy — the paired forms, where the other token group reads the same
        // rows; 2 = nt — one region x all token blocks, every weight byte is read once
        if (i < NPA) dma_piece<WPOL>(rs_w, d, voff[i], so);
        else dma_piece<0>(rs_x, d, voff[i], so);
    };
    // STG: piece i of stage sidx -> registers (the same addresses and cache policies as the DMA form), and registers -> its LDS slot
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    auto load_piece = [&](int sidx, int i) -> u32x4_t {
        int kt = t0 + sidx * KS + pkk[i];
        kt = kt < t1 ? kt : t1 - 1;
        const unsigned so = (unsigned)kt * gstr[i];
        if (i < NPA) return __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)voff[i], (int)so, WPOL);
        return __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)voff[i], (int)so, 0);
    };
    auto store_piece = [&](int sidx, int i, const u32x4_t& v) {
        *(u32x4_t*)(lds_raw + (sidx & 1) * (STAGE * 1024) + pdst[i] + lane * 16) = v;
    };

    f32x16 acc[RPW][TW];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][t][i] = 0.f;

    const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds_raw + (unsigned)lane * 16u;
    constexpr int NRD = RPW + TW, NMMA = RPW * TW;
    constexpr int NG = NMMA > NRD ? NMMA : NRD;          // issue groups per half stage (TW = 1: 5 reads behind 4 MFMAs)
    // fragment j of k-tile (sidx, kk): the RPW weight fragments of the row group, then the TW x fragments of the token group
    auto read_one = [&](int sidx, int kk, int j, bf16x8 (&fa)[RPW], bf16x8 (&fb)[TW]) {
        const unsigned S = lds0 + (unsigned)((sidx % NR) * (STAGE * 1024));
        const unsigned A = S + (unsigned)((kk * RBV + RPW * rg) * 1024);
        const unsigned B = S + (unsigned)((A_STAGE + kk * NTBP + tq * TW) * 1024);
        if (j < RPW) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[j]) : "v"(A), "n"(j * 1024) : "memory");
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[j - RPW]) : "v"(B), "n"((j - RPW) * 1024) : "memory");
    };
    // MFMA m of a k-tile: row-block fastest, so that consecutive MFMAs write different accumulators
    auto mma = [&](int m, const bf16x8 (&fa)[RPW], const bf16x8 (&fb)[TW]) {
        const int r = m % RPW, t = m / RPW;
        acc[r][t] = LA_MFMA(fa[r], fb[t], acc[r][t], 0, 0, 0);
    };

    bf16x8 fa0[RPW], fb0[TW], fa1[RPW], fb1[TW];
    constexpr int H2 = NP - H;
    if constexpr (STG == 0) {
    #pragma unroll
        for (int i = 0; i < NR - 1; ++i)
    #pragma unroll
            for (int q = 0; q < NP; ++q) issue_one(i, q);
        vm_wait<(NR - 2) * NP>();
        __builtin_amdgcn_s_barrier();
    #pragma unroll
        for (int j = 0; j < NRD; ++j) read_one(0, 0, j, fa0, fb0);
        for (int s = 0; s < nst; ++s) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // set0 (read during the previous half) is complete
            __builtin_amdgcn_sched_barrier(0);
            // first half: MFMAs of set0 | one fragment read of set1 (stage s, k-tile 1) | one DMA piece of stage s + 3 after each
    #pragma unroll
            for (int m = 0; m < NG; ++m) {
                if (m < NMMA) mma(m, fa0, fb0);
                if (m < NRD) read_one(s, 1, m, fa1, fb1);
                if (m < H) issue_one(s + NR - 1, m);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            vm_wait<(NR - 3) * NP + H>();                                // own pieces of stage s + 1 landed
            __builtin_amdgcn_s_barrier();                                // stage s + 1 complete for everyone; the slot of stage s is free
            __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
            for (int m = 0; m < NG; ++m) {
                if (m < NMMA) mma(m, fa1, fb1);
                if (m < NRD) read_one(s + 1, 0, m, fa0, fb0);
                if (m < H2) issue_one(s + NR - 1, H + m);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

    } else if constexpr (STG == 1) {
        // ---- register-staged pipeline (see the kernel header): sets sA / sB hold the stages of even / odd parity that are in flight
        u32x4_t sA[NP], sB[NP];
        // (the loads stay in piece order, here as in the loop: hipcc merges the vmcnt bookkeeping of the two loop entries, and a prologue it
        //  had reordered made the loop wait for loads half a stage old — vmcnt(1) where vmcnt(8) is exact)
#pragma unroll
        for (int q = 0; q < NP; ++q) { sA[q] = load_piece(0, q); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int q = 0; q < NP; ++q) { sB[q] = load_piece(1, q); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int q = 0; q < NP; ++q) store_piece(0, q, sA[q]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NP; ++q) { sA[q] = load_piece(2, q); __builtin_amdgcn_sched_barrier(0); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < NRD; ++j) read_one(0, 0, j, fa0, fb0);
        // one stage: `cur` = the register set of stage s + 1 (written to LDS in the first half, refilled with stage s + 3 in the second)
        auto stage = [&](int s, auto parity, u32x4_t (&cur)[NP]) {
            constexpr int P = decltype(parity)::value;               // s % 2, compile-time: the LDS slots are immediates
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // set0 (read during the previous half) is complete
            __builtin_amdgcn_sched_barrier(0);
            constexpr int NG1 = NG > NP ? NG : NP;
#pragma unroll
            for (int m = 0; m < NG1; ++m) {
                if (m < NMMA) mma(m, fa0, fb0);
                if (m < NRD) read_one(P, 1, m, fa1, fb1);
                if (m < NP) store_piece(1 - P, m, cur[m]);           // (the compiler waits for exactly this load: vmcnt counts the younger ones)
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the reads of stage s and the writes of stage s + 1 are complete
            __builtin_amdgcn_s_barrier();                            // stage s + 1 is visible to everyone
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NG1; ++m) {
                if (m < NMMA) mma(m, fa1, fb1);
                if (m < NRD) read_one(1 - P, 0, m, fa0, fb0);
                if (m < NP) cur[m] = load_piece(s + 3, m);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        int s = 0;
        for (; s + 1 < nst; s += 2) {
            stage(s, std::integral_constant<int, 0>{}, sB);
            stage(s + 1, std::integral_constant<int, 1>{}, sA);
        }
        if (s < nst) stage(s, std::integral_constant<int, 0>{}, sB);
    } else {
        // ---- STG = 2 (round 6, late; lab knob 36, the default of the two-token-tile slab launch at 5-8 blocks): x DIRECT.  In every form with one row group (RG = 1: the slab, QKV and one-region gate/up launches)
        // the four waves share the weight row-blocks and each owns ITS token tiles — the x fragments are private to a wave, and a 1 KiB piece of the x
        // image IS the B-operand fragment.  So x goes global -> VGPR (a four-stage register ring, as the weights of k_gemm_fatd) and only the shared
        // operand, the weights, goes through the LDS ring: A_STAGE / 4 LDS-DMA pieces per wave and stage instead of (A_STAGE + B_STAGE) / 4, RPW fragment
        // reads per k-tile instead of RPW + TW.  Same MFMA chain per output element: bit-identical.  Measured on every form it fits (profiles/
        // r06_gateup_direct_weights.txt): -4.4 % per launch for <4, 2, SLAB> at 512 rows, level or slower for the quarters, QKV and gate/up forms.
        static_assert(GEO::RG == 1 && NPA >= 1, "x direct: every wave owns its token tiles");
        constexpr int DX = 4, HA = (NPA + 1) / 2, HA2 = NPA - HA;
        constexpr int NGX = NMMA > RPW + TW + HA ? NMMA : RPW + TW + HA;
        unsigned xv[TW];
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            const int tbg = tq * TW + t;
            int xb = zb0 + (tbg >> 1);
            xb = xb < a.nblk ? xb : a.nblk - 1;
            xv[t] = ((unsigned)(((xb * a.K16) * 2 + (tbg & 1)) * 64) + (unsigned)lane) * 16u;
        }
        auto loadXf = [&](int sidx, int kk, int t) -> u32x4_t {
            int kt = t0 + sidx * KS + kk;
            kt = kt < t1 ? kt : t1 - 1;
            return __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)xv[t], (int)((unsigned)kt * 2048u), 0);
        };
        u32x4_t xr[DX][KS][TW];
        // prologue: the x loads of stages 0 .. 2 and k-tile 0 of stage 3 first (pinned: see k_gemm_fatd), then the weight pieces of stages 0 .. NR - 2
#pragma unroll
        for (int sidx = 0; sidx < DX; ++sidx)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                if (sidx == DX - 1 && kk == 1) continue;
#pragma unroll
                for (int t = 0; t < TW; ++t) { xr[sidx][kk][t] = loadXf(sidx, kk, t); __builtin_amdgcn_sched_barrier(0); }
            }
#pragma unroll
        for (int sidx = 0; sidx < NR - 1; ++sidx)
#pragma unroll
            for (int q = 0; q < NPA; ++q) { issue_one(sidx, q); __builtin_amdgcn_sched_barrier(0); }
        vm_wait<0>();
        __builtin_amdgcn_s_barrier();
        bf16x8 fbx[TW];                      // (unused operand of read_one)
#pragma unroll
        for (int j = 0; j < RPW; ++j) read_one(0, 0, j, fa0, fbx);
        auto mmaX = [&](int m, const bf16x8 (&fa)[RPW], const u32x4_t (&xb)[TW]) {
            const int r = m % RPW, t = m / RPW;
            acc[r][t] = LA_MFMA(fa[r], __builtin_bit_cast(bf16x8, xb[t]), acc[r][t], 0, 0, 0);
        };
        auto stage = [&](int s, auto uc, auto tailc) {
            constexpr int U = decltype(uc)::value;
            constexpr bool TAIL = decltype(tailc)::value;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NGX; ++m) {
                if (m < NMMA) mmaX(m, fa0, xr[U][0]);
                if (m < RPW) read_one(s, 1, m, fa1, fbx);
                if (m >= RPW && m < RPW + TW) xr[(U + 3) % DX][1][m - RPW] = loadXf(s + 3, 1, m - RPW);          // freed by the second half of stage s - 1
                if (m >= RPW + TW && m < RPW + TW + HA) issue_one(s + NR - 1, m - RPW - TW);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // own weight pieces of stage s + 1 (issued in stage s - 2, NR = 4; NR = 3: in stage s - 1).  Younger VMEM operations: what stage s - 2 issued
            // behind its last piece (the x loads of its second half when that half issues no piece), the 2 TW + NPA of stage s - 1, the TW + HA of this half.
            // The single stages of a remainder drain the queue (dead loads are dropped there: k_gemm_fatd).
            if constexpr (TAIL) vm_wait<0>();
            else if constexpr (NR == 4) vm_wait<(HA2 > 0 ? 0 : TW) + 2 * TW + NPA + TW + HA>();
            else vm_wait<(HA2 > 0 ? 0 : TW) + TW + HA>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NGX; ++m) {
                if (m < NMMA) mmaX(m, fa1, xr[U][1]);
                if (m < RPW) read_one(s + 1, 0, m, fa0, fbx);
                if (m >= RPW && m < RPW + TW) xr[U][0][m - RPW] = loadXf(s + DX, 0, m - RPW);                    // freed by the first half of this stage
                if (m >= RPW + TW && m < RPW + TW + HA2) issue_one(s + NR - 1, HA + m - RPW - TW);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        int s = 0;
        for (; s + 4 <= nst; s += 4) {
            stage(s, std::integral_constant<int, 0>{}, std::false_type{});
            stage(s + 1, std::integral_constant<int, 1>{}, std::false_type{});
            stage(s + 2, std::integral_constant<int, 2>{}, std::false_type{});
            stage(s + 3, std::integral_constant<int, 3>{}, std::false_type{});
        }
        if (s < nst) stage(s, std::integral_constant<int, 0>{}, std::true_type{});
        if (s + 1 < nst) stage(s + 1, std::integral_constant<int, 1>{}, std::true_type{});
        if (s + 2 < nst) stage(s + 2, std::integral_constant<int, 2>{}, std::true_type{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    vm_wait<0>();

    const int tl = lane & 31, hh = lane >> 5;
    if constexpr (EPI == MB_SLAB) {
        // split-K partial sums straight from the accumulators (the layout k_row_norm_mb sums: [ks][row][N])
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            const int tbg = tq * TW + t, blk = zb0 + (tbg >> 1), tok = (tbg & 1) * 32 + tl;
            if (blk >= a.nblk) continue;
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    const f32x4 v = {acc[r][t][4 * gi], acc[r][t][4 * gi + 1], acc[r][t][4 * gi + 2], acc[r][t][4 * gi + 3]};
                    float* o = a.slabs + ((size_t)ks * a.M + blk * 64 + tok) * a.N + (bx * RBV + RPW * rg + r) * 32 + 8 * gi + 4 * hh;
                    *(f32x4*)o = v;
                }
        }
    } else if constexpr (EPI == MB_QKV) {
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            const int tbg = tq * TW + t, blk = zb0 + (tbg >> 1), tok = (tbg & 1) * 32 + tl;
            if (blk >= a.nblk) continue;
#pragma unroll
            for (int j = 0; j < RPW / 2; ++j)            // region (RBV / 2) x + j = row-blocks {2 j, 2 j + 1} = its {lo, hi} halves
                fat_qkv_tile<RV>(a, acc[2 * j][t], acc[2 * j + 1][t], bx * (RBV / 2) + (RPW / 2) * rg + j, blk, tok, hh);
        }
    } else {
        // ---- SwiGLU epilogue, as k_gemm_wide<8, TW, MB_SWIGLU>: act = bf16(silu(bf16(g)) * bf16(u)) parked as tile[token][sh + feature -
        //      lo] in the drained ring, then 16-byte chunks of the activation image
        constexpr int NREG = RBV / 4;                          // planned regions per workgroup
        const int sw_lo = a.R * NREG * bx, sw_r = NREG * a.R, sw_sh = sw_lo & 7;
        const int sw_nch = (sw_sh + sw_r + 7) >> 3;
        const int sw_stride = (sw_nch * 8) % 64 == 0 ? sw_nch * 8 + 8 : sw_nch * 8;
        __syncthreads();
        bf16_t* tile = (bf16_t*)lds_raw;
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            const int tbg = tq * TW + t, blk = zb0 + (tbg >> 1), tok = (tbg & 1) * 32 + tl;
            if (blk >= a.nblk) continue;
            const int trow = (tbg >> 1) * 64 + tok;
#pragma unroll
            for (int q = 0; q < 2; ++q) {                    // {G_q, U_q} = row-blocks q and q + 2 of the region
                const int nvg = a.nv[RPW * rg + q];
                const int c0 = sw_sh + rg * a.R + 32 * q;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (8 * (i >> 2) >= nvg) break;
                    const int f = 8 * (i >> 2) + 4 * hh + (i & 3);
                    const float gv = bfr_hw(acc[q][t][i]), uv = bfr_hw(acc[q + 2][t][i]);
                    const float sv = bfr_hw(gv * __builtin_amdgcn_rcpf(1.0f + __expf(-gv)));
                    const bf16_t o = f2bf_hw(sv * uv);
                    if (f < nvg) tile[trow * sw_stride + c0 + f] = o;
                }
            }
        }
        __syncthreads();
        const int ntok = GEO::BLOCKS * 64;
        for (int it = threadIdx.x; it < ntok * sw_nch; it += NW * 64) {
            const int trow = it % ntok, c = it / ntok, blk = zb0 + (trow >> 6);
            if (blk >= a.nblk) continue;
            const int fa_ = (sw_lo & ~7) + 8 * c;
            bf16_t* dst = a.act_xp + (size_t)blk * 64 * a.N + xp_offset(trow & 63, fa_);
            const bf16_t* srcp = tile + trow * sw_stride + 8 * c;
            if (fa_ >= sw_lo && fa_ + 8 <= sw_lo + sw_r) {
                *(bf16x8*)dst = *(const bf16x8*)srcp;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (fa_ + e >= sw_lo && fa_ + e < sw_lo + sw_r) dst[e] = srcp[e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_gemm_fatd (round 6, late; default, lab knob 35 = 0 turns it off): the paired gate/up launch at 5-8 blocks with the WEIGHTS streamed straight
// into MFMA operand registers instead of through LDS.
//   Geometry per workgroup as k_gemm_fat<8, TW, MB_SWIGLU>: two adjacent planned regions {G0,G1,U0,U1} x 2 (row-blocks 0..7) x 2 TW token
//   tiles (TW = 4: 4 blocks = 256 rows, TW = 3: 192; grid.z = the two token halves).  The four waves split the ROW-BLOCKS, not the tokens: wave (rg, q) owns the
//   SwiGLU pair {G_q, U_q} of region rg = row-blocks {4 rg + q, 4 rg + q + 2} over ALL 8 token tiles: 2 x 8 accumulator tiles (256
//   registers, as before).  No other wave needs its weight fragments, so they never visit LDS: per k-tile a wave issues 2 buffer loads
//   (1 KiB each, MFMA A-operand order as packed) into a 4-stage register ring and 8 ds_reads of the x fragments all four waves share —
//   8 reads per 16 MFMAs as before (4 + 4), but HALF the LDS-DMA pieces (x only: 4 per wave and stage instead of 8) and half the LDS
//   write traffic.  What this targets: an LDS-DMA piece blocks its wave's issue for 60-185 cycles and with one wave per SIMD nothing else
//   issues meanwhile (k_gemm_fat header); a plain buffer load issues in a few cycles.
//   Same MFMA chain per output element (k-tiles ascending into one accumulator): bit-identical to k_gemm_fat / k_gemm_wide.
//   Any even number of k-tiles: whole groups of four stages, then 1-3 single stages.
//   The same split for the slab launches (8 row-blocks x 4 token tiles per workgroup, wave = 2 row-blocks x 4 tiles) was built and measured:
//   43.8 -> 42.6 us per launch, -0.6 % per Mistral bs=8 step — not kept; neither were the QKV form of it (four regions x 2 / 4 token tiles: +10 % per
//   launch on GQA / 13B images, where the launch is bound by the weight bytes a CU pulls) and a partner prefetch of the weight stream (each of the two
//   workgroups of a column touching every other k-tile seven stages ahead: +3.5 % per launch) — profiles/r06_gateup_direct_weights.txt.
// ---------------------------------------------------------------------------------------------------------------
template <int TW, int WPOL>
__global__ __launch_bounds__(256) void k_gemm_fatd(MbArgs a) {
    constexpr int KS = 2, NW = 4, NTB = 2 * TW, B_STAGE = KS * NTB, NR = 4, NPB = B_STAGE / NW, DA = 4;
    constexpr int HB = (NPB + 1) / 2;                    // x pieces issued in the first half of a stage (the rest in the second)
    static_assert(B_STAGE % NW == 0 && NPB >= 2 && NTB + 2 + HB <= 2 * NTB, "x pieces per wave; issue slots of a half stage");
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rg = wave >> 1, q = wave & 1;
    const int bx = blockIdx.x, bz = blockIdx.z;
    const int t1 = a.K16;
    const int nst = t1 / KS;
    const int zb0 = bz * (NTB / 2);
    const bf16x8* wg_w = (const bf16x8*)a.wp + (size_t)bx * (unsigned)a.wg_chunks;
    const __amdgpu_buffer_rsrc_t rs_w = dma_rsrc(wg_w), rs_x = dma_rsrc(a.xp);
    // weights: row-block 4 rg + q + 2 j (j = 0 gate, 1 up) of the planned pair
    unsigned a_gstr[2], a_voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rb = 4 * rg + q + 2 * j;
        const int nvb = a.nvl[rb];
        const int rr = (lane & 31) < nvb ? (lane & 31) : nvb - 1;
        a_gstr[j] = (unsigned)__builtin_amdgcn_readfirstlane(32 * nvb);
        a_voff[j] = ((unsigned)a.boff[rb] + (unsigned)((lane >> 5) * nvb + rr)) * 16u;
    }
    auto loadA = [&](int sidx, int kk, int j) -> u32x4_t {
        int kt = sidx * KS + kk;
        kt = kt < t1 ? kt : t1 - 1;
        return __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)a_voff[j], (int)((unsigned)kt * a_gstr[j]), WPOL);
    };
    // x pieces of this wave: p = wave + NW i -> (k-tile p / NTB of the stage, token tile p % NTB)
    unsigned b_voff[NPB];
    int b_kk[NPB], b_dst[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int p = wave + NW * i, kk = p / NTB, tb = p % NTB;
        int xb = zb0 + (tb >> 1);
        xb = xb < a.nblk ? xb : a.nblk - 1;
        b_voff[i] = ((unsigned)(((xb * a.K16) * 2 + (tb & 1)) * 64) + (unsigned)lane) * 16u;
        b_kk[i] = __builtin_amdgcn_readfirstlane(kk);
        b_dst[i] = __builtin_amdgcn_readfirstlane((kk * NTB + tb) * 1024);
    }
    auto issueB = [&](int sidx, int i) {
        int kt = sidx * KS + b_kk[i];
        kt = kt < t1 ? kt : t1 - 1;
        dma_piece<0>(rs_x, lds_raw + (sidx % NR) * (B_STAGE * 1024) + b_dst[i], b_voff[i], (unsigned)kt * 2048u);
    };
    f32x16 acc[2][NTB];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < NTB; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][t][i] = 0.f;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds_raw + (unsigned)lane * 16u;
    auto readB = [&](int sidx, int kk, int j, bf16x8 (&fb)[NTB]) {
        const unsigned B = lds0 + (unsigned)((sidx % NR) * (B_STAGE * 1024) + kk * NTB * 1024);
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[j]) : "v"(B), "n"(j * 1024) : "memory");
    };
    u32x4_t ra[DA][KS][2];
    bf16x8 fb0[NTB], fb1[NTB];
    // prologue: the weights of stages 0 .. 2 and k-tile 0 of stage 3 FIRST, then x of stages 0 .. NR - 2 — every load pinned in this order.
    // hipcc merges the vmcnt bookkeeping of the loop's two entries: a stage-0 weight load followed by fewer VMEM operations here than on
    // the back edge (26) would make the loop head drain the queue on every pass (first build: vmcnt(3) at the head of every fourth stage)
#pragma unroll
    for (int sidx = 0; sidx < DA; ++sidx)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (sidx == DA - 1 && kk == 1) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) { ra[sidx][kk][j] = loadA(sidx, kk, j); __builtin_amdgcn_sched_barrier(0); }
        }
#pragma unroll
    for (int sidx = 0; sidx < NR - 1; ++sidx)
#pragma unroll
        for (int i = 0; i < NPB; ++i) { issueB(sidx, i); __builtin_amdgcn_sched_barrier(0); }
    // own x pieces of stages 0 AND 1 landed (and every weight load of the prologue with them): the mid-stage wait of the loop counts 12 younger
    // operations behind the pieces of stage s + 1, and at s = 0 only 8 exist (x of stage 2 + the first half's four)
    vm_wait<(NR - 3) * NPB>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < NTB; ++j) readB(0, 0, j, fb0);
    // one stage; U = s % DA is a compile-time constant through the 4-fold unroll, so every register-ring index is static
    // TAIL: one of the 1-3 single stages behind the four-stage groups.  The compiler drops the weight loads such a stage would issue for stages
    // that do not exist there (dead), so the hand-counted vmcnt of the mid-stage wait does not hold in those copies of the body: they drain the
    // queue instead (first build of the remainder path: wrong sums on every reduction with a remainder in the sibling kernels)
    auto stage = [&](int s, auto uc, auto tailc) {
        constexpr int U = decltype(uc)::value;
        constexpr bool TAIL = decltype(tailc)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // fb0 (read during the previous half) is complete
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2 * NTB; ++m) {
            const int r = m & 1, t = m >> 1;
            acc[r][t] = LA_MFMA(__builtin_bit_cast(bf16x8, ra[U][0][r]), fb0[t], acc[r][t], 0, 0, 0);
            if (m < NTB) readB(s, 1, m, fb1);
            if (m == NTB || m == NTB + 1) ra[(U + 3) % DA][1][m - NTB] = loadA(s + 3, 1, m - NTB);        // freed by the second half of stage s - 1
            if (m >= NTB + 2 && m < NTB + 2 + HB) issueB(s + NR - 1, m - NTB - 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // own x pieces of stage s + 1: issued in stage s - 2, the last of them as the last VMEM operation of that stage; younger: the 4 + NPB of
        // stage s - 1 and the 2 + HB of this half
        if constexpr (TAIL) vm_wait<0>(); else vm_wait<4 + NPB + 2 + HB>();
        __builtin_amdgcn_s_barrier();                                   // stage s + 1 complete for everyone; the slot of stage s - 1 is free
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2 * NTB; ++m) {
            const int r = m & 1, t = m >> 1;
            acc[r][t] = LA_MFMA(__builtin_bit_cast(bf16x8, ra[U][1][r]), fb1[t], acc[r][t], 0, 0, 0);
            if (m < NTB) readB(s + 1, 0, m, fb0);
            if (m == NTB || m == NTB + 1) ra[U][0][m - NTB] = loadA(s + DA, 0, m - NTB);                   // freed by the first half of this stage
            if (m >= NTB + 2 && m < NTB + 2 + NPB - HB) issueB(s + NR - 1, HB + m - NTB - 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the fragment reads of this half are inline asm: the compiler does not know their results are still in flight.  Inside the loop nothing
        // touches them before the next stage's wait, but at the loop's exit into the 1-3 single stages it may copy registers (phi moves) — of values
        // that have not landed (seen: bitwise failures of a sibling kernel on reductions with a remainder).  So a stage ends with its reads complete.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    int s = 0;
    for (; s + 4 <= nst; s += 4) {
        stage(s, std::integral_constant<int, 0>{}, std::false_type{});
        stage(s + 1, std::integral_constant<int, 1>{}, std::false_type{});
        stage(s + 2, std::integral_constant<int, 2>{}, std::false_type{});
        stage(s + 3, std::integral_constant<int, 3>{}, std::false_type{});
    }
    if (s < nst) stage(s, std::integral_constant<int, 0>{}, std::true_type{});                // 1-3 stages left (s is a multiple of 4: ring positions 0, 1, 2)
    if (s + 1 < nst) stage(s + 1, std::integral_constant<int, 1>{}, std::true_type{});
    if (s + 2 < nst) stage(s + 2, std::integral_constant<int, 2>{}, std::true_type{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    vm_wait<0>();

    // ---- SwiGLU epilogue (as k_gemm_fat<8, TW, MB_SWIGLU>): act = bf16(silu(bf16(g)) * bf16(u)) parked as tile[token][sh + feature - lo]
    const int tl = lane & 31, hh = lane >> 5;
    const int sw_lo = a.R * 2 * bx, sw_r = 2 * a.R, sw_sh = sw_lo & 7;
    const int sw_nch = (sw_sh + sw_r + 7) >> 3;
    const int sw_stride = (sw_nch * 8) % 64 == 0 ? sw_nch * 8 + 8 : sw_nch * 8;
    __syncthreads();
    bf16_t* tile = (bf16_t*)lds_raw;
    const int nvg = a.nv[4 * rg + q];
    const int c0 = sw_sh + rg * a.R + 32 * q;
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
        const int blk = zb0 + (t >> 1), tok = (t & 1) * 32 + tl;
        if (blk >= a.nblk) continue;
        const int trow = (t >> 1) * 64 + tok;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (8 * (i >> 2) >= nvg) break;
            const int f = 8 * (i >> 2) + 4 * hh + (i & 3);
            const float gv = bfr_hw(acc[0][t][i]), uv = bfr_hw(acc[1][t][i]);
            const float sv = bfr_hw(gv * __builtin_amdgcn_rcpf(1.0f + __expf(-gv)));
            const bf16_t o = f2bf_hw(sv * uv);
            if (f < nvg) tile[trow * sw_stride + c0 + f] = o;
        }
    }
    __syncthreads();
    const int ntok = (NTB / 2) * 64;
    for (int it = threadIdx.x; it < ntok * sw_nch; it += NW * 64) {
        const int trow = it % ntok, c = it / ntok, blk = zb0 + (trow >> 6);
        if (blk >= a.nblk) continue;
        const int fa_ = (sw_lo & ~7) + 8 * c;
        bf16_t* dst = a.act_xp + (size_t)blk * 64 * a.N + xp_offset(trow & 63, fa_);
        const bf16_t* srcp = tile + trow * sw_stride + 8 * c;
        if (fa_ >= sw_lo && fa_ + 8 <= sw_lo + sw_r) {
            *(bf16x8*)dst = *(const bf16x8*)srcp;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (fa_ + e >= sw_lo && fa_ + e < sw_lo + sw_r) dst[e] = srcp[e];
        }
    }
}
#define LA_FATD_LDS (96 * 1024)

// ---------------------------------------------------------------------------------------------------------------
// Row kernels over M rows: embedding gather + RMSNorm / residual + split-K slab sum + RMSNorm (same arithmetic and rounding
// points as k_row_norm in la_kernels.hip; LlamaRMSNorm :76-90, LlamaDecoderLayer residual adds :352-363).
// ---------------------------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(512) void k_row_norm_mb(const bf16_t* __restrict__ embed, const int* __restrict__ ids,
                                                      bf16_t* __restrict__ h, const float* __restrict__ slabs,
                                                      const bf16_t* __restrict__ nw, int hidden, float eps,
                                                      bf16_t* __restrict__ xp, int slab_rows, int cast_first) {
    __shared__ float sh[8];
    const int t = blockIdx.x;
    const int nchunk = hidden >> 3;
    const bf16_t* src = embed ? embed + (size_t)ids[t] * hidden : h + (size_t)t * hidden;
    bf16x8 hv[2], wv[2];
    f32x4 sl[NS > 0 ? NS : 1][2][2];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int c = threadIdx.x + ci * 512;
        if (c < nchunk) {
            hv[ci] = *(const bf16x8*)(src + c * 8);
            wv[ci] = *(const bf16x8*)(nw + c * 8);
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                const float* sp = slabs + ((size_t)s2 * slab_rows + t) * hidden + c * 8;
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
                }
                vals[ci][j] = v;
                ho[j] = (short)f2bf(v);
                ss += v * v;
            }
            *(bf16x8*)(h + (size_t)t * hidden + c * 8) = ho;
        }
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) tot += sh[i];
    const float rs = 1.0f / sqrtf(tot / (float)hidden + eps);
    bf16_t* xo_base = xp + (size_t)(t >> 6) * 64 * hidden;
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int c = threadIdx.x + ci * 512;
        if (c < nchunk) {
            bf16x8 xo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float nv = cast_first ? bfr(vals[ci][j] * rs) : vals[ci][j] * rs;
                xo[j] = (short)f2bf(bf2f((bf16_t)wv[ci][j]) * nv);
            }
            *(bf16x8*)(xo_base + xp_offset(t & 63, c * 8)) = xo;
        }
    }
}

// Post-attention norm of a Mixtral layer over M rows: residual + split-K slabs + RMSNorm, then the router
// (MixtralSparseMoeBlock.forward, mixtral/modeling_mixtral.py:723-729: logits = gate(x) in the activation dtype, softmax in fp32,
// top-k, renormalise, cast back) -> route_w[t][e] = weight of expert e for row t, 0 if not routed (rows >= the block's T: 0).
// Same arithmetic, summation order and rounding points as k_row_norm<NS, true> of the 64-row path.
template <int NS>
__global__ __launch_bounds__(512) void k_row_norm_router_mb(bf16_t* __restrict__ h, const float* __restrict__ slabs, const bf16_t* __restrict__ nw,
                                                             int hidden, float eps, bf16_t* __restrict__ xp, int slab_rows, int cast_first,
                                                             const bf16_t* __restrict__ wrouter, int n_experts, int top_k,
                                                             float* __restrict__ route_w, const int* __restrict__ meta) {
    __shared__ float sh[8];
    __shared__ float shr[8][LA_MOE_MAX_E];
    const int t = blockIdx.x;
    const int nchunk = hidden >> 3;
    bf16x8 hv[2], wv[2];
    bf16x8 gwv[LA_MOE_MAX_E][2];
    f32x4 sl[NS][2][2];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int c = threadIdx.x + ci * 512;
        if (c < nchunk) {
            hv[ci] = *(const bf16x8*)(h + (size_t)t * hidden + c * 8);
            wv[ci] = *(const bf16x8*)(nw + c * 8);
#pragma unroll
            for (int e = 0; e < LA_MOE_MAX_E; ++e)
                if (e < n_experts) gwv[e][ci] = *(const bf16x8*)(wrouter + (size_t)e * hidden + c * 8);
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                const float* sp = slabs + ((size_t)s2 * slab_rows + t) * hidden + c * 8;
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
                float add = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < NS; ++s2) add += sl[s2][ci][j >> 2][j & 3];
                v = bfr(v + bfr(add));
                vals[ci][j] = v;
                ho[j] = (short)f2bf(v);
                ss += v * v;
            }
            *(bf16x8*)(h + (size_t)t * hidden + c * 8) = ho;
        }
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) tot += sh[i];
    const float rs = 1.0f / sqrtf(tot / (float)hidden + eps);
    float rl[LA_MOE_MAX_E];
#pragma unroll
    for (int e = 0; e < LA_MOE_MAX_E; ++e) rl[e] = 0.f;
    bf16_t* xo_base = xp + (size_t)(t >> 6) * 64 * hidden;
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int c = threadIdx.x + ci * 512;
        if (c < nchunk) {
            bf16x8 xo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float nv = cast_first ? bfr(vals[ci][j] * rs) : vals[ci][j] * rs;
                xo[j] = (short)f2bf(bf2f((bf16_t)wv[ci][j]) * nv);
            }
            *(bf16x8*)(xo_base + xp_offset(t & 63, c * 8)) = xo;
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
#pragma unroll
    for (int e = 0; e < LA_MOE_MAX_E; ++e) rl[e] = wave_sum(rl[e]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int e = 0; e < LA_MOE_MAX_E; ++e) shr[threadIdx.x >> 6][e] = rl[e];
    }
    __syncthreads();
    if (threadIdx.x < 64)
        moe_router_tail<LA_MOE_MAX_E>(shr, n_experts, top_k, (t & 63) < meta[(t >> 6) * LA_MB_META + LA_MBM_T], route_w + (size_t)t * LA_MOE_MAX_E);
}

// residual + accumulated expert outputs (bf16 + bf16) + next RMSNorm over M rows (k_row_norm<0> with an addend)
__global__ __launch_bounds__(512) void k_row_norm_addend_mb(bf16_t* __restrict__ h, const bf16_t* __restrict__ addend,
                                                             const bf16_t* __restrict__ nw, int hidden, float eps,
                                                             bf16_t* __restrict__ xp, int cast_first) {
    __shared__ float sh[8];
    const int t = blockIdx.x;
    const int nchunk = hidden >> 3;
    bf16x8 hv[2], wv[2], av[2];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int c = threadIdx.x + ci * 512;
        if (c < nchunk) {
            hv[ci] = *(const bf16x8*)(h + (size_t)t * hidden + c * 8);
            wv[ci] = *(const bf16x8*)(nw + c * 8);
            av[ci] = *(const bf16x8*)(addend + (size_t)t * hidden + c * 8);
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
                const float v = bfr(bf2f((bf16_t)hv[ci][j]) + bf2f((bf16_t)av[ci][j]));
                vals[ci][j] = v;
                ho[j] = (short)f2bf(v);
                ss += v * v;
            }
            *(bf16x8*)(h + (size_t)t * hidden + c * 8) = ho;
        }
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) tot += sh[i];
    const float rs = 1.0f / sqrtf(tot / (float)hidden + eps);
    bf16_t* xo_base = xp + (size_t)(t >> 6) * 64 * hidden;
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int c = threadIdx.x + ci * 512;
        if (c < nchunk) {
            bf16x8 xo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float nv = cast_first ? bfr(vals[ci][j] * rs) : vals[ci][j] * rs;
                xo[j] = (short)f2bf(bf2f((bf16_t)wv[ci][j]) * nv);
            }
            *(bf16x8*)(xo_base + xp_offset(t & 63, c * 8)) = xo;
        }
    }
}

// MoE accumulation over M rows (MixtralSparseMoeBlock.forward :731-756): final[t] = sum over the experts row t is routed to, in
// expert-index order, of bf16(bf16(expert_out[t]) * w[t][e]) (index_add_ into a bf16 buffer).  Same arithmetic as k_moe_accum_all.
// ---------------------------------------------------------------------------------------------------------------
// Gathered MoE (M >= 128 rows): instead of running every expert over ALL rows (4x the MLP flops at top-2 of 8), the rows an
// expert received are packed into their own 64-row blocks (ascending row order -> deterministic), the expert's GEMMs run over
// ceil(count / 64) blocks (decided on the device: MbArgs.nblk_dev), and the accumulation reads row t's result at its position
// in each selected expert's block list.  Same arithmetic per row as the dense form (MixtralSparseMoeBlock.forward:
// index_add of routing_weight * expert(x) in expert order).
//   k_moe_plan_mb: one workgroup; perm[e][i] = i-th row routed to e, pos[t][e] = i (or -1), cnt[e], nb[e] = ceil(cnt / 64)
//   k_moe_gather_mb: grid (E, blocks): the packed activation image of block j of expert e, 16-byte chunks (token, 8 features)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_moe_plan_mb(const float* __restrict__ route_w, int M, int n_experts, int* __restrict__ perm,
                                                      int* __restrict__ pos, int* __restrict__ cnt_nb) {
    __shared__ int wsum[8];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int e = 0; e < n_experts; ++e) {
        const bool on = t < M && route_w[(size_t)t * LA_MOE_MAX_E + e] != 0.f;
        const unsigned long long bal = __ballot(on);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int base = 0, total = 0;
        for (int w = 0; w < 8; ++w) { if (w < wave) base += wsum[w]; total += wsum[w]; }
        if (t < M) pos[(size_t)t * LA_MOE_MAX_E + e] = on ? base + before : -1;
        if (on) perm[(size_t)e * (LA_MB_MAX * 64) + base + before] = t;
        if (t == 0) { cnt_nb[e] = total; cnt_nb[LA_MOE_MAX_E + e] = (total + 63) >> 6; }
        __syncthreads();
    }
}

__global__ __launch_bounds__(512) void k_moe_gather_mb(const bf16_t* __restrict__ xp, const int* __restrict__ perm, const int* __restrict__ cnt_nb,
                                                        int hidden, bf16_t* __restrict__ xg, long xg_stride) {
    const int e = blockIdx.x, j = blockIdx.y;
    const int cnt = cnt_nb[e];
    if (j * 64 >= cnt) return;
    __shared__ int rows[64];
    if (threadIdx.x < 64) rows[threadIdx.x] = j * 64 + threadIdx.x < cnt ? perm[(size_t)e * (LA_MB_MAX * 64) + j * 64 + threadIdx.x] : -1;
    __syncthreads();
    const size_t blk_elems = (size_t)64 * hidden;
    bf16x8* dst = (bf16x8*)(xg + (size_t)e * xg_stride + (size_t)j * blk_elems);
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    // chunk index inside a block image: ((k-tile * 2 + token block) * 64 + lane'), lane' = (token & 31) + 32 * ((k >> 3) & 1)
    const int nchunk = 64 * (hidden >> 3), per = (nchunk + gridDim.z - 1) / gridDim.z;       // gridDim.z workgroups share a block
    const int cend = (blockIdx.z + 1) * per < nchunk ? (blockIdx.z + 1) * per : nchunk;
    for (int c = blockIdx.z * per + threadIdx.x; c < cend; c += 512) {
        const int lp = c & 63, tbk = (c >> 6) & 1, kt = c >> 7;
        const int tok = tbk * 32 + (lp & 31);
        const int r = rows[tok];
        bf16x8 v = z;
        if (r >= 0) {
            const int sb = r >> 6, st = r & 63;
            v = *((const bf16x8*)(xp + (size_t)sb * blk_elems) + ((kt * 2 + (st >> 5)) * 64 + (st & 31) + 32 * (lp >> 5)));
        }
        dst[c] = v;
    }
}

// k_moe_plan_mb + k_moe_gather_mb in ONE launch (round 5): every workgroup (e, j, z) derives expert e's row list itself — at most 512 route
// weights, one ballot pass — instead of waiting for the one-workgroup plan launch (5.5 us per layer at Mixtral bs=4, all of it latency);
// workgroup (e, 0, 0) publishes perm / pos / cnt / nb for the expert GEMMs and the accumulation.  Integer work: the same lists, the same image.
__global__ __launch_bounds__(512) void k_moe_plan_gather_mb(const float* __restrict__ route_w, int M, const bf16_t* __restrict__ xp, int hidden,
                                                             bf16_t* __restrict__ xg, long xg_stride, int* __restrict__ perm, int* __restrict__ pos,
                                                             int* __restrict__ cnt_nb) {
    __shared__ int wsum[8];
    __shared__ int rows[64];
    const int e = blockIdx.x, j = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool on = t < M && route_w[(size_t)t * LA_MOE_MAX_E + e] != 0.f;
    const unsigned long long bal = __ballot(on);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    if (t < 64) rows[t] = -1;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < 8; ++w) { if (w < wave) base += wsum[w]; total += wsum[w]; }
    const int idx = base + before;                     // position of row t in expert e's list (ascending row order)
    if (j == 0 && blockIdx.z == 0) {
        if (t < M) pos[(size_t)t * LA_MOE_MAX_E + e] = on ? idx : -1;
        if (on) perm[(size_t)e * (LA_MB_MAX * 64) + idx] = t;
        if (t == 0) { cnt_nb[e] = total; cnt_nb[LA_MOE_MAX_E + e] = (total + 63) >> 6; }
    }
    if (j * 64 >= total) return;                        // (uniform) this block of the expert is empty
    if (on && idx >= j * 64 && idx < j * 64 + 64) rows[idx - j * 64] = t;
    __syncthreads();
    const size_t blk_elems = (size_t)64 * hidden;
    bf16x8* dst = (bf16x8*)(xg + (size_t)e * xg_stride + (size_t)j * blk_elems);
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    const int nchunk = 64 * (hidden >> 3), per = (nchunk + gridDim.z - 1) / gridDim.z;       // as k_moe_gather_mb
    const int cend = (blockIdx.z + 1) * per < nchunk ? (blockIdx.z + 1) * per : nchunk;
    for (int c = blockIdx.z * per + threadIdx.x; c < cend; c += 512) {
        const int lp = c & 63, tbk = (c >> 6) & 1, kt = c >> 7;
        const int r = rows[tbk * 32 + (lp & 31)];
        bf16x8 v = z;
        if (r >= 0) {
            const int sb = r >> 6, st = r & 63;
            v = *((const bf16x8*)(xp + (size_t)sb * blk_elems) + ((kt * 2 + (st >> 5)) * 64 + (st & 31) + 32 * (lp >> 5)));
        }
        dst[c] = v;
    }
}

// k_moe_accum_mb + k_row_norm_addend_mb in ONE launch (round 3): the row's accumulated expert outputs never leave the registers
// on their way into the residual add and the next RMSNorm.  Same values at every rounding point: the accumulator is bf16-exact
// after every add (index_add_ into a bf16 buffer), so "store as bf16, load as bf16" between the two kernels was the identity.
template <int NS>
__global__ __launch_bounds__(512) void k_moe_accum_norm_mb(const float* __restrict__ slabs, long ex_slab, int slab_rows,
                                                            const float* __restrict__ route_w, int n_experts, int hidden,
                                                            const int* __restrict__ pos, bf16_t* __restrict__ h,
                                                            const bf16_t* __restrict__ nw, float eps, bf16_t* __restrict__ xp, int cast_first) {
    __shared__ float sh[8];
    const int t = blockIdx.x;
    const int nchunk = hidden >> 3;
    int sel[4], srow[4];
    float wsel[4];
    int ns = 0;
    for (int e = 0; e < n_experts && ns < 4; ++e) {
        const float w = route_w[(size_t)t * LA_MOE_MAX_E + e];
        if (w != 0.f) { sel[ns] = e; wsel[ns] = w; srow[ns] = pos ? pos[(size_t)t * LA_MOE_MAX_E + e] : t; ++ns; }
    }
    for (int k = ns; k < 4; ++k) { sel[k] = 0; wsel[k] = 0.f; srow[k] = 0; }
    float vals[2][8];
    bf16x8 wv[2];
    float ss = 0.f;
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int c = threadIdx.x + ci * 512;
        if (c < nchunk) {
            const bf16x8 hv = *(const bf16x8*)(h + (size_t)t * hidden + c * 8);
            wv[ci] = *(const bf16x8*)(nw + c * 8);
            float cur[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < ns) {
                    f32x4 v[NS][2];
#pragma unroll
                    for (int s2 = 0; s2 < NS; ++s2) {
                        const float* sp = slabs + (size_t)sel[k] * ex_slab + ((size_t)s2 * slab_rows + srow[k]) * hidden + c * 8;
                        v[s2][0] = *(const f32x4*)sp;
                        v[s2][1] = *(const f32x4*)(sp + 4);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float add = 0.f;
#pragma unroll
                        for (int s2 = 0; s2 < NS; ++s2) add += v[s2][j >> 2][j & 3];
                        const float contrib = bfr(bfr(add) * wsel[k]);
                        cur[j] = k ? bfr(cur[j] + contrib) : contrib;
                    }
                }
            bf16x8 ho;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = bfr(bf2f((bf16_t)hv[j]) + cur[j]);
                vals[ci][j] = v;
                ho[j] = (short)f2bf(v);
                ss += v * v;
            }
            *(bf16x8*)(h + (size_t)t * hidden + c * 8) = ho;
        }
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) tot += sh[i];
    const float rs = 1.0f / sqrtf(tot / (float)hidden + eps);
    bf16_t* xo_base = xp + (size_t)(t >> 6) * 64 * hidden;
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
        const int c = threadIdx.x + ci * 512;
        if (c < nchunk) {
            bf16x8 xo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float nv = cast_first ? bfr(vals[ci][j] * rs) : vals[ci][j] * rs;
                xo[j] = (short)f2bf(bf2f((bf16_t)wv[ci][j]) * nv);
            }
            *(bf16x8*)(xo_base + xp_offset(t & 63, c * 8)) = xo;
        }
    }
}

template <int NS>
__global__ __launch_bounds__(256) void k_moe_accum_mb(const float* __restrict__ slabs, long ex_slab, int slab_rows,
                                                       const float* __restrict__ route_w, int n_experts, int hidden,
                                                       bf16_t* __restrict__ acc, const int* __restrict__ pos) {
    const int t = blockIdx.x;
    int sel[4], srow[4];
    float wsel[4];
    int ns = 0;
    for (int e = 0; e < n_experts && ns < 4; ++e) {
        const float w = route_w[(size_t)t * LA_MOE_MAX_E + e];
        if (w != 0.f) { sel[ns] = e; wsel[ns] = w; srow[ns] = pos ? pos[(size_t)t * LA_MOE_MAX_E + e] : t; ++ns; }
    }
    for (int k = ns; k < 4; ++k) { sel[k] = 0; wsel[k] = 0.f; srow[k] = 0; }
    for (int c = threadIdx.x; c < (hidden >> 3); c += 256) {
        f32x4 v[4][NS][2];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < ns) {
#pragma unroll
                for (int s2 = 0; s2 < NS; ++s2) {
                    const float* sp = slabs + (size_t)sel[k] * ex_slab + ((size_t)s2 * slab_rows + srow[k]) * hidden + c * 8;
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

// ---------------------------------------------------------------------------------------------------------------
// Step inputs: per block ids / 64-bit ancestor masks / positions, and the block meta records.
// in: [LA_MIN_*] (include/lookahead_hip.h).  pos = committed keys of the slot + rows of earlier blocks of the same slot in
// this step (prefill chain) + popcount(rowmask) - 1  (model hook position_ids = mask.sum(-1) - 1, modeling_llama.py:584-588).
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_build_inputs_mb(const int* __restrict__ in, const int* __restrict__ bstate, int* __restrict__ meta,
                                  int* __restrict__ pos, unsigned long long* __restrict__ rowmask, int* __restrict__ ids) {
    const int b = blockIdx.x, t = threadIdx.x;       // 64 threads
    const int* rec = in + LA_MIN_BLK + 4 * b;
    int slot = rec[0];
    slot = slot < 0 ? 0 : (slot >= LA_MAX_SEQ ? LA_MAX_SEQ - 1 : slot);
    int T = rec[1];
    T = T < 1 ? 1 : (T > 64 ? 64 : T);
    int base = 0, first = b;
    for (int p = b - 1; p >= 0; --p) {
        const int* r2 = in + LA_MIN_BLK + 4 * p;
        if (r2[0] == slot) { int t2 = r2[1]; t2 = t2 < 1 ? 1 : (t2 > 64 ? 64 : t2); base += t2; first = p; }
    }
    const int nkeys = bstate[LA_BST_NKEYS + slot];
    const unsigned long long* rmin = (const unsigned long long*)(in + LA_MIN_ROWMASK) + b * 64;
    const unsigned long long rm = t < T ? rmin[t] : (1ull << t);
    rowmask[b * 64 + t] = rm;
    ids[b * 64 + t] = t < T ? in[LA_MIN_IDS + b * 64 + t] : 0;
    int p0 = nkeys + base + __popcll(rm) - 1;
    if (rec[2] == LA_MODE_TREE_PIECE) {
        // a later 64-row piece of a wide tree: depth = ancestors in the earlier pieces + in the own block (the hook
        // position_ids = mask.sum(-1) - 1 over the whole tree row, modeling_llama.py:584-588), not a chain offset
        const unsigned long long* xm = (const unsigned long long*)(in + LA_MIN_XMASK) + ((size_t)b * 64 + t) * 3;
        int anc = 0;
        const int np = b - first;                          // earlier pieces (<= 3)
        for (int q = 0; q < 3; ++q) if (q < np && t < T) anc += __popcll(xm[q]);
        p0 = nkeys + anc + __popcll(rm) - 1;
    }
    pos[b * 64 + t] = t < T ? p0 : 0;
    if (t == 0) {
        int* m = meta + b * LA_MB_META;
        m[LA_MBM_SLOT] = slot; m[LA_MBM_T] = T; m[LA_MBM_MODE] = rec[2]; m[LA_MBM_LIMIT] = rec[3];
        m[LA_MBM_NKEYS] = nkeys; m[LA_MBM_BASE] = base; m[LA_MBM_FIRST] = first;
    }
}

// Step input from the DEVICE trie (la_llama_mstep_trie): block b = query b of k_trie_hier_get, whose ids / row masks / count are
// copied into the step-input block where k_build_inputs_mb expects them; a query that returned nothing becomes the 1-row tree
// [last token] (the host loops' fallback, pretrained_model_batch.py:706-743).
struct MbTrieFill { int slot[LA_MB_MAX], limit[LA_MB_MAX], last[LA_MB_MAX]; };
__global__ void k_mb_fill_from_trie(const int* __restrict__ t_ids, const unsigned long long* __restrict__ t_rm,
                                    const int* __restrict__ t_n, MbTrieFill f, int nblk, int* __restrict__ in) {
    const int b = blockIdx.x, t = threadIdx.x;       // 64 threads
    int T = t_n[b];
    T = T < 0 ? 0 : (T > 64 ? 64 : T);
    const bool empty = T == 0;
    if (empty) T = 1;
    unsigned long long* rmo = (unsigned long long*)(in + LA_MIN_ROWMASK) + b * 64;
    if (t < T) {
        in[LA_MIN_IDS + b * 64 + t] = empty ? f.last[b] : t_ids[b * 64 + t];
        rmo[t] = empty ? 1ull : t_rm[b * 64 + t];
    }
    if (t == 0) {
        int* rec = in + LA_MIN_BLK + 4 * b;
        rec[0] = f.slot[b]; rec[1] = T; rec[2] = 0; rec[3] = f.limit[b];
        if (b == 0) in[LA_MIN_NBLK] = nblk;
    }
}
int lk_mb_fill_from_trie(hipStream_t st, const int* t_ids, const uint64_t* t_rm, const int* t_n, const int* slots, const int* limits,
                         const int* last, int nblk, int* d_in) {
    if (nblk < 1 || nblk > LA_MB_MAX) return -1;
    MbTrieFill f{};
    for (int b = 0; b < nblk; ++b) { f.slot[b] = slots[b]; f.limit[b] = limits[b]; f.last[b] = last[b]; }
    k_mb_fill_from_trie<<<nblk, 64, 0, st>>>(t_ids, (const unsigned long long*)t_rm, t_n, f, nblk, d_in);
    hipError_t e_ = hipGetLastError(); return e_ == hipSuccess ? 0 : (int)e_;
}

// ---------------------------------------------------------------------------------------------------------------
// Tree attention of block `blk` (LlamaAttention.forward, modeling_llama.py:270-296 under the rank-4 mask): committed keys of
// the block's slot mask-free, the fresh tiles of EARLIER blocks of the same slot in this step fully visible (prefill chain),
// the block's own 64 fresh keys under its ancestor masks.  Same tile arithmetic and rounding points as k_tree_attn.
// grid = (heads, key splits, blocks), 8 waves = 2 token blocks x 4 key-tile parities.
// ---------------------------------------------------------------------------------------------------------------
struct MbAttnArgs {
    const bf16_t* qf; const bf16_t* kmain; const bf16_t* vmain; const bf16_t* kfresh; const bf16_t* vfresh;
    const unsigned long long* rowmask;
    const unsigned long long* xmask;                   // [blk][64][3] wide-tree pieces (mode 3): masks over the earlier pieces' rows
    const int* meta;
    int nh, nkv, total_keys, slot_tiles, nsplit, window, ring;      // ring: the slot's main cache is a ring of slot_tiles tiles
    float qk;                                          // la_qk_scale(head_dim) (attn_scale, la_common.h)
    int rot;                                           // 1: the query heads of a kv head start their tile lists at different offsets (GQA)
    float* opart; float* mpart; float* lpart;          // [blk][nh][nsplit][64][128] ...
    bf16_t* attn_xp;                                   // nsplit == 1: the normalised output goes straight into o_proj's operand image
};
#define MB_NEG (-1.0e30f)
// PIECE = true: the pass holds wide-tree pieces (mode-3 blocks); the instantiation without them is the round-2 kernel unchanged —
// the extra ancestor words cost registers the 256-VGPR budget does not have (measured: 960 B/lane of scratch and 22 -> 134 us when
// both forms shared one body), so la_llama_mstep picks the instantiation per pass (bit 8 of the LA_MIN_NBLK word).
// VR = true (round 4, la_lab_set(20, 1)): the V tile of the NEXT key tile is already on its way while a tile is computed — LDS-DMA
// into a per-wave ring of two 8 KiB slots that aliases the merge buffer (no registers).  Built on the reading that the launch pays
// one exposed V round trip per tile; measured slightly SLOWER (Mistral bs=8 10.07 vs 10.00 ms, 13B bs=4 11.05 vs 10.97, Mixtral bs=4
// 22.8 vs 21.6): the launch is bound by the bytes each CU pulls (a workgroup streams its slot's whole K/V: 352 KiB at ~700 keys, at
// the HBM-class rate of one CU), not by the order in which a wave requests them.  Off by default; same values into the same MFMAs.
template <bool PIECE, bool VR>
__global__ __launch_bounds__(512) void k_tree_attn_mb(MbAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float mgbuf[];   // [4][2][66][64] merge buffer + [2][8] Q fragments (16 KiB)
    // GQA: the query heads of one kv head read the SAME K/V tiles.  Block x of a grid runs on XCD x % 8 and every XCD has its own
    // L2, so consecutive head ids would put the 4 readers of a Mistral kv head on 4 different XCDs — 4 HBM reads of every tile.
    // Head id from the block id so that the group of kv head g sits on one XCD (x % nkv = g: x, x + nkv, x + 2 nkv, ... share
    // x % 8 when nkv % 8 == 0); any nkv: still a permutation of the heads.
    const int grp_ = a.nh / a.nkv;
    const int h = ((int)blockIdx.x % a.nkv) * grp_ + (int)blockIdx.x / a.nkv, sp = blockIdx.y, blk = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tb = wave & 1, par = wave >> 1;
    const int hk = h / grp_;
    const bool g_rot = a.rot != 0 && grp_ > 1;
    const int KB = a.total_keys >> 5;
    const int* mt = a.meta + blk * LA_MB_META;
    const int slot = mt[LA_MBM_SLOT], T = mt[LA_MBM_T], nkeys = mt[LA_MBM_NKEYS], first = mt[LA_MBM_FIRST];
    const int nprev = blk - first;                     // earlier blocks of the chain (each 64 rows)
    const int row = tb * 32 + (lane & 31);
    const bool mine = row < T;
    const unsigned long long rm = mine ? a.rowmask[blk * 64 + row] : 0ull;
    // wide-tree piece: the earlier blocks of the slot are the first rows of the SAME tree, visible under this row's ancestor words
    const bool piece = PIECE && mt[LA_MBM_MODE] == LA_MODE_TREE_PIECE;
    // the ancestor words over the earlier pieces are NOT kept in registers across the tile loop (the kernel sits at the 256-VGPR
    // limit): a prior tile re-reads its word (8 bytes per lane, an L1 / L2 hit); only the window rule needs their popcount up front
    const unsigned long long* xrow = PIECE ? a.xmask + ((size_t)blk * 64 + (mine ? row : 0)) * 3 : nullptr;
    int anc_prev = nprev * 64;
    if constexpr (PIECE) {
        if (piece) {
            anc_prev = 0;
            if (a.window > 0 && mine) for (int q2 = 0; q2 < nprev && q2 < 3; ++q2) anc_prev += __popcll(xrow[q2]);
        }
    }
    const int NPall = (nkeys + 31) >> 5;
    const int qpos0 = piece ? nkeys : nkeys + nprev * 64;      // lowest position a row of the block can have
    const int ts = (a.window > 0 && qpos0 - a.window > 0) ? ((qpos0 - a.window) >> 5) : 0;
    const int tsm = ts < NPall ? ts : NPall;           // skipped main tiles
    const int NP = NPall - tsm;
    const int NF = 2 * nprev;                          // fresh tiles of earlier chain blocks
    const int NT = NP + NF + 2;
    const int tile0 = slot * a.slot_tiles + tsm;
    const int key_lo = (a.window > 0) ? nkeys + anc_prev + __popcll(rm) - 1 - a.window : -0x40000000;
    const int i0 = (NT * sp) / a.nsplit;
    const int i1 = (__ballot(mine) == 0ull) ? i0 : (NT * (sp + 1)) / a.nsplit;

    // Q fragments of both token blocks live in LDS (16 KiB behind the merge buffer), not in 32 registers per lane: wave (tb, par)
    // brings fragments 2 par and 2 par + 1 of its token block, every wave reads the 8 fragments of its block back per tile.  With
    // K double-buffered, V and the 64 accumulators the kernel otherwise sits past the 256-VGPR budget (round 3: 56 / 72 B per lane
    // of scratch); same operands in the same MFMA order, bit-identical results.
    bf16x8* const qs = (bf16x8*)(mgbuf + 4 * 2 * 66 * 64);
    {
        const bf16x8* qsrc = (const bf16x8*)(a.qf + (((size_t)blk * a.nh + h) * 2 + tb) * 4096) + lane;
        qs[(tb * 8 + 2 * par) * 64 + lane] = qsrc[(2 * par) * 64];
        qs[(tb * 8 + 2 * par + 1) * 64 + lane] = qsrc[(2 * par + 1) * 64];
    }
    const unsigned qbase = (unsigned)(tb * 8 * 64 + lane);      // fragment s of this wave's token block: qs[qbase + s * 64]
    const int hh = lane >> 5;
    f32x16 o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    float m = MB_NEG, l = 0.f;

    auto tptr = [&](const bf16_t* mainp, const bf16_t* freshp, int it) -> const bf16x8* {
        if (it < NP) return (const bf16x8*)(mainp + ((size_t)hk * KB + (a.ring ? slot * a.slot_tiles + (tsm + it) % a.slot_tiles : tile0 + it)) * 4096);
        const int j = it - NP;                          // fresh tile index: earlier blocks first, then own
        const int fb = j < NF ? first + (j >> 1) : blk;
        return (const bf16x8*)(freshp + (((size_t)fb * a.nkv + hk) * 2 + (j & 1)) * 4096);
    };
    char* const vring = (char*)mgbuf + wave * 16384;           // VR: this wave's two V slots (aliases the merge buffer until the loop ends)
    auto issue_v = [&](int it, int slot) {
        const bf16x8* vt = tptr(a.vmain, a.vfresh, it) + lane;
#pragma unroll
        for (int s = 0; s < 8; ++s)
            __builtin_amdgcn_global_load_lds((gptr_t)(vt + s * 64), (lptr_t)(vring + slot * 8192 + s * 1024), 16, 0, 0);
    };
    auto tile = [&](int it, const bf16x8 (&kf)[8], int vslot, bool more) {
        const bool own = it >= NP + NF;
        const bool prior = it >= NP && !own;
        const int kb = own ? it - NP - NF : it;
        bf16x8 vf[8];
        if constexpr (!VR) {
            const bf16x8* vt = tptr(a.vmain, a.vfresh, it);
#pragma unroll
            for (int s = 0; s < 8; ++s) vf[s] = vt[s * 64 + lane];
        }
        f32x16 sc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[i] = 0.f;
        unsigned qo = qbase;
        asm volatile("" : "+v"(qo));                        // opaque per tile: the compiler must not hoist the 8 fragment reads out of the tile loop (32 VGPRs)
#pragma unroll
        for (int s = 0; s < 8; ++s) sc = LA_MFMA(kf[s], qs[qo + s * 64], sc, 0, 0, 0);
        float mx = MB_NEG;
        unsigned long long xw_tile = 0ull;
        if constexpr (PIECE) { if (prior && piece) xw_tile = xrow[(it - NP) >> 1]; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = (i & 3) + 8 * (i >> 2) + 4 * hh;
            // bf16(x / sqrt(128)) == bf16(x * fp32(1 / sqrt(128))) for EVERY finite bf16 x (checked exhaustively over the 65536 bit
            // patterns, tests/test_oracle_llama.py::test_attention_scale_as_multiply_is_exact): one multiply instead of an IEEE division
            float v = attn_scale(sc[i], a.qk);
            bool ok;
            if (own) ok = ((rm >> (kb * 32 + kk)) & 1ull) != 0ull;
            else if (PIECE && prior && piece) {
                const int jf = it - NP;                      // fresh tile of piece jf >> 1, rows 32 * (jf & 1) ...
                ok = mine && ((xw_tile >> ((jf & 1) * 32 + kk)) & 1ull) != 0ull;
            }
            else if (prior) { const int kpos = nkeys + (it - NP) * 32 + kk; ok = mine && kpos >= key_lo; }
            else { const int kidx = (tsm + kb) * 32 + kk; ok = mine && kidx < nkeys && kidx >= key_lo; }
            v = ok ? v : MB_NEG;
            sc[i] = v;
            mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float alpha = __expf(m - mn);
        float ps = 0.f;
        bf16x8 pf[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float p = (sc[i] > -1.0e29f) ? __expf(sc[i] - mn) : 0.f;
            ps += p;
            pf[i >> 3][i & 7] = (short)f2bf(p);
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
        if constexpr (VR) {
            // this tile's V copies have landed when at most the next tile's 8 K loads + 8 V copies (issued after them) are pending;
            // the ancestor word of a wide-tree piece (PIECE) is one more, younger load — waiting for it too is harmless
            if (more) vm_wait<16>(); else vm_wait<0>();
            const bf16x8* vl = (const bf16x8*)(vring + vslot * 8192) + lane;
#pragma unroll
            for (int s = 0; s < 8; ++s) vf[s] = vl[s * 64];
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            o[db] = LA_MFMA(vf[db * 2 + 0], pf[0], o[db], 0, 0, 0);
            o[db] = LA_MFMA(vf[db * 2 + 1], pf[1], o[db], 0, 0, 0);
        }
    };
    {
        // This wave's tiles: i0 + par + 4 k, k = 0 .. cnt - 1.  GQA: the grp_ query heads of a kv head stream the SAME tiles (their
        // workgroups share an XCD, see the head map above); started together they all miss on the same lines and each pulls the
        // whole K/V at the HBM-class rate of one CU.  Query head g of the group starts its list at k = g * cnt / grp_ and wraps
        // (round 4, as k_tree_attn1): every part of the K/V is first touched by one of them and found in L2 by the others.  The
        // order of the online-softmax updates changes with it (deterministic per head; grp_ = 1: unchanged).
        const int cnt = i1 > i0 + par ? (i1 - i0 - par + 3) >> 2 : 0;
        int kx = (cnt > 0 && g_rot) ? ((h % grp_) * cnt) / grp_ : 0;
        auto next_k = [&]() { kx = kx + 1 == cnt ? 0 : kx + 1; return i0 + par + 4 * kx; };
        bf16x8 kA[8], kB[8];
        int it = i0 + par + 4 * kx;
        if (cnt > 0) {
            const bf16x8* kt = tptr(a.kmain, a.kfresh, it);
#pragma unroll
            for (int s = 0; s < 8; ++s) kA[s] = kt[s * 64 + lane];
            if constexpr (VR) issue_v(it, 0);
        }
        __syncthreads();                                    // the Q fragments are in LDS
        int left = cnt;                                     // tiles still to compute, `it` included
        while (left > 0) {
            int nx = it;
            if (left > 1) {
                nx = next_k();
                const bf16x8* kt = tptr(a.kmain, a.kfresh, nx);
#pragma unroll
                for (int s = 0; s < 8; ++s) kB[s] = kt[s * 64 + lane];
                if constexpr (VR) issue_v(nx, 1);
            }
            tile(it, kA, 0, left > 1);
            it = nx;
            if (--left == 0) break;
            if (left > 1) {
                nx = next_k();
                const bf16x8* kt = tptr(a.kmain, a.kfresh, nx);
#pragma unroll
                for (int s = 0; s < 8; ++s) kA[s] = kt[s * 64 + lane];
                if constexpr (VR) issue_v(nx, 0);
            }
            tile(it, kB, 1, left > 1);
            it = nx;
            --left;
        }
    }
    if constexpr (VR) {                                     // every wave is done with its V slots: the region becomes the merge buffer
        vm_wait<0>();
        __syncthreads();
    }
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
    float mp[4], wp[4];
    float Mx = MB_NEG, L = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) { mp[p] = mgbuf[(size_t)(((p * 2 + tb) * 66) + 64) * 64 + lane]; Mx = fmaxf(Mx, mp[p]); }
#pragma unroll
    for (int p = 0; p < 4; ++p) { wp[p] = __expf(mp[p] - Mx); L += mgbuf[(size_t)(((p * 2 + tb) * 66) + 65) * 64 + lane] * wp[p]; }
    float od[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float v = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) v += mgbuf[(size_t)(((p * 2 + tb) * 66) + par * 16 + i) * 64 + lane] * wp[p];
        od[i] = v;
    }
    if (a.attn_xp) {
        // one key split: what k_attn_combine_mb<1> would compute from the partial — bf16(o * (1 / L)), zeros for rows past T —
        // written here (8 bytes = 4 head dims per lane), one launch and the fp32 partial round trip less per layer
        const float inv = 1.0f / L;
        bf16_t* dst = a.attn_xp + (size_t)blk * 64 * a.nh * 128;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t lo = 0, hi2 = 0;
            if (row < T) {
                lo = (uint32_t)f2bf(od[4 * g] * inv) | ((uint32_t)f2bf(od[4 * g + 1] * inv) << 16);
                hi2 = (uint32_t)f2bf(od[4 * g + 2] * inv) | ((uint32_t)f2bf(od[4 * g + 3] * inv) << 16);
            }
            uint2 v; v.x = lo; v.y = hi2;
            *(uint2*)(dst + xp_offset(row, h * 128 + par * 32 + 8 * g + 4 * hh)) = v;
        }
        return;
    }
    const size_t pidx = (((size_t)blk * a.nh + h) * a.nsplit + sp) * 64 + row;
    float* op = a.opart + pidx * 128;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v = {od[4 * g], od[4 * g + 1], od[4 * g + 2], od[4 * g + 3]};
        *(f32x4*)(op + par * 32 + 8 * g + 4 * hh) = v;
    }
    if (hh == 0 && par == 0) { a.mpart[pidx] = Mx; a.lpart[pidx] = L; }
}

// merge key splits, normalise, round to bf16 and emit the packed operand of o_proj (rows past T carry zeros)
template <int NS>
__global__ __launch_bounds__(256) void k_attn_combine_mb(const float* __restrict__ opart, const float* __restrict__ mpart,
                                                          const float* __restrict__ lpart, int nh, const int* __restrict__ meta,
                                                          bf16_t* __restrict__ attn_xp) {
    const int blk = blockIdx.y;
    const int gid = blockIdx.x * 256 + threadIdx.x;   // (h, tok, d8)
    if (gid >= nh * 64 * 16) return;
    const int d8 = gid & 15, tok = (gid >> 4) & 63, h = gid >> 10;
    bf16_t* dst = attn_xp + (size_t)blk * 64 * nh * 128 + xp_offset(tok, h * 128 + d8 * 8);
    if (tok >= meta[blk * LA_MB_META + LA_MBM_T]) {
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        *(bf16x8*)dst = z;
        return;
    }
    float ms[NS], ls[NS];
    f32x4 o0[NS], o1[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const size_t pidx = (((size_t)blk * nh + h) * NS + s) * 64 + tok;
        ms[s] = mpart[pidx];
        ls[s] = lpart[pidx];
        const float* op = opart + pidx * 128 + d8 * 8;
        o0[s] = *(const f32x4*)op;
        o1[s] = *(const f32x4*)(op + 4);
    }
    float Mx = MB_NEG;
#pragma unroll
    for (int s = 0; s < NS; ++s) Mx = fmaxf(Mx, ms[s]);
    float L = 0.f, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float w = __expf(ms[s] - Mx);
        L += w * ls[s];
        acc[0] += w * o0[s][0]; acc[1] += w * o0[s][1]; acc[2] += w * o0[s][2]; acc[3] += w * o0[s][3];
        acc[4] += w * o1[s][0]; acc[5] += w * o1[s][1]; acc[6] += w * o1[s][2]; acc[7] += w * o1[s][3];
    }
    const float inv = 1.0f / L;
    bf16x8 ov;
#pragma unroll
    for (int j = 0; j < 8; ++j) ov[j] = (short)f2bf(acc[j] * inv);
    *(bf16x8*)dst = ov;
}

__global__ __launch_bounds__(256) void k_argmax_mb(const float* __restrict__ cv, const int* __restrict__ ci, int n_tiles,
                                                    int* __restrict__ out) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int t = blockIdx.x, blk = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* cvb = cv + (size_t)blk * n_tiles * 64;
    const int* cib = ci + (size_t)blk * n_tiles * 64;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int i = threadIdx.x; i < n_tiles; i += 256) {
        const float v = cvb[(size_t)i * 64 + t];
        const int idx = cib[(size_t)i * 64 + t];
        if (v > best || (v == best && idx < bidx)) { best = v; bidx = idx; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    if (lane == 0) { sv[wave] = best; si[wave] = bidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bidx)) { best = sv[w]; bidx = si[w]; }
        out[blk * 64 + t] = bidx;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Accept scan + commit plan, one wavefront per block (same walk as k_accept_scan / pretrained_model_batch.py:814-905):
// DST[row] = absolute main-cache key row every kept row goes to (root + accepted drafts, or all rows of a prefill chain),
// then the slots' cursors advance (block order -> deterministic).  One workgroup of nblk waves.
// A wide tree (block b in mode 0 followed by mode-3 blocks of the same slot) is walked by wave b alone: lane j owns rows
// j, 64 + j, 128 + j, 192 + j; a row's parent is the highest set bit below it over the concatenated ancestor words; the walk
// keeps every row whose root path spells the accepted tokens alive (the reference's surviving leaf branches,
// pretrained_model.py:831, 850-860) and continues on the lowest one.  The waves of the continuation blocks only publish
// their argmax rows.
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_accept_scan_mb(const int* __restrict__ meta, const int* __restrict__ ids,
                                 const unsigned long long* __restrict__ rowmask, const unsigned long long* __restrict__ xmask,
                                 const int* __restrict__ argmax, int nblk,
                                 int slot_keys, int ring, int* __restrict__ bstate, int* __restrict__ out) {
    __shared__ int ncommit[8];
    const int b = threadIdx.x >> 6, j = threadIdx.x & 63;
    if (b < nblk) {
        const int* mt = meta + b * LA_MB_META;
        const int slot = mt[LA_MBM_SLOT], T = mt[LA_MBM_T], mode = mt[LA_MBM_MODE];
        int limit = mt[LA_MBM_LIMIT];
        limit = limit < 1 ? 1 : (limit > LA_MOUT_TOKS ? LA_MOUT_TOKS : limit);
        const int pos0 = mt[LA_MBM_NKEYS] + mt[LA_MBM_BASE];
        auto row_of = [&](int k) { return slot * slot_keys + (ring ? (pos0 + k) % slot_keys : pos0 + k); };
        const int am = argmax[b * 64 + j];
        out[LA_MOUT_ARGMAX + b * 64 + j] = am;
        if (j == 0) out[LA_MOUT_T + b] = T;
        int nc = 0;
        if (mode == LA_MODE_TREE_PIECE) {
            // rows, DST and counters of this block belong to the tree's first wave
            if (j == 0) out[LA_MOUT_NOUT + b] = 0;
        } else if (mode == 2) {
            // forward only: the host decides the commit (la_llama_mcommit), see k_accept_scan_b.  Nothing of this block — nor of
            // the wide-tree pieces that follow it — may move: their DST records still hold an earlier step's plan.
            if (j == 0) out[LA_MOUT_NOUT + b] = 0;
            out[LA_MOUT_DST + b * 64 + j] = -1;
            for (int q = 1; q < 4 && b + q < nblk && meta[(b + q) * LA_MB_META + LA_MBM_MODE] == LA_MODE_TREE_PIECE
                            && meta[(b + q) * LA_MB_META + LA_MBM_SLOT] == slot; ++q)
                out[LA_MOUT_DST + (b + q) * 64 + j] = -1;
        } else if (mode == 1) {
            const int tok = __shfl(am, T - 1, 64);
            out[LA_MOUT_DST + b * 64 + j] = j < T ? row_of(j) : -1;
            if (j == 0) { out[LA_MOUT_OUTTOK + b * LA_MOUT_TOKS] = tok; out[LA_MOUT_NOUT + b] = 1; }
            nc = T;
        } else {
            int np = 1;                                        // pieces of the tree (wave-uniform)
            while (np < 4 && b + np < nblk && meta[(b + np) * LA_MB_META + LA_MBM_MODE] == LA_MODE_TREE_PIECE
                   && meta[(b + np) * LA_MB_META + LA_MBM_SLOT] == slot) ++np;
            int Tq[4], amq[4], idq[4], par[4], dst[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                Tq[q] = 0; amq[q] = 0; idq[q] = 0; par[q] = -1; dst[q] = -1;
                if (q < np) {
                    Tq[q] = q == 0 ? T : meta[(b + q) * LA_MB_META + LA_MBM_T];
                    amq[q] = q == 0 ? am : argmax[(b + q) * 64 + j];
                    idq[q] = ids[(b + q) * 64 + j];
                    if (j < Tq[q] && (q > 0 || j > 0)) {
                        const unsigned long long below = rowmask[(b + q) * 64 + j] & ((1ull << j) - 1ull);
                        if (below != 0ull) par[q] = 64 * q + 63 - __clzll((long long)below);
                        else {
                            const unsigned long long* xm = xmask + ((size_t)(b + q) * 64 + j) * 3;
                            for (int r = q - 1; r >= 0; --r) {
                                const unsigned long long w = xm[r];
                                if (w != 0ull) { par[q] = 64 * r + 63 - __clzll((long long)w); break; }
                            }
                        }
                    }
                }
            }
            unsigned long long live[4] = {1ull, 0ull, 0ull, 0ull};
            int cur = 0, depth = 0;
            if (j == 0) dst[0] = row_of(0);
            while (true) {
                const int cq = cur >> 6, cl = cur & 63;
                const int want = __shfl(cq == 0 ? amq[0] : cq == 1 ? amq[1] : cq == 2 ? amq[2] : amq[3], cl, 64);
                if (j == 0) out[LA_MOUT_OUTTOK + b * LA_MOUT_TOKS + depth] = want;
                if (depth + 1 >= limit) break;
                unsigned long long cand[4];
                bool any = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int p = par[q];
                    const unsigned long long lw = p < 0 ? 0ull : ((p >> 6) == 0 ? live[0] : (p >> 6) == 1 ? live[1] : (p >> 6) == 2 ? live[2] : live[3]);
                    cand[q] = q < np ? __ballot(p >= 0 && ((lw >> (p & 63)) & 1ull) != 0ull && idq[q] == want) : 0ull;
                    any = any || cand[q] != 0ull;
                }
                if (!any) break;
                int nq = 0;
                while (cand[nq] == 0ull) ++nq;
                cur = 64 * nq + __ffsll((long long)cand[nq]) - 1;
                ++depth;
#pragma unroll
                for (int q = 0; q < 4; ++q) { live[q] = cand[q]; if (cur == 64 * q + j) dst[q] = row_of(depth); }
            }
            if (j == 0) out[LA_MOUT_NOUT + b] = depth + 1;
            nc = depth + 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (q < np) out[LA_MOUT_DST + (b + q) * 64 + j] = dst[q];
        }
        if (j == 0) ncommit[b] = nc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int p = 0; p < nblk; ++p) bstate[LA_BST_NKEYS + meta[p * LA_MB_META + LA_MBM_SLOT]] += ncommit[p];
        for (int p = nblk; p < 8; ++p) out[LA_MOUT_NOUT + p] = 0;
    }
    __syncthreads();
    if (threadIdx.x < LA_MAX_SEQ) out[LA_MOUT_NKEYS + threadIdx.x] = bstate[LA_BST_NKEYS + threadIdx.x];
}

// fresh rows -> their DST rows of the main cache.  grid = (layers * kv heads, blocks), 256 threads.
__global__ __launch_bounds__(256) void k_kv_commit_mb(const bf16_t* __restrict__ kfresh, const bf16_t* __restrict__ vfresh,
                                                       bf16_t* __restrict__ kmain, bf16_t* __restrict__ vmain,
                                                       const int* __restrict__ out, int nkv, int blk_stride, int total_keys) {
    const int lh = blockIdx.x, blk = blockIdx.y;
    const int layer = lh / nkv, head = lh - layer * nkv;
    const size_t KB = (size_t)(total_keys >> 5);
    const bf16_t* kf = kfresh + (((size_t)layer * blk_stride + blk) * nkv + head) * 8192;
    const bf16_t* vf = vfresh + (((size_t)layer * blk_stride + blk) * nkv + head) * 8192;
    bf16_t* km = kmain + (size_t)lh * KB * 4096;
    bf16_t* vm = vmain + (size_t)lh * KB * 4096;
    const int* dstp = out + LA_MOUT_DST + blk * 64;
    // kept rows first (a verify block keeps <= LA_MOUT_TOKS of its 64 rows; round 2 walked all 64 x 144 items and re-read DST for
    // every one of them: 45-170 us per step at 13B bs=4): wave 0 compacts (row, dst) pairs with one ballot, everybody then copies
    // only those rows — 16-byte chunks of the K row-fragments, 2-byte elements of the transposed V fragments
    __shared__ int krow[64], kdst[64];
    __shared__ int nkeep;
    if (threadIdx.x < 64) {
        const int d = dstp[threadIdx.x];
        const bool keep = d >= 0 && d < total_keys;
        const unsigned long long m = __ballot(keep);
        if (keep) { const int k = __popcll(m & ((1ull << threadIdx.x) - 1ull)); krow[k] = threadIdx.x; kdst[k] = d; }
        if (threadIdx.x == 0) nkeep = __popcll(m);
    }
    __syncthreads();
    const int n = nkeep;
    for (int i = threadIdx.x; i < n * 16; i += 256) {
        const int k = i >> 4, p = i & 15;
        *(bf16x8*)(km + rf_offset(kdst[k], p * 8)) = *(const bf16x8*)(kf + rf_offset(krow[k], p * 8));
    }
    for (int i = threadIdx.x; i < n * 128; i += 256) {
        const int k = i >> 7, d = i & 127;
        vm[vf_offset(kdst[k], d)] = vf[vf_offset(krow[k], d)];
    }
}

// =============================================================================================================
// launchers
// =============================================================================================================
static bool g_mb_attr = false;
template <int RBV, int TW, int EPI>
static void wide_launch(dim3 grid, hipStream_t st, const MbArgs& a) {
    // schedule (k_gemm_wide SCH): buffer-addressed pieces everywhere; a fragment read after every MFMA where a wave has >= 3 token tiles
    // (6+ MFMAs per half stage to cover a read; with 1-2 tiles the reads-together form covers them better).  Mistral bs=8 9.88-9.97 ->
    // 9.70-9.72 ms, 13B bs=4 10.68-10.73 -> 10.59-10.63 ms per step (profiles/r04_wide_gemm_schedule.txt); la_lab_set(24, 0) = schedule 0
    constexpr int SCH = TW >= 3 ? 3 : 2;
    if (g_la_mb_sch) k_gemm_wide<RBV, TW, EPI, 0, SCH><<<grid, 512, WideGeom<RBV, TW>::LDS, st>>>(a);
    else k_gemm_wide<RBV, TW, EPI, 0, 0><<<grid, 512, WideGeom<RBV, TW>::LDS, st>>>(a);
}
template <typename K> static hipError_t set_lds(K k, int bytes) {
    return hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
static constexpr int mb_lds(int nt) { return (2 * 4 * nt * 1024) > 65536 ? (2 * 4 * nt * 1024) : 65536; }   // x double buffer vs reduction (64 KiB)

int lk_mb_init() {
    if (g_mb_attr) return 0;
    hipError_t e = hipSuccess;
#define SETALL(NT) \
    if (e == hipSuccess) e = set_lds(k_gemm_mb<2, NT, MB_SLAB>, mb_lds(NT)); \
    if (e == hipSuccess) e = set_lds(k_gemm_mb<4, NT, MB_SWIGLU>, mb_lds(NT)); \
    if (e == hipSuccess) e = set_lds(k_gemm_mb<2, NT, MB_QKV>, mb_lds(NT)); \
    if (e == hipSuccess) e = set_lds(k_gemm_mb<4, NT, MB_LOGITS>, mb_lds(NT));
    SETALL(2) SETALL(4) SETALL(8)
#undef SETALL
    if (e == hipSuccess) e = set_lds(k_gemm_mb<4, 4, MB_SLAB, true>, mb_lds(4));
    if (e == hipSuccess) e = set_lds(k_gemm_mb<8, 4, MB_SWIGLU, true>, mb_lds(4));
    if (e == hipSuccess) e = set_lds(k_gemm_mb<2, 4, MB_SLAB, true>, mb_lds(4));
    if (e == hipSuccess) e = set_lds(k_gemm_mb<4, 4, MB_SWIGLU, true>, mb_lds(4));
    if (e == hipSuccess) e = set_lds(k_gemm_mb<2, 4, MB_SLAB, true, 4>, mb_lds(4));
    if (e == hipSuccess) e = set_lds(k_gemm_mb<4, 4, MB_SWIGLU, true, 4>, mb_lds(4));
#define SETW4(T) \
    if (e == hipSuccess) e = set_lds(k_gemm_wide<8, T, MB_SWIGLU>, WideGeom<8, T>::LDS); \
    if (e == hipSuccess) e = set_lds(k_gemm_wide<4, T, MB_SWIGLU>, WideGeom<4, T>::LDS); \
    if (e == hipSuccess) e = set_lds(k_gemm_wide<4, T, MB_LOGITS>, WideGeom<4, T>::LDS); \
    if (e == hipSuccess) e = set_lds(k_gemm_wide<8, T, MB_SWIGLU, 0, (T >= 3 ? 3 : 2)>, WideGeom<8, T>::LDS); \
    if (e == hipSuccess) e = set_lds(k_gemm_wide<4, T, MB_SWIGLU, 0, (T >= 3 ? 3 : 2)>, WideGeom<4, T>::LDS); \
    if (e == hipSuccess) e = set_lds(k_gemm_wide<4, T, MB_LOGITS, 0, (T >= 3 ? 3 : 2)>, WideGeom<4, T>::LDS);
#define SETW2(T) \
    if (e == hipSuccess) e = set_lds(k_gemm_wide<2, T, MB_SLAB>, WideGeom<2, T>::LDS); \
    if (e == hipSuccess) e = set_lds(k_gemm_wide<2, T, MB_QKV>, WideGeom<2, T>::LDS); \
    if (e == hipSuccess) e = set_lds(k_gemm_wide<4, T, MB_SLAB>, WideGeom<4, T>::LDS); \
    if (e == hipSuccess) e = set_lds(k_gemm_wide<4, T, MB_QKV>, WideGeom<4, T>::LDS); \
    if (e == hipSuccess) e = set_lds(k_gemm_wide<2, T, MB_SLAB, 0, (T >= 3 ? 3 : 2)>, WideGeom<2, T>::LDS); \
    if (e == hipSuccess) e = set_lds(k_gemm_wide<2, T, MB_QKV, 0, (T >= 3 ? 3 : 2)>, WideGeom<2, T>::LDS); \
    if (e == hipSuccess) e = set_lds(k_gemm_wide<4, T, MB_SLAB, 0, (T >= 3 ? 3 : 2)>, WideGeom<4, T>::LDS); \
    if (e == hipSuccess) e = set_lds(k_gemm_wide<4, T, MB_QKV, 0, (T >= 3 ? 3 : 2)>, WideGeom<4, T>::LDS);
    SETW4(2) SETW4(3) SETW4(4) SETW2(1) SETW2(2)
#undef SETW4
#undef SETW2
    if (e == hipSuccess) e = set_lds(k_gemm_wide<4, 2, MB_SLAB, 4>, WideGeom<4, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_wide<4, 2, MB_QKV, 4>, WideGeom<4, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_wide<8, 2, MB_QKV>, WideGeom<8, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_wide<8, 2, MB_QKV, 0, 2>, WideGeom<8, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_wide<4, 4, MB_SWIGLU, 1>, WideGeom<4, 4>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_wide<4, 4, MB_SWIGLU, 2>, WideGeom<4, 4>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_wide<4, 4, MB_SWIGLU, 3>, WideGeom<4, 4>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_wide<4, 4, MB_SWIGLU, 4>, WideGeom<4, 4>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_wide<4, 4, MB_SWIGLU, 5>, WideGeom<4, 4>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_wide<4, 4, MB_SWIGLU, 6, 3>, WideGeom<4, 4>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<8, 2, MB_SWIGLU>, FatGeom<8, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<8, 3, MB_SWIGLU>, FatGeom<8, 3>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<8, 4, MB_SWIGLU>, FatGeom<8, 4>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<4, 2, MB_SLAB, 4, 0, 4, 2>, FatGeom<4, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fatd<4, 0>, LA_FATD_LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fatd<3, 0>, LA_FATD_LDS);
#if LA_LAB
    if (e == hipSuccess) e = set_lds(k_gemm_fat<8, 2, MB_SWIGLU, 4, 0, 4, 1>, FatGeom<8, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<8, 3, MB_SWIGLU, 4, 0, 4, 1>, FatGeom<8, 3>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<8, 4, MB_SWIGLU, 4, 0, 4, 1>, FatGeom<8, 4>::LDS);
#endif
    if (e == hipSuccess) e = set_lds(k_gemm_fat<4, 2, MB_SWIGLU, 4, 2>, FatGeom<4, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<4, 3, MB_SWIGLU, 4, 2>, FatGeom<4, 3>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<4, 4, MB_SWIGLU, 4, 2>, FatGeom<4, 4>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<4, 1, MB_SLAB>, FatGeom<4, 1>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<4, 2, MB_SLAB>, FatGeom<4, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<2, 2, MB_SLAB, 4, 2, 2>, FatGeom<2, 2, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<2, 2, MB_QKV, 4, 2, 2>, FatGeom<2, 2, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<2, 2, MB_QKV, 2, 2, 2>, FatGeom<2, 2, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<2, 2, MB_QKV, 4, 0, 2>, FatGeom<2, 2, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<2, 2, MB_QKV, 2, 0, 2>, FatGeom<2, 2, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<2, 1, MB_QKV, 4, 0, 2>, FatGeom<2, 1, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<2, 1, MB_QKV, 2, 0, 2>, FatGeom<2, 1, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<4, 1, MB_QKV, 4>, FatGeom<4, 1>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<4, 2, MB_QKV, 4>, FatGeom<4, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<4, 1, MB_QKV, 2>, FatGeom<4, 1>::LDS);
    if (e == hipSuccess) e = set_lds(k_gemm_fat<4, 2, MB_QKV, 2>, FatGeom<4, 2>::LDS);
    if (e == hipSuccess) e = set_lds(k_tree_attn_mb<false, true>, 8 * 66 * 64 * 4 + 16384);
    if (e == hipSuccess) e = set_lds(k_tree_attn_mb<false, false>, 8 * 66 * 64 * 4 + 16384);
    if (e == hipSuccess) e = set_lds(k_tree_attn_mb<true, true>, 8 * 66 * 64 * 4 + 16384);
    if (e == hipSuccess) e = set_lds(k_tree_attn_mb<true, false>, 8 * 66 * 64 * 4 + 16384);
    if (e != hipSuccess) return (int)e;
    g_mb_attr = true;
    return 0;
}

int lk_mb_build_inputs(hipStream_t st, const int* d_in, const int* d_bstate, int nblk, int* d_meta, int* d_pos,
                       uint64_t* d_rowmask, int* d_ids) {
    if (nblk < 1 || nblk > LA_MB_MAX) return -1;
    k_build_inputs_mb<<<nblk, 64, 0, st>>>(d_in, d_bstate, d_meta, d_pos, (unsigned long long*)d_rowmask, d_ids);
    LAUNCH_CHECK(); return 0;
}
int lk_mb_embed_norm(hipStream_t st, const void* embed, const int* ids, const void* nw, int hidden, float eps, void* h, void* xp,
                     int M, int cast_first) {
    if (hidden > 8192 || (hidden & 7)) return -1;
    k_row_norm_mb<0><<<M, 512, 0, st>>>((const bf16_t*)embed, ids, (bf16_t*)h, nullptr, (const bf16_t*)nw, hidden, eps, (bf16_t*)xp, M, cast_first);
    LAUNCH_CHECK(); return 0;
}
int lk_mb_resid_norm(hipStream_t st, void* h, const float* slabs, int n_slabs, int slab_rows, const void* nw, int hidden, float eps,
                     void* xp, int M, int cast_first) {
    if (hidden > 8192 || (hidden & 7)) return -1;
#define RN(NS) k_row_norm_mb<NS><<<M, 512, 0, st>>>(nullptr, nullptr, (bf16_t*)h, slabs, (const bf16_t*)nw, hidden, eps, (bf16_t*)xp, slab_rows, cast_first)
    switch (n_slabs) {
        case 1: RN(1); break; case 2: RN(2); break; case 3: RN(3); break; case 4: RN(4); break; case 6: RN(6); break; case 8: RN(8); break;
        default: return -1;
    }
#undef RN
    LAUNCH_CHECK(); return 0;
}

int lk_mb_resid_norm_router(hipStream_t st, void* h, const float* slabs, int n_slabs, int slab_rows, const void* nw, int hidden, float eps,
                            void* xp, int M, int cast_first, const void* wrouter, int n_experts, int top_k, float* route_w, const int* meta) {
    if (hidden > 8192 || (hidden & 7) || n_experts < 1 || n_experts > LA_MOE_MAX_E || top_k < 1 || top_k > n_experts) return -1;
#define RN(NS) k_row_norm_router_mb<NS><<<M, 512, 0, st>>>((bf16_t*)h, slabs, (const bf16_t*)nw, hidden, eps, (bf16_t*)xp, slab_rows, cast_first, \
                                                           (const bf16_t*)wrouter, n_experts, top_k, route_w, meta)
    switch (n_slabs) {
        case 1: RN(1); break; case 2: RN(2); break; case 4: RN(4); break; case 8: RN(8); break;
        default: return -1;
    }
#undef RN
    LAUNCH_CHECK(); return 0;
}
int lk_mb_resid_norm_addend(hipStream_t st, void* h, const void* addend, const void* nw, int hidden, float eps, void* xp, int M, int cast_first) {
    if (hidden > 8192 || (hidden & 7) || !addend) return -1;
    k_row_norm_addend_mb<<<M, 512, 0, st>>>((bf16_t*)h, (const bf16_t*)addend, (const bf16_t*)nw, hidden, eps, (bf16_t*)xp, cast_first);
    LAUNCH_CHECK(); return 0;
}
int lk_mb_moe_plan(hipStream_t st, const float* route_w, int M, int E, int* perm, int* pos, int* cnt_nb) {
    if (M < 1 || M > LA_MB_MAX * 64 || E < 1 || E > LA_MOE_MAX_E) return -1;
    k_moe_plan_mb<<<1, 512, 0, st>>>(route_w, M, E, perm, pos, cnt_nb);
    LAUNCH_CHECK(); return 0;
}
int lk_mb_moe_gather(hipStream_t st, const void* xp, const int* perm, const int* cnt_nb, int hidden, int nblk, int E, void* xg, long xg_stride) {
    if (hidden & 15) return -1;
    k_moe_gather_mb<<<dim3(E, nblk, 8), 512, 0, st>>>((const bf16_t*)xp, perm, cnt_nb, hidden, (bf16_t*)xg, xg_stride);
    LAUNCH_CHECK(); return 0;
}
int lk_mb_moe_plan_gather(hipStream_t st, const float* route_w, int M, const void* xp, int hidden, int nblk, int E, void* xg, long xg_stride,
                          int* perm, int* pos, int* cnt_nb) {
    if (M < 1 || M > LA_MB_MAX * 64 || M > 512 || E < 1 || E > LA_MOE_MAX_E || (hidden & 15)) return -1;
    k_moe_plan_gather_mb<<<dim3(E, nblk, 8), 512, 0, st>>>(route_w, M, (const bf16_t*)xp, hidden, (bf16_t*)xg, xg_stride, perm, pos, cnt_nb);
    LAUNCH_CHECK(); return 0;
}
int lk_mb_moe_accum(hipStream_t st, const float* slabs0, long slab_stride, int n_slabs, int slab_rows, const float* route_w, int E, int hidden,
                    void* acc, int M, const int* pos) {
    if (hidden & 7) return -1;
#define MA(NS) k_moe_accum_mb<NS><<<M, 256, 0, st>>>(slabs0, slab_stride, slab_rows, route_w, E, hidden, (bf16_t*)acc, pos)
    switch (n_slabs) {
        case 1: MA(1); break; case 2: MA(2); break; case 4: MA(4); break; case 8: MA(8); break;
        default: return -1;
    }
#undef MA
    LAUNCH_CHECK(); return 0;
}

int lk_mb_moe_accum_norm(hipStream_t st, const float* slabs0, long slab_stride, int n_slabs, int slab_rows, const float* route_w, int E,
                         int hidden, int M, const int* pos, void* h, const void* nw, float eps, void* xp, int cast_first) {
    if (hidden > 8192 || (hidden & 7)) return -1;
#define MA(NS) k_moe_accum_norm_mb<NS><<<M, 512, 0, st>>>(slabs0, slab_stride, slab_rows, route_w, E, hidden, pos, (bf16_t*)h, \
                                                           (const bf16_t*)nw, eps, (bf16_t*)xp, cast_first)
    switch (n_slabs) {
        case 1: MA(1); break; case 2: MA(2); break; case 4: MA(4); break; case 8: MA(8); break;
        default: return -1;
    }
#undef MA
    LAUNCH_CHECK(); return 0;
}

int lk_mb_cand_slots(int n_wg) { return n_wg * 4; }

template <int RBV, int EPI>
static int launch_mb(hipStream_t st, const MbArgs& a, int n_wg, int ksplit, int nblk) {
    // mb_k_range cuts an even K16 on even k-tiles: with fewer than two k-tiles per split a split would be EMPTY (t0 == t1), which only the
    // fat launches guard against — k_gemm_wide's DMA clamp (t1 - 1) would read below its range.  No engine shape gets here (resolve_cfg keeps
    // >= 4 k-tiles per split); refuse instead of computing garbage.
    if (ksplit < 1 || (ksplit > 1 && a.K16 < 2 * ksplit)) return -1;
    if (a.nblk_dev) {
        // gathered expert: the block count is a device value (typically 1-2 of the step's nblk): passes of two blocks, a pass
        // past the expert's count returns before it touches the weights
        if constexpr (EPI == MB_SLAB || EPI == MB_SWIGLU) {
            if (a.ex_n > 1 && (g_la_ex_d4 & (RBV == 4 ? 4 : 8)) && n_wg % 2 == 0 && (EPI == MB_SLAB || (a.planned && !a.gu_interleaved))) {
                // round 5 (la_lab_set key 25, bit 2 = gate/up, bit 3 = down): TWO adjacent weight regions per workgroup.  Every workgroup of an
                // expert stages the expert's whole x through LDS: 0.5-1.8 MB from L2 per 0.9 MB of weights from HBM, and the per-CU times of
                // the two ADD (T = W / 25 GB/s + x / 130 GB/s) — the gap between these launches (4.6-5.2 TB/s of weights) and the
                // dense 64-row kernels (5.9).  With two regions per workgroup the x traffic and the LDS reads per weight byte halve: the 8 waves
                // are 2 RBV row-blocks x 4 / RBV K parts (gate/up: one row-block per wave over the whole K range; down: 2 K parts instead of 4).
                // FEWER K parts = ANOTHER fp32 summation order than the one-region launch: results agree to tolerance, not bitwise
                // (include/lookahead_hip_lab.h key 25; tests/test_gpu_mblock.py::test_merged_expert_launch_forms_agree asserts 1e-2) —
                // so Mixtral's default numerics moved by that much when this form became the default in round 5.
                MbArgs p = a;
                if (p.planned) {
                    for (int i = 0; i < 4; ++i) { p.boff[4 + i] = a.wg_chunks + a.boff[i]; p.nv[4 + i] = a.nv[i]; p.nvl[4 + i] = a.nvl[i]; }
                    p.wg_chunks = 2 * a.wg_chunks;
                }
                k_gemm_mb<2 * RBV, 4, EPI, true><<<dim3(n_wg / 2, ksplit, (nblk + 1) / 2 * a.ex_n), 512, mb_lds(4), st>>>(p);
                LAUNCH_CHECK(); return 0;
            }
            if (a.ex_n > 1) {
                // two workgroups per CU (D = 4, <= 128 VGPRs): la_lab_set key 25 bit 0 = the gate/up launch (default on: Mixtral bs=4
                // 21.0-21.5 -> 20.1-20.2 ms per step), bit 1 = the down launch (RBV = 2: 44 B per lane of scratch at that bound)
                if (g_la_ex_d4 & (RBV == 4 ? 1 : 2)) k_gemm_mb<RBV, 4, EPI, true, 4><<<dim3(n_wg, ksplit, (nblk + 1) / 2 * a.ex_n), 512, mb_lds(4), st>>>(a);
                else k_gemm_mb<RBV, 4, EPI, true><<<dim3(n_wg, ksplit, (nblk + 1) / 2 * a.ex_n), 512, mb_lds(4), st>>>(a);
                LAUNCH_CHECK(); return 0;
            }
        }
        k_gemm_mb<RBV, 4, EPI><<<dim3(n_wg, ksplit, (nblk + 1) / 2), 512, mb_lds(4), st>>>(a);
        LAUNCH_CHECK(); return 0;
    }
    // nblk >= 3: every token block in ONE weight pass (k_gemm_wide): RBV = 4 -> 4 TW token blocks per workgroup, RBV = 2 -> 8 TW
    if (nblk >= 3 && !g_la_mb_narrow) {
        const dim3 grid(n_wg, ksplit, 1);
#if LA_LAB
        if constexpr (RBV == 4 && EPI == MB_SWIGLU) {           // measurement builds of the 512-row gate/up launch (la_debug_set key 4)
            if (g_la_mb_dbg == 6 && nblk >= 7) {
                MbArgs p = a; p.dbg_times = g_la_dbg_times;
                k_gemm_wide<4, 4, MB_SWIGLU, 6, 3><<<grid, 512, WideGeom<4, 4>::LDS, st>>>(p);
                LAUNCH_CHECK(); return 0;
            }
            if (g_la_mb_dbg && nblk >= 7) {
                if (g_la_mb_dbg == 1) k_gemm_wide<4, 4, MB_SWIGLU, 1><<<grid, 512, WideGeom<4, 4>::LDS, st>>>(a);
                else if (g_la_mb_dbg == 2) k_gemm_wide<4, 4, MB_SWIGLU, 2><<<grid, 512, WideGeom<4, 4>::LDS, st>>>(a);
                else if (g_la_mb_dbg == 3) k_gemm_wide<4, 4, MB_SWIGLU, 3><<<grid, 512, WideGeom<4, 4>::LDS, st>>>(a);
                else if (g_la_mb_dbg == 4) k_gemm_wide<4, 4, MB_SWIGLU, 4><<<grid, 512, WideGeom<4, 4>::LDS, st>>>(a);
                else k_gemm_wide<4, 4, MB_SWIGLU, 5><<<grid, 512, WideGeom<4, 4>::LDS, st>>>(a);
                LAUNCH_CHECK(); return 0;
            }
        }
#endif
        if constexpr (RBV == 2) {
            // Paired form (MB_SLAB, MB_QKV): TWO adjacent weight regions per workgroup and HALF the token blocks — the RBV = 4 wave
            // grid (2 row groups x 4 token groups) over regions {2 x, 2 x + 1}, grid.z = 2 token halves.  Every workgroup re-reads the
            // whole x operand of its token blocks from L2 (1 GB per launch at 512 rows and 256 workgroups: the dominant term of these
            // launches); pairing halves that, and the second reader of a weight region (z = 1, 128 workgroup ids later: same XCD)
            // finds it in L2.  Same MFMAs on the same operands in the same order per output: bit-identical results.
            // Quad form (MB_QKV of a GQA model at >= 7 blocks: small N, many rows — the x re-reads dominate): FOUR adjacent regions and
            // a QUARTER of the token blocks per workgroup.  L2 -> LDS traffic of the launch = x bytes x (workgroup columns) + W bytes x
            // (token groups): Mistral QKV at 512 rows 512 + 100 MB paired -> 256 + 200 MB quad; not taken where W dominates
            // (MHA / 13B shapes, <= 4 blocks).  Same MFMA chain per output element: bit-identical to the paired and unpaired forms.
            if constexpr (EPI == MB_QKV) {
                const double xb = (double)nblk * 64 * a.K16 * 32, wb = (double)a.N * a.K16 * 32;       // bytes of x and of W
                const double paired = xb * (n_wg / 2) + wb * ((nblk + 3) / 4), quad = xb * (n_wg / 4) + wb * ((nblk + 1) / 2);
                if ((g_la_mb_pair & 4) && a.planned && nblk >= 7 && n_wg % 4 == 0 && (n_wg / 4 * ksplit) % 8 == 0 && quad < 0.85 * paired) {
                    MbArgs p = a;
                    p.w_keep = 1;
                    for (int r = 1; r < 4; ++r)
                        for (int j = 0; j < 2; ++j) {
                            p.boff[2 * r + j] = r * a.wg_chunks + a.boff[j]; p.nv[2 * r + j] = a.nv[j]; p.nvl[2 * r + j] = a.nvl[j];
                        }
                    p.wg_chunks = 4 * a.wg_chunks;
                    wide_launch<8, 2, EPI>(dim3(n_wg / 4, ksplit, (nblk + 1) / 2), st, p);
                    LAUNCH_CHECK(); return 0;
                }
            }
            if constexpr (EPI == MB_QKV) {
                // round 5 (bit 8 of key 6): ONE {lo, hi} region x 256 rows per workgroup as four fat waves of 2 row-blocks x 2 token blocks.
                // The paired quarters give a CU TWO regions' weights for 128 rows, and the token groups run in lockstep, so every CU waits for
                // its weights at the HBM-class rate: 1.05 MB per CU at the Mistral shape (512 rows) = the 42 of the launch's 45 us.  One
                // region x 256 rows: half the weight bytes per CU, twice the x (which comes from L2).  Taken at <= 4 blocks, and at 5-8
                // blocks when the two token halves still fit one wave of workgroups (the fuller QKV image of a GQA model: 96 x 2).
                // bit 12 of key 6: the same one-region form with ONE token tile per wave (128 rows per workgroup, grid.z = pairs of blocks) where 256
                // rows per workgroup leave most of the chip idle — the fuller GQA image (Mistral / Mixtral: 96 regions) at 3-4 blocks is 96
                // workgroups on 256 CUs, each waiting for 0.5 MB of weights at the HBM-class per-CU rate; two token groups = 192 workgroups
                // (the second reader of a region finds it in L2).  <= 2 blocks: the 2 x 2 form would run MFMAs on absent blocks.
                if ((g_la_mb_pair & 4096) != 0 && (g_la_mb_pair & 256) != 0 && ksplit == 1 && (a.K16 & 1) == 0 && (a.R & 1) == 0 && g_la_mb_dbg == 0 && nblk <= 4 && (nblk <= 2 || n_wg <= 128)) {
                    const dim3 gq(n_wg, 1, (nblk + 1) / 2);
                    if ((a.R & 3) == 0) k_gemm_fat<2, 1, MB_QKV, 4, 0, 2><<<gq, 256, FatGeom<2, 1, 2>::LDS, st>>>(a);
                    else k_gemm_fat<2, 1, MB_QKV, 2, 0, 2><<<gq, 256, FatGeom<2, 1, 2>::LDS, st>>>(a);
                    LAUNCH_CHECK(); return 0;
                }
                const int zq = (nblk + 3) / 4;
                // Measured (profiles/r05_fat_waves.txt, call 7): 13B bs=4 10.64 -> 10.59 ms, Mixtral bs=4 19.65 -> 19.41 at 4 blocks; Mistral bs=8
                // (8 blocks, two token halves) 9.465 -> 9.489: taken at <= 4 blocks only (bit 9 forces the 5-8 block form: measurement).
                if ((g_la_mb_pair & 256) && ksplit == 1 && (a.K16 & 1) == 0 && (a.R & 1) == 0 && g_la_mb_dbg == 0 && (zq == 1 || ((g_la_mb_pair & 512) && n_wg * zq <= 256))) {
                    const dim3 gq(n_wg, 1, zq);
                    if (zq == 1) {
                        if ((a.R & 3) == 0) k_gemm_fat<2, 2, MB_QKV, 4, 2, 2><<<gq, 256, FatGeom<2, 2, 2>::LDS, st>>>(a);
                        else k_gemm_fat<2, 2, MB_QKV, 2, 2, 2><<<gq, 256, FatGeom<2, 2, 2>::LDS, st>>>(a);
                    } else {
                        if ((a.R & 3) == 0) k_gemm_fat<2, 2, MB_QKV, 4, 0, 2><<<gq, 256, FatGeom<2, 2, 2>::LDS, st>>>(a);
                        else k_gemm_fat<2, 2, MB_QKV, 2, 0, 2><<<gq, 256, FatGeom<2, 2, 2>::LDS, st>>>(a);
                    }
                    LAUNCH_CHECK(); return 0;
                }
            }
            if constexpr (EPI == MB_SLAB) {
                // round 5 (bit 10 of key 6): the slab launches (o_proj / down) at <= 4 blocks as ONE 64-row weight region x 256 rows per workgroup
                // (four fat waves of 2 row-blocks x 2 token blocks) instead of two regions x 128 rows: half the weight bytes per CU at the
                // HBM-class rate, twice the x from L2 — the trade that paid for gate/up and QKV at 256 rows.  Measured SLOWER here (13B bs=4 10.50 ->
                // 10.63 ms, Mixtral bs=4 19.15 -> 19.45, profiles/r05_fat_waves.txt call 8): opt-in, bit-identical.
                if ((g_la_mb_pair & 1024) && nblk <= 4 && (a.K16 & 1) == 0 && a.K16 >= 2 * ksplit && g_la_mb_dbg == 0 && !a.planned) {
                    k_gemm_fat<2, 2, MB_SLAB, 4, 2, 2><<<dim3(n_wg, ksplit, 1), 256, FatGeom<2, 2, 2>::LDS, st>>>(a);
                    LAUNCH_CHECK(); return 0;
                }
            }
            if ((g_la_mb_pair & 1) && n_wg % 2 == 0 && (n_wg / 2 * ksplit) % 8 == 0) {
                MbArgs p = a;
                p.w_keep = 1;
                if (p.planned) {
                    p.boff[2] = a.wg_chunks + a.boff[0]; p.boff[3] = a.wg_chunks + a.boff[1];
                    p.nv[2] = a.nv[0]; p.nv[3] = a.nv[1]; p.nvl[2] = a.nvl[0]; p.nvl[3] = a.nvl[1];
                    p.wg_chunks = 2 * a.wg_chunks;
                }
                // la_debug_set key 12 (slab launches with <= 2 K splits): more token groups instead — 2 blocks per workgroup at every block count
                // QKV over at most 128 (fuller) workgroups — the multi-block image of a GQA model, cfg.qkv_mb_wg — takes token QUARTERS at
                // every block count: n_wg / 2 x 4 <= 256 workgroups of 4 row-blocks x 4 token tiles (la_debug_set(6, 9) forces the form)
                // round 5 (bit 5 of key 6): the paired slab / QKV launches as four fat waves of 4 row-blocks x TW token blocks (k_gemm_fat);
                // every K split must hold an even number of k-tiles
                const bool fat = (g_la_mb_pair & 32) && (a.K16 & 1) == 0 && a.K16 >= 2 * ksplit && g_la_mb_dbg == 0;
                const bool quarters = nblk <= 4 || (EPI == MB_SLAB && g_la_mb_ks2 && ksplit <= 2) || (EPI == MB_QKV && (n_wg <= 128 || (g_la_mb_pair & 8)));
                if constexpr (EPI == MB_SLAB) {
                    if (fat) {
                        const dim3 gs(n_wg / 2, ksplit, quarters ? (nblk + 1) / 2 : (nblk + 3) / 4);
                        // bit 13 of key 6: one K split per XCD (k_gemm_fat, MbArgs.xcd_map) — 2 / 4 / 8 splits over a grid of a multiple of 8 workgroups
                        // (bit 14: also on grids that are not whole multiples of 256 workgroups — the forced-4-split 13B launch over 160 workgroups that measured
                        //  30 % slower with the mapping, profiles/r05_fat_waves.txt call 14: measurement only)
                        const unsigned total = gs.x * gs.y * gs.z;
                        p.xcd_map = ((g_la_mb_pair & 8192) && (ksplit == 2 || ksplit == 4 || ksplit == 8) && total % 8 == 0 && (total % 256 == 0 || (g_la_mb_pair & 16384))) ? 1 : 0;
                        // round 6 (late; knob 36, default on): the two-token-tile form with x DIRECT (k_gemm_fat, STG = 2): 43.1 -> 41.2 us per launch at the
                        // Mistral shape, 512 rows.  The other one-row-group forms (quarters, QKV, one-region gate/up) measured level or slower with it.
                        if (g_la_fatx && !quarters) k_gemm_fat<4, 2, EPI, 4, 0, 4, 2><<<gs, 256, FatGeom<4, 2>::LDS, st>>>(p);
                        else if (quarters) k_gemm_fat<4, 1, EPI><<<gs, 256, FatGeom<4, 1>::LDS, st>>>(p);
                        else k_gemm_fat<4, 2, EPI><<<gs, 256, FatGeom<4, 2>::LDS, st>>>(p);
                        LAUNCH_CHECK(); return 0;
                    }
                }
                if constexpr (EPI == MB_QKV) {
                    if (fat && (a.R & 1) == 0) {              // RoPE pairs per workgroup: groups of 4 (7B, Mistral, Mixtral) or 2 (13B) rows per store
                        const dim3 gq(n_wg / 2, ksplit, quarters ? (nblk + 1) / 2 : (nblk + 3) / 4);
                        if ((a.R & 3) == 0) {
                            if (quarters) k_gemm_fat<4, 1, EPI, 4><<<gq, 256, FatGeom<4, 1>::LDS, st>>>(p);
                            else k_gemm_fat<4, 2, EPI, 4><<<gq, 256, FatGeom<4, 2>::LDS, st>>>(p);
                        } else {
                            if (quarters) k_gemm_fat<4, 1, EPI, 2><<<gq, 256, FatGeom<4, 1>::LDS, st>>>(p);
                            else k_gemm_fat<4, 2, EPI, 2><<<gq, 256, FatGeom<4, 2>::LDS, st>>>(p);
                        }
                        LAUNCH_CHECK(); return 0;
                    }
                }
                if (quarters) wide_launch<4, 1, EPI>(dim3(n_wg / 2, ksplit, (nblk + 1) / 2), st, p);
#if LA_LAB
                else if (g_la_mb_dbg == 4) k_gemm_wide<4, 2, EPI, 4><<<dim3(n_wg / 2, ksplit, (nblk + 3) / 4), 512, WideGeom<4, 2>::LDS, st>>>(p);   // measurement: no epilogue
#endif
                else wide_launch<4, 2, EPI>(dim3(n_wg / 2, ksplit, (nblk + 3) / 4), st, p);
                LAUNCH_CHECK(); return 0;
            }
        }
        if constexpr (RBV == 4 && EPI == MB_SWIGLU) {
            // round 5 (bit 6 of key 6): ONE planned region x ALL token blocks per workgroup as four fat waves (4 row-blocks x TW token blocks each,
            // 1 x 4 wave grid).  The paired fat form makes every CU pull TWO regions' weights, and since the two token halves run in lockstep
            // both wait for the same HBM miss: 1.83 MB per CU at the HBM-class per-CU rate (~25 GB/s) = 73 us of the 103-113 us launch at the
            // Mistral shape, 512 rows.  One region per workgroup: 0.92 MB of weights (nt) + 4 MB of x from L2 (~130 GB/s) per CU.
            // Measured (profiles/r05_fat_waves.txt, call 6): it wins at <= 4 blocks (13B 256 rows 90.3 -> 84.6 us, 13B bs=4 10.77 -> 10.63 ms per
            // step) and LOSES at 512 rows (Mistral 115.8 -> 121.2 us: 4 MB of x per CU instead of 2) — so it is taken at <= 4 blocks only;
            // bit 7 forces it at every block count (measurement).
            if ((g_la_mb_pair & 64) && a.planned && !a.gu_interleaved && ksplit == 1 && (a.K16 & 1) == 0 && (nblk <= 4 || ((g_la_mb_pair & 128) && nblk <= 8))) {
                const dim3 g1(n_wg, 1, 1);
                switch ((nblk + 1) / 2) {
                    case 2: k_gemm_fat<4, 2, MB_SWIGLU, 4, 2><<<g1, 256, FatGeom<4, 2>::LDS, st>>>(a); break;
                    case 3: k_gemm_fat<4, 3, MB_SWIGLU, 4, 2><<<g1, 256, FatGeom<4, 3>::LDS, st>>>(a); break;
                    default: k_gemm_fat<4, 4, MB_SWIGLU, 4, 2><<<g1, 256, FatGeom<4, 4>::LDS, st>>>(a); break;
                }
                LAUNCH_CHECK(); return 0;
            }
            // paired gate/up launch (planned images): regions {2 x, 2 x + 1} = row-blocks 0..7, half the token blocks per workgroup.
            // Bit-identical, but NOT faster (this launch is bound by its MFMA / ds_read side, not by the x traffic: 512 rows 116.5 vs
            // 116.4 us, 256 rows 68.4 vs 76.6 us at the 7B shape, profiles/r02b_mblock_paired_ab.txt): opt-in (la_debug_set(6, 3)).
            if ((g_la_mb_pair & (2 | 16)) && a.planned && !a.gu_interleaved && n_wg % 16 == 0 && ksplit == 1) {
                const bool fat = (g_la_mb_pair & 16) && (a.K16 % 2) == 0;
                MbArgs p = a;
                p.w_keep = 1;
                for (int i = 0; i < 4; ++i) { p.boff[4 + i] = a.wg_chunks + a.boff[i]; p.nv[4 + i] = a.nv[i]; p.nvl[4 + i] = a.nvl[i]; }
                p.wg_chunks = 2 * a.wg_chunks;
                const dim3 g2(n_wg / 2, 1, 2);
                if (fat) {
                    // round 5: the same pair of regions as FOUR fat waves (4 x TW accumulator tiles each, one wave per SIMD): k_gemm_fat
#if LA_LAB
                    if (g_la_mb_pair & 2048) {                  // bit 11: the register-staged form (k_gemm_fat, STG = 1)
                        switch ((nblk + 1) / 2) {
                            case 2: k_gemm_fat<8, 2, MB_SWIGLU, 4, 0, 4, 1><<<g2, 256, FatGeom<8, 2>::LDS, st>>>(p); break;
                            case 3: k_gemm_fat<8, 3, MB_SWIGLU, 4, 0, 4, 1><<<g2, 256, FatGeom<8, 3>::LDS, st>>>(p); break;
                            default: k_gemm_fat<8, 4, MB_SWIGLU, 4, 0, 4, 1><<<g2, 256, FatGeom<8, 4>::LDS, st>>>(p); break;
                        }
                        LAUNCH_CHECK(); return 0;
                    }
#endif
                    // lab knob 35 (round 6, late): weights straight into MFMA operand registers (k_gemm_fatd), 7-8 blocks, K16 % 8 == 0
                    // round 6 (late; knob 35, default on): weights straight into MFMA operand registers (k_gemm_fatd) at 5-8 blocks (an even number of
                    // k-tiles).  Measured (profiles/r06_gateup_direct_weights.txt): 512 rows 115.4 -> 109.1 us per launch at the Mistral
                    // shape, Mistral bs=8 9.55 -> 9.36 ms per step, Llama-2-7B bs=8 9.94 -> 9.76; bit-identical.
                    if (g_la_fatd && (nblk + 1) / 2 >= 3 && (a.K16 & 1) == 0 && a.R <= 64) {
                        if ((nblk + 1) / 2 == 4) k_gemm_fatd<4, 0><<<g2, 256, LA_FATD_LDS, st>>>(p);
                        else k_gemm_fatd<3, 0><<<g2, 256, LA_FATD_LDS, st>>>(p);
                        LAUNCH_CHECK(); return 0;
                    }
                    switch ((nblk + 1) / 2) {
                        case 2: k_gemm_fat<8, 2, MB_SWIGLU><<<g2, 256, FatGeom<8, 2>::LDS, st>>>(p); break;
                        case 3: k_gemm_fat<8, 3, MB_SWIGLU><<<g2, 256, FatGeom<8, 3>::LDS, st>>>(p); break;
                        default: k_gemm_fat<8, 4, MB_SWIGLU><<<g2, 256, FatGeom<8, 4>::LDS, st>>>(p); break;
                    }
                    LAUNCH_CHECK(); return 0;
                }
                switch ((nblk + 1) / 2) {                       // 64-row blocks per workgroup = TW
                    case 2: wide_launch<8, 2, EPI>(g2, st, p); break;
                    case 3: wide_launch<8, 3, EPI>(g2, st, p); break;
                    default: wide_launch<8, 4, EPI>(g2, st, p); break;
                }
                LAUNCH_CHECK(); return 0;
            }
        }
        if constexpr (RBV == 4) {
            switch ((nblk + 1) / 2) {
                case 2: wide_launch<4, 2, EPI>(grid, st, a); break;
                case 3: wide_launch<4, 3, EPI>(grid, st, a); break;
                default: wide_launch<4, 4, EPI>(grid, st, a); break;
            }
        } else {
            if (nblk <= 4) wide_launch<2, 1, EPI>(dim3(n_wg, ksplit, (nblk + 3) / 4), st, a);
            else wide_launch<2, 2, EPI>(grid, st, a);
        }
        LAUNCH_CHECK(); return 0;
    }
    // token blocks per pass: the smallest template that covers the step in <= 2 passes
    if (nblk <= 1) { k_gemm_mb<RBV, 2, EPI><<<dim3(n_wg, ksplit, 1), 512, mb_lds(2), st>>>(a); }
    else if (nblk == 2) { k_gemm_mb<RBV, 4, EPI><<<dim3(n_wg, ksplit, 1), 512, mb_lds(4), st>>>(a); }
    else { k_gemm_mb<RBV, 8, EPI><<<dim3(n_wg, ksplit, (nblk + 3) / 4), 512, mb_lds(8), st>>>(a); }
    LAUNCH_CHECK(); return 0;
}

static void fill_plan(MbArgs& a, int R, int bpm, int mats, int K16) {
    for (int m = 0; m < mats; ++m)
        for (int b = 0; b < bpm; ++b) { int v = R - 32 * b; if (v > 32) v = 32; a.nv[m * bpm + b] = v; }
    int off = 0;
    for (int i = 0; i < mats * bpm; ++i) { a.nvl[i] = (a.nv[i] + 3) & ~3; a.boff[i] = off; off += a.nvl[i] * 2 * K16; }
    a.wg_chunks = off;
}

int lk_mb_gemm(hipStream_t st, int kind, const MbGemm& g) {
    if (lk_mb_init() != 0) return -1;
    if (g.nblk < 1 || g.nblk > LA_MB_MAX) return -1;
    MbArgs a{};
    a.wp = (const bf16_t*)g.wp; a.xp = (const bf16_t*)g.xp; a.K16 = g.K / 16; a.N = g.N; a.M = g.slab_rows; a.nblk = g.nblk;
    a.route_col = g.route_col; a.route_rows = g.nblk * 64; a.nblk_dev = g.nblk_dev;
    a.ex_n = g.ex_n; a.ex_w_stride = g.ex_w_stride; a.ex_x_stride = g.ex_x_stride; a.ex_o_stride = g.ex_o_stride;
    a.ex_pad = (g_la_ex_split >> 1) & 1;
    if (g.ex_n > 1 && (!g.nblk_dev || (kind != 0 && kind != 1))) return -1;
    if (kind == 0) {
        if (g.N % 64) return -1;
        a.planned = 0; a.slabs = g.slabs;
        return launch_mb<2, MB_SLAB>(st, a, g.N / 64, g.ksplit, g.nblk);
    }
    a.planned = g.n_wg > 0 ? 1 : 0;
    for (int i = 0; i < 4; ++i) { a.nv[i] = 32; a.nvl[i] = 32; }
    if (kind == 1) {
        a.act_xp = (bf16_t*)g.act_xp;
        if (!a.planned) {                          // classic interleaved gate/up image: 4 row-blocks {G,U,G,U} per workgroup
            if (g.N % 64) return -1;
            a.gu_interleaved = 1; a.R = 64;
            return launch_mb<4, MB_SWIGLU>(st, a, g.N / 64, 1, g.nblk);
        }
        a.R = g.N / g.n_wg; if (g.N % g.n_wg || a.R > 64 || a.R <= 32) return -1;
        fill_plan(a, a.R, 2, 2, a.K16);
        return launch_mb<4, MB_SWIGLU>(st, a, g.n_wg, 1, g.nblk);
    }
    if (kind == 2) {
        a.pos = g.pos; a.rcos = (const bf16_t*)g.rcos; a.rsin = (const bf16_t*)g.rsin;
        a.qf = (bf16_t*)g.qf; a.kfresh = (bf16_t*)g.kfresh; a.vfresh = (bf16_t*)g.vfresh; a.nh = g.nh; a.nkv = g.nkv;
        if (!a.planned) {                          // classic row-permuted image (la_qkv_row_perm): pair index = 32 * workgroup + f
            a.R = 32;
            return launch_mb<2, MB_QKV>(st, a, (g.nh + 2 * g.nkv) * 2, 1, g.nblk);
        }
        const int pairs = (g.nh + 2 * g.nkv) * 64;
        a.R = pairs / g.n_wg; if (pairs % g.n_wg || a.R > 32) return -1;
        a.nv[0] = a.nv[1] = a.R; a.nvl[0] = a.nvl[1] = (a.R + 3) & ~3;
        a.boff[0] = 0; a.boff[1] = a.nvl[0] * 2 * a.K16; a.wg_chunks = 2 * a.boff[1];
        return launch_mb<2, MB_QKV>(st, a, g.n_wg, 1, g.nblk);
    }
    if (kind == 3) {
        a.logits = (bf16_t*)g.logits; a.cand_val = g.cand_val; a.cand_idx = g.cand_idx;
        if (!a.planned) {
            if (g.N % 128) return -1;
            a.R = 128;
            return launch_mb<4, MB_LOGITS>(st, a, g.N / 128, 1, g.nblk);
        }
        a.R = g.N / g.n_wg; if (g.N % g.n_wg || a.R > 128 || a.R <= 96) return -1;
        fill_plan(a, a.R, 4, 1, a.K16);
        return launch_mb<4, MB_LOGITS>(st, a, g.n_wg, 1, g.nblk);
    }
    return -1;
}

// workgroups of the lm_head launch (argmax candidate slots = 4 per workgroup and block)
int lk_mb_logits_wgs(int V, int n_wg) { return n_wg > 0 ? n_wg : V / 128; }

int lk_mb_argmax(hipStream_t st, const float* cv, const int* ci, int n_tiles, int nblk, int* out_rows) {
    k_argmax_mb<<<dim3(64, nblk), 256, 0, st>>>(cv, ci, n_tiles, out_rows);
    LAUNCH_CHECK(); return 0;
}

int lk_mb_tree_attn(hipStream_t st, const void* qf, const void* kmain, const void* vmain, const void* kfresh, const void* vfresh,
                    const uint64_t* rowmask, const int* meta, int nblk, int nh, int nkv, int slot_keys, int n_slots, int nsplit,
                    float* opart, float* mpart, float* lpart, void* attn_xp, int window, int ring, const uint64_t* xmask, int head_dim) {
    if (lk_mb_init() != 0 || nblk < 1 || nblk > LA_MB_MAX || (slot_keys & 31)) return -1;
    MbAttnArgs a{};
    a.qk = la_qk_scale(head_dim);
    a.xmask = (const unsigned long long*)xmask;
    a.qf = (const bf16_t*)qf; a.kmain = (const bf16_t*)kmain; a.vmain = (const bf16_t*)vmain;
    a.kfresh = (const bf16_t*)kfresh; a.vfresh = (const bf16_t*)vfresh;
    a.rowmask = (const unsigned long long*)rowmask; a.meta = meta;
    a.nh = nh; a.nkv = nkv; a.total_keys = slot_keys * n_slots; a.slot_tiles = slot_keys >> 5; a.nsplit = nsplit; a.window = window; a.ring = ring; a.rot = g_la_mb_attn_rot;
    a.opart = opart; a.mpart = mpart; a.lpart = lpart;
    a.attn_xp = nsplit == 1 ? (bf16_t*)attn_xp : nullptr;
    const size_t lds = 8 * 66 * 64 * sizeof(float) + 16384;
    if (g_la_mb_attn_vring) {
        if (xmask) k_tree_attn_mb<true, true><<<dim3(nh, nsplit, nblk), 512, lds, st>>>(a);
        else k_tree_attn_mb<false, true><<<dim3(nh, nsplit, nblk), 512, lds, st>>>(a);
    } else {
        if (xmask) k_tree_attn_mb<true, false><<<dim3(nh, nsplit, nblk), 512, lds, st>>>(a);
        else k_tree_attn_mb<false, false><<<dim3(nh, nsplit, nblk), 512, lds, st>>>(a);
    }
    LAUNCH_CHECK();
    if (nsplit == 1) return 0;
    const int total = nh * 64 * 16;
#define AC(NS) k_attn_combine_mb<NS><<<dim3((total + 255) / 256, nblk), 256, 0, st>>>(opart, mpart, lpart, nh, meta, (bf16_t*)attn_xp)
    switch (nsplit) {
        case 1: AC(1); break; case 2: AC(2); break; case 4: AC(4); break; case 8: AC(8); break;
        default: return -1;
    }
#undef AC
    LAUNCH_CHECK(); return 0;
}

int lk_mb_accept_scan(hipStream_t st, const int* meta, const int* ids, const uint64_t* rowmask, const uint64_t* xmask, const int* argmax, int nblk,
                      int slot_keys, int ring, int* bstate, int* d_out) {
    if (nblk < 1 || nblk > LA_MB_MAX) return -1;
    k_accept_scan_mb<<<1, 512, 0, st>>>(meta, ids, (const unsigned long long*)rowmask, (const unsigned long long*)xmask, argmax, nblk, slot_keys, ring, bstate, d_out);
    LAUNCH_CHECK(); return 0;
}
int lk_mb_kv_commit(hipStream_t st, const void* kfresh, const void* vfresh, void* kmain, void* vmain, const int* d_out, int nblk,
                    int n_layers, int nkv, int total_keys) {
    k_kv_commit_mb<<<dim3(n_layers * nkv, nblk), 256, 0, st>>>((const bf16_t*)kfresh, (const bf16_t*)vfresh, (bf16_t*)kmain,
                                                              (bf16_t*)vmain, d_out, nkv, LA_MB_MAX, total_keys);
    LAUNCH_CHECK(); return 0;
}
