// la_common.h — shared device helpers and the HBM data layouts of liblookahead_hip (gfx950 only).
//
// Layouts (all bf16 unless stated; "tile" = 512 elements = 1 KiB = one wave64 x 16 B load):
//
//  WP  packed weight  W[N][K]   : tile (nb=n/32, kb=k/16) at ((nb*(K/16))+kb)*512,
//                                 inside: lane = (n%32) + 32*((k%16)/8), e = k%8   -> lane*8+e
//      = the A-operand fragment of v_mfma_f32_32x32x16_bf16, so a wave streams a 32-row block of W
//      as consecutive, perfectly coalesced 1 KiB loads straight into MFMA operand registers.
//  XP  packed activations x[64][K] : k-tile kb at kb*1024; token block tb=t/32 at +tb*512;
//                                 inside: lane = (t%32) + 32*((k%16)/8), e = k%8   (B-operand fragment)
//  QF / KF  q or k rows [rows][128] per head: 32-row block b, d-step s=d/16 at (b*8+s)*512,
//                                 inside: lane = (row%32) + 32*((d%16)/8), e = d%8
//  VF  v rows, transposed fragments : 32-key block b, d-block db=d/32, key-step s2 at (b*8+db*2+s2)*512,
//                                 inside: lane = (d%32) + 32*hh, e, where key%32 = 16*s2 + (e&3) + 8*(e>>2) + 4*hh
//      (the key order in which the S^T accumulator registers of the QK MFMA already sit, so P needs
//       no cross-lane movement before the PV MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <math.h>

// Storage / MFMA-input type of THIS build of the library.  Every kernel is written once; the build compiles the sources twice:
//   LA_DTYPE 0 -> liblookahead_hip.so      bfloat16, v_mfma_f32_32x32x16_bf16 (BASELINE's dtype)
//   LA_DTYPE 1 -> liblookahead_hip_f16.so  float16,  v_mfma_f32_32x32x16_f16  (what the reference's own examples and benchmarks run:
//                 lookahead/benchmarks/llama_benchmark.py:27, examples/llama_example.py:19)
// Same fragment layouts, same rounding POINTS (every nn.Linear output, RoPE product / sum, residual add, SiLU, softmax P and attention
// output are rounded to the storage type exactly where the reference's eager graph rounds); `bf16_t` / bf2f / f2bf / bfr name "the
// 16-bit storage element" and its conversions in both builds.  la_abi_dtype() reports which build a process loaded.
#ifndef LA_DTYPE
#define LA_DTYPE 0
#endif
typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define LA_TB 64          // tokens per block (rows of every activation matrix)

__device__ __forceinline__ float bf2f(bf16_t v) {
    if constexpr (LA_DTYPE == 1) return (float)__builtin_bit_cast(_Float16, v);
    else return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even, NaN preserving: same rounding torch uses for float -> bfloat16
// (the gfx950 converter v_cvt_pk_bf16_f32: IEEE round-to-nearest-even, quiet NaN — one instruction; the integer form
//  u + 0x7fff + ((u >> 16) & 1) it replaces is kept as f2bf_int for the host-visible packers' reference)
__device__ __forceinline__ bf16_t f2bf(float f) {
    if constexpr (LA_DTYPE == 1) return __builtin_bit_cast(unsigned short, (_Float16)f);       // v_cvt_f16_f32: round-to-nearest-even, inf on overflow (as torch)
    else return __builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ bf16_t f2bf_int(float f) {
    const uint32_t u = __float_as_uint(f);
    const uint32_t r = u + 0x7fffu + ((u >> 16) & 1u);
    const bool nan = (u & 0x7fffffffu) > 0x7f800000u;
    return (bf16_t)((nan ? (u | 0x400000u) : r) >> 16);
}
__device__ __forceinline__ float bfr(float f) { return bf2f(f2bf(f)); }
// the 32x32x16 MFMA of the build's storage type (identical operand layouts): LA_MFMA(a, b, c, 0, 0, 0)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
#if LA_DTYPE == 1
#define LA_MFMA(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (a)), __builtin_bit_cast(f16x8, (b)), (c), (x), (y), (z))
#else
#define LA_MFMA(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), (x), (y), (z))
#endif
// attn_weights = (q @ k^T in the storage type) / sqrt(head_dim), rounded to the storage type (modeling_llama.py:270).  `qk` is a kernel
// argument (la_qk_scale(head_dim), la_kernels.h): bf16 build: fp32(1 / sqrt(head_dim)) — x / sqrt(d) == x * fp32(1 / sqrt(d)) after rounding for
// EVERY finite bf16 x and every d = 16, 24, .. 128 (exhaustive check, tests/test_oracle_llama.py::test_attention_scale_as_multiply_is_exact;
// la_llama_create repeats it for the d it is given) — one multiply; fp16 build: fp32(sqrt(head_dim)) — 52 of the 65536 fp16 patterns differ
// at d = 128 (54 at d = 32), so the fp16 build divides (torch: fp32 division of the upcast value, then one rounding).
__host__ __device__ __forceinline__ float la_qk_scale(int head_dim) {
#if LA_DTYPE == 1
    return (float)sqrt((double)head_dim);
#else
    return (float)(1.0 / sqrt((double)head_dim));
#endif
}
__device__ __forceinline__ float attn_scale(float s, float qk) {
    if constexpr (LA_DTYPE == 1) return bfr(bfr(s) / qk);
    else return bfr(bfr(s) * qk);
}

// accumulator register r of a 32x32 MFMA tile -> row inside the tile (col = lane&31)
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__host__ __device__ __forceinline__ size_t xp_offset(int t, int k) {
    return (size_t)(k >> 4) * 1024 + (size_t)(t >> 5) * 512 + (size_t)(((t & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7));
}
// row-fragment layout (QF/KF): element (row, d) of a [rows][128] matrix
__host__ __device__ __forceinline__ size_t rf_offset(int row, int d) {
    return ((size_t)(row >> 5) * 8 + (d >> 4)) * 512 + (size_t)(((row & 31) + 32 * ((d >> 3) & 1)) * 8 + (d & 7));
}
// transposed-fragment layout (VF): element (key, d)
__host__ __device__ __forceinline__ size_t vf_offset(int key, int d) {
    int kk = key & 31, s2 = kk >> 4, rr = kk & 15;
    int hh = (rr >> 2) & 1, e = (rr & 3) + 4 * (rr >> 3);
    return ((size_t)(key >> 5) * 8 + (d >> 5) * 2 + s2) * 512 + (size_t)(((d & 31) + 32 * hh) * 8 + e);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Router tail of a Mixtral row (MixtralSparseMoeBlock.forward, mixtral/modeling_mixtral.py:723-729): softmax over the experts' logits in
// fp32, top-k, renormalise, cast back.  Called by ALL 64 lanes of wave 0 after the per-wave partial logits shr[wave][e] are in LDS; lane e
// (< E) owns expert e and writes route_row[e] (weight of expert e for this row, 0 = not routed).  Bit-identical to the one-thread form it
// replaces (rounds 1-5: thread 0 walked lg[] / pr[] / pick[] with dynamic indices — private arrays the compiler keeps in scratch, a chain of
// dependent scratch round trips worth ~8 us per launch next to the 7 us of the norm itself): same sums in the same order (w = 0..7 per logit,
// e = 0..n-1 for the denominator, pick order for the renormaliser), max is exact in any order, the earliest expert wins a tie.
template <int E>
__device__ __forceinline__ void moe_router_tail(const float (*shr)[E], int n_experts, int top_k, bool live, float* __restrict__ route_row) {
    const int lane = threadIdx.x & 63;
    const int e = lane < E ? lane : E - 1;
    const bool valid = lane < E && lane < n_experts;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += shr[w][e];
    const float lg = bfr(v);
    const float lgm = valid ? lg : -INFINITY;
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < E; ++i) mx = fmaxf(mx, __shfl(lgm, i, 64));
    float pr = valid ? expf(lg - mx) : 0.f;
    float den = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) den += __shfl(pr, i, 64);            // + 0.f for the lanes past n_experts: exact
    pr = pr / den;
    bool taken = false;
    float ksum = 0.f;
    for (int k = 0; k < top_k; ++k) {
        const float cand = (valid && !taken) ? pr : -1.f;             // probabilities are >= 0
        float m = -1.f;
#pragma unroll
        for (int i = 0; i < E; ++i) m = fmaxf(m, __shfl(cand, i, 64));
        const unsigned long long hit = __ballot(valid && !taken && cand == m);
        const unsigned long long avail = __ballot(valid && !taken);
        // the earliest expert among equals (strict > in the serial scan); NaN probabilities compare false everywhere: the serial scan then
        // keeps the first expert not taken yet, and so does this
        const int best = hit ? __builtin_ctzll(hit) : avail ? __builtin_ctzll(avail) : 0;
        ksum += __shfl(pr, best, 64);
        if (lane == best) taken = true;
    }
    if (lane < E) route_row[lane] = (valid && taken && live) ? bfr(pr / ksum) : 0.f;
}
