// la_oproj_merge.hip — LAB experiment of round 6 (review item 1b): tree attention over key splits with NO combine launch; o_proj merges
// the splits' un-normalised (O, m, l) partials while it builds its x operand ("merge on load").
//
// Reference arithmetic: LlamaAttention.forward (lookahead/lookahead/models/llama/modeling_llama.py:270-296, fp32 softmax) followed by
// o_proj (:298-308).  Same numbers as the key-split pair k_tree_attn + k_attn_combine + k_gemm64<2, SLAB, 8, 8>: the merge below is
// k_attn_combine's expression per (head, token, 8 head dims), evaluated by the o_proj wave that needs that fragment, and the MFMA
// chain / cross-wave reduction / slab store are k_gemm64's.
//
// Geometry (7B: o_k = 4096, 4 K splits): a workgroup owns 64 output rows x one K split of 64 k-tiles = 8 heads; each of its 8
// waves owns ONE head (8 k-tiles), i.e. 64 tokens x 128 dims x NS partials = NS x 32 KiB of fp32 instead of 16 KiB of bf16.
// Measured (profiles/r06_attn_merge_on_load_ab.txt): slower than the single-launch attention + plain o_proj; kept as lab knob 33.
#include "../la_common.h"
#include "../la_kernels.h"

#define LA_NEG (-1.0e30f)

struct OMergeArgs {
    const float* opart;      // [nh][NS][64][128] un-normalised partial outputs
    const float* mpart;      // [nh][NS][64] running maxima
    const float* lpart;      // [nh][NS][64] running sums
    float* slabs;            // [ksplit][64][N] fp32 split-K partial sums of o_proj
    int N;
};

template <int NS>
__global__ __launch_bounds__(512) void k_oproj_merge(const bf16_t* __restrict__ wp_s, int K16_s, OMergeArgs a) {
    __shared__ __attribute__((aligned(16))) float red[8][2 * 2 * 16 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nb0 = blockIdx.x * 2, ks = blockIdx.y, ksplit = gridDim.y;
    const int t0 = (int)(((long)K16_s * ks) / ksplit);
    const int wb = t0 + wave * 8;                      // this wave's 8 k-tiles = head wb / 8
    const int h = wb >> 3;
    const bf16x8* __restrict__ wbase = (const bf16x8*)wp_s;
    // weights first: the whole K range of the wave in flight (as k_gemm64<2, SLAB, 8, 8>)
    bf16x8 fa[8][2];
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
            fa[d][rb] = __builtin_nontemporal_load(wbase + (size_t)((nb0 + rb) * K16_s + wb + d) * 64 + lane);
    // merge weights of this lane's two token columns (tb = 0, 1): w[s] = exp(m_s - M), inv = 1 / sum w_s l_s   (k_attn_combine)
    const int tl = lane & 31, hh = lane >> 5;
    float w[2][NS], inv[2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int tok = tb * 32 + tl;
        float ms[NS], ls[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            ms[s] = a.mpart[((size_t)h * NS + s) * LA_TB + tok];
            ls[s] = a.lpart[((size_t)h * NS + s) * LA_TB + tok];
        }
        float M = LA_NEG;
#pragma unroll
        for (int s = 0; s < NS; ++s) M = fmaxf(M, ms[s]);
        float L = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) { w[tb][s] = __expf(ms[s] - M); L += w[tb][s] * ls[s]; }
        inv[tb] = 1.0f / L;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][tb][i] = 0.f;
    // k-tile d of the head = dims [16 d, 16 d + 16): this lane's B fragment = 8 dims d8 * 8 .. of token tb * 32 + tl, d8 = 2 d + hh
    f32x4 po[2][2][NS][2];                              // [buffer][tb][split][half]
    auto issue = [&](int d, int buf) {
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float* op = a.opart + (((size_t)h * NS + s) * LA_TB + (tb * 32 + tl)) * 128 + (2 * d + hh) * 8;
                po[buf][tb][s][0] = *(const f32x4*)op;
                po[buf][tb][s][1] = *(const f32x4*)(op + 4);
            }
    };
    issue(0, 0);
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        if (d + 1 < 8) issue(d + 1, (d + 1) & 1);
        bf16x8 fb[2];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            float x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float ww = w[tb][s];
                const f32x4 o0 = po[d & 1][tb][s][0], o1 = po[d & 1][tb][s][1];
                x[0] += ww * o0[0]; x[1] += ww * o0[1]; x[2] += ww * o0[2]; x[3] += ww * o0[3];
                x[4] += ww * o1[0]; x[5] += ww * o1[1]; x[6] += ww * o1[2]; x[7] += ww * o1[3];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) fb[tb][j] = (short)f2bf(x[j] * inv[tb]);
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            acc[rb][0] = LA_MFMA(fa[d][rb], fb[0], acc[rb][0], 0, 0, 0);
            acc[rb][1] = LA_MFMA(fa[d][rb], fb[1], acc[rb][1], 0, 0, 0);
        }
    }
    // deterministic cross-wave reduction + slab store: k_gemm64<2, EPI_SLAB, 8, 8>'s, verbatim in structure
    f32x4* red4 = (f32x4*)&red[0][0];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 v = {acc[rb][tb][4 * i4], acc[rb][tb][4 * i4 + 1], acc[rb][tb][4 * i4 + 2], acc[rb][tb][4 * i4 + 3]};
                red4[((wave * 4 + rb * 2 + tb) * 4 + i4) * 64 + lane] = v;
            }
    __syncthreads();
    const int tbo = wave & 1, g0 = wave >> 1;
    const int tok = tbo * 32 + tl;
    float* o = a.slabs + (size_t)ks * LA_TB * a.N;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 8; ++p) v += red4[((p * 4 + rb * 2 + tbo) * 4 + g0) * 64 + lane];
        *(f32x4*)(o + (size_t)tok * a.N + (nb0 + rb) * 32 + 8 * g0 + 4 * hh) = v;
    }
}

// N output features, K = nh * 128 input features, ksplit K splits with (K / 16 / ksplit) == 64 k-tiles per workgroup (8 heads)
int lk_oproj_merge(hipStream_t st, const void* wp, int N, int K, int ksplit, int nsplit, const float* opart, const float* mpart,
                   const float* lpart, float* slabs) {
    if (N % 64 || K % 16 || ksplit < 1 || (K / 16) % ksplit || (K / 16) / ksplit != 64) return -1;
    OMergeArgs a{opart, mpart, lpart, slabs, N};
    const dim3 g(N / 64, ksplit);
    if (nsplit == 2) k_oproj_merge<2><<<g, 512, 0, st>>>((const bf16_t*)wp, K / 16, a);
    else if (nsplit == 4) k_oproj_merge<4><<<g, 512, 0, st>>>((const bf16_t*)wp, K / 16, a);
    else return -1;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
