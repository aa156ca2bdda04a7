// probe_mfma.hip — the issue rate of the bf16 MFMAs on one MI355X: 256 workgroups x 8 waves, every wave a loop of 8 independent
// accumulators.  Build: hipcc --offload-arch=gfx950 -O3 -o probe_mfma probe_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(short)))) short bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SHAPE>
__global__ __launch_bounds__(512) void k_mfma(int iters, float* out) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + i); }
    if constexpr (SHAPE == 0) {
        f32x16 acc[8];
        for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
        }
        float s = 0.f;
        for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        f32x4 acc[8];
        for (int t = 0; t < 8; ++t) for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
        }
        float s = 0.f;
        for (int t = 0; t < 8; ++t) for (int i = 0; i < 4; ++i) s += acc[t][i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

// 8 MFMAs (32x32x16) per iteration with R ds_read_b128 (lane-linear, conflict-free) issued before them, software-pipelined like
// the wide GEMM: the reads of iteration i+1 are in flight under the MFMAs of iteration i.
template <int R>
__global__ __launch_bounds__(512) void k_mfma_lds(int iters, float* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) ((int*)lds)[i] = 0x3f803f80 + i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds + (wave * 8) * 1024 + lane * 16;
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    bf16x8 f0[R > 0 ? R : 1], f1[R > 0 ? R : 1];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + i); }
    for (int r = 0; r < R; ++r) { f0[r] = a; f1[r] = b; }
#pragma unroll
    for (int r = 0; r < R; ++r) asm volatile("ds_read_b128 %0, %1" : "=v"(f0[r]) : "v"(base + (r % 8) * 1024u) : "memory");
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("ds_read_b128 %0, %1" : "=v"(f1[r]) : "v"(base + (r % 8) * 1024u) : "memory");
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(R) : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R > 0 ? f0[t % (R > 0 ? R : 1)] : a, R > 1 ? f0[(t + 1) % (R > 0 ? R : 1)] : b, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("ds_read_b128 %0, %1" : "=v"(f0[r]) : "v"(base + (r % 8) * 1024u) : "memory");
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(R) : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(R > 0 ? f1[t % (R > 0 ? R : 1)] : a, R > 1 ? f1[(t + 1) % (R > 0 ? R : 1)] : b, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int R> void run_lds(int threads, hipEvent_t e0, hipEvent_t e1, float* out) {
    const int iters = 4096;
    CK(hipFuncSetAttribute((const void*)k_mfma_lds<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int it = 0; it < 2; ++it) {
        CK(hipEventRecord(e0));
        k_mfma_lds<R><<<256, threads, 65536>>>(iters, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double n = 256.0 * (threads / 64) * iters * 8;
        if (it == 1) printf("32x32x16 + %2d ds_read_b128 per 8 MFMAs, %d waves/CU: %8.1f us  %7.1f TFLOP/s\n", R, threads / 64, ms * 1e3, n * 32768.0 / ms / 1e9);
    }
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 512 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4096;
    for (int shape = 0; shape < 2; ++shape)
        for (int threads : {256, 512})
            for (int it = 0; it < 2; ++it) {
                CK(hipEventRecord(e0));
                if (shape == 0) k_mfma<0><<<256, threads>>>(iters, out); else k_mfma<1><<<256, threads>>>(iters, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double flop = (shape == 0 ? 32.0 * 32 * 16 * 2 : 16.0 * 16 * 32 * 2), n = 256.0 * (threads / 64) * iters * 8;
                if (it == 1) printf("%s  %d waves/CU: %8.1f us  %7.1f TFLOP/s  %5.1f clk per MFMA per SIMD @2.4GHz\n", shape == 0 ? "32x32x16 bf16" : "16x16x32 bf16", threads / 64,
                                    ms * 1e3, n * flop / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 8.0 * (threads / 64) / 4.0));
            }
    for (int threads : {256, 512}) {
        run_lds<0>(threads, e0, e1, out); run_lds<4>(threads, e0, e1, out); run_lds<6>(threads, e0, e1, out);
        run_lds<8>(threads, e0, e1, out); run_lds<12>(threads, e0, e1, out);
    }
    return 0;
}
