// probe_l2.hip — what one MI355X delivers when EVERY workgroup re-reads the same few MB (the x operand of a weight-stationary
// M = 512 GEMM): aggregate bytes/s by access form.  Build: hipcc --offload-arch=gfx950 -O3 -o probe_l2 probe_l2.hip
//   mode 0: register loads (8 x 16 B per lane in flight), every workgroup sweeps the SAME buffer
//   mode 1: the same through LDS-DMA (global_load_lds_dwordx4 into a 64 KiB ring, 8 pieces per wave in flight)
//   mode 2: register loads, every workgroup its OWN slice of a large buffer (no sharing: HBM / Infinity Cache stream)
//   mode 3: mode 0 with the sweep of each workgroup rotated by its index (same bytes, different addresses at any instant)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k_read(const i32x4* __restrict__ buf, size_t chunks_per_wg, size_t wg_stride, int reps, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i32x4* base = buf + (size_t)blockIdx.x * wg_stride;
    const size_t per_wave = chunks_per_wg / 8;             // 16-byte chunks per wave
    const size_t rot = MODE == 3 ? ((size_t)blockIdx.x * 977 * 64) % per_wave : 0;
    i32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        if constexpr (MODE == 1) {
            for (size_t c = 0; c < per_wave; c += 64 * 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    __builtin_amdgcn_global_load_lds((gptr_t)(base + wave * per_wave + c + j * 64 + lane), (lptr_t)(lds + (wave * 8 + j) * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            for (size_t c = 0; c < per_wave; c += 64 * 8) {
                i32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    size_t o = c + j * 64 + lane + rot;
                    o = o >= per_wave ? o - per_wave : o;
                    v[j] = base[wave * per_wave + o];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += v[j];
            }
        }
    }
    if (MODE == 1) acc[0] = ((int*)lds)[threadIdx.x];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678) sink[0] = 1;
}

// mode 4: the operand mix of the wide GEMM: waves 0,1 stream private (HBM) data, 4 pieces per step, 5 steps deep; waves 2..7 read the
// shared buffer, 5 pieces per step, 2 steps deep; all by LDS-DMA.  wshare = 0: only the shared readers run; 2: only the streamers.
__global__ __launch_bounds__(512) void k_mix(const i32x4* __restrict__ shared_buf, size_t shared_chunks, const i32x4* __restrict__ priv, size_t priv_chunks_per_wg,
                                             int steps, int which, int nt, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 2) {
        if (which == 0) return;
        const i32x4* base = priv + (size_t)blockIdx.x * priv_chunks_per_wg + (size_t)wave * (priv_chunks_per_wg / 2);
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (nt) __builtin_amdgcn_global_load_lds((gptr_t)(base + ((size_t)s * 4 + j) * 64 + lane), (lptr_t)(lds + (wave * 4 + j) * 1024), 16, 0, 2);
                else __builtin_amdgcn_global_load_lds((gptr_t)(base + ((size_t)s * 4 + j) * 64 + lane), (lptr_t)(lds + (wave * 4 + j) * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        }
    } else {
        if (which == 2) return;
        const int xw = wave - 2;
        const size_t per_wave = shared_chunks / 6;
        const i32x4* base = shared_buf + xw * per_wave;
        size_t c = 0;
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                __builtin_amdgcn_global_load_lds((gptr_t)(base + c + lane), (lptr_t)(lds + 8192 + (xw * 5 + j) * 1024), 16, 0, 0);
                c += 64; c = c >= per_wave ? 0 : c;
            }
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (((int*)lds)[threadIdx.x] == 0x12345678) sink[0] = 1;
}

int main() {
    const size_t big = (size_t)4 << 30;                     // 4 GiB
    i32x4* buf; int* sink;
    CK(hipMalloc(&buf, big)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, big));
    CK(hipFuncSetAttribute((const void*)k_read<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t sizes[] = {1u << 20, 4u << 20, 16u << 20};
    for (int mode = 0; mode < 4; ++mode) {
        for (size_t sz : sizes) {
            size_t chunks = sz / 16, stride = 0;
            int reps = (int)((64u << 20) / sz);             // 64 MiB per workgroup
            if (mode == 2) { stride = chunks; if (sz * 256 > ((size_t)1 << 30)) continue; reps = reps > 16 ? 16 : reps; }
            for (int it = 0; it < 2; ++it) {
                CK(hipEventRecord(e0));
                switch (mode) {
                    case 0: k_read<0><<<256, 512>>>(buf, chunks, stride, reps, sink); break;
                    case 1: k_read<1><<<256, 512, 65536>>>(buf, chunks, stride, reps, sink); break;
                    case 2: k_read<2><<<256, 512>>>(buf, chunks, stride, reps, sink); break;
                    default: k_read<3><<<256, 512>>>(buf, chunks, stride, reps, sink); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it == 1) printf("mode %d  %3zu MiB per workgroup-sweep x %3d: %8.1f us  %7.2f TB/s aggregate  %6.1f B/clk/CU @2.4GHz\n", mode, sz >> 20, reps,
                                    ms * 1e3, 256.0 * sz * reps / ms / 1e9, 256.0 * sz * reps / (ms * 1e-3) / 256 / 2.4e9);
            }
        }
    }
    CK(hipFuncSetAttribute((const void*)k_mix, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    {
        const int steps = 512;                                  // per workgroup: 2 x 4 KiB private + 6 x 5 KiB shared per step
        const size_t priv_chunks = (size_t)steps * 8 * 64;      // 16-byte chunks per workgroup (16 MiB... 2048*8 KiB)
        const i32x4* priv = buf + (64u << 20) / 16;
        for (int nt = 0; nt < 2; ++nt)
            for (int which = 0; which < 3; ++which)
                for (int it = 0; it < 2; ++it) {
                    // rotate the private region so that it never sits in the Infinity Cache from the previous launch
                    const i32x4* pp = priv + (size_t)((which * 2 + it + nt * 6) % 3) * ((size_t)256 * priv_chunks);
                    CK(hipEventRecord(e0));
                    k_mix<<<256, 512, 65536>>>(buf, (4u << 20) / 16, pp, priv_chunks, steps, which, nt, sink);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    const double sh = which == 2 ? 0 : 256.0 * steps * 30 * 1024, pv = which == 0 ? 0 : 256.0 * steps * 8 * 1024;
                    if (it == 1) printf("mix nt=%d %s: %8.1f us  shared %6.2f TB/s (%5.1f B/clk/CU)  private %5.2f TB/s\n", nt,
                                        which == 0 ? "shared readers only " : which == 1 ? "shared + HBM streamers" : "HBM streamers only  ",
                                        ms * 1e3, sh / ms / 1e9, sh / (ms * 1e-3) / 256 / 2.4e9, pv / ms / 1e9);
                }
    }
    return 0;
}
