// Micro-benchmark: how fast can one CU pull L2-resident data into LDS with global_load_lds?
// Each workgroup (8 waves) streams its own `span` bytes (re-read `reps` times) into an LDS ring with
// `depth` x 32 KiB stages, counted vmcnt, one barrier per stage -- the GEMM's staging skeleton without MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int DEPTH, bool BARRIER>
__global__ __launch_bounds__(512) void stream_kernel(const char* __restrict__ src, size_t span, int steps, long long* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)blockIdx.x * span;
    const long long t0 = __builtin_readcyclecounter();
    // per step: 32 KiB = 4 x (512 threads x 16 B)
    auto stage = [&](int s) {
        const size_t off = ((size_t)s * 32768) % span;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            __builtin_amdgcn_global_load_lds((gptr_t)(base + off + (r * 512 + tid) * 16),
                                             (lptr_t)(lds + (s % DEPTH) * 32768 + (r * 512 + wave * 64) * 16), 16, 0, 0);
    };
    for (int s = 0; s < DEPTH - 1; ++s) stage(s);
    for (int s = 0; s < steps; ++s) {
        wait_vmcnt<4 * (DEPTH - 2)>();
        if (BARRIER) __builtin_amdgcn_s_barrier();
        stage(s + DEPTH - 1);
    }
    wait_vmcnt<0>();
    __syncthreads();
    if (tid == 0) { out[blockIdx.x * 2] = __builtin_readcyclecounter() - t0; out[blockIdx.x * 2 + 1] = lds[123]; }
}

template <int DEPTH, bool BARRIER>
void run(const char* src, size_t span, int steps, int blocks, long long* d_out, const char* label) {
    hipFuncSetAttribute((const void*)stream_kernel<DEPTH, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * 32768);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 2; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL((stream_kernel<DEPTH, BARRIER>), dim3(blocks), dim3(512), DEPTH * 32768, 0, src, span, steps, d_out);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * steps * 32768;
    printf("%-34s depth %d  blocks %4d  span/blk %7zu KiB : %7.2f TB/s aggregate, %6.1f GB/s per CU (%.3f ms)\n", label, DEPTH, blocks,
           span >> 10, bytes / ms / 1e9, bytes / ms / 1e6 / (blocks < 256 ? blocks : 256), ms);
}

int main() {
    const size_t total = (size_t)256 << 20;
    char* src; hipMalloc(&src, total); hipMemset(src, 1, total);
    long long* d_out; hipMalloc(&d_out, 4096 * 16);
    const int steps = 4096;
    // span per block: 64 KiB (L2/L1 resident), 1 MiB (L2: 256 MiB total / XCD share), shared small buffer
    for (size_t span : {(size_t)65536, (size_t)1 << 20}) {
        run<2, true>(src, span, steps, 256, d_out, "1 tile in flight, barrier");
        run<4, true>(src, span, steps, 256, d_out, "3 tiles in flight, barrier");
        run<4, false>(src, span, steps, 256, d_out, "3 tiles in flight, no barrier");
    }
    // every block reads the SAME 4 MiB region (pure L2 hits after the first touch)
    hipFuncSetAttribute((const void*)stream_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int it = 0; it < 2; ++it) {
            hipEventRecord(a);
            hipLaunchKernelGGL((stream_kernel<4, true>), dim3(256), dim3(512), 4 * 32768, 0, src, (size_t)32768, 8, d_out);
            hipEventRecord(b); hipEventSynchronize(b);
        }
    }
    return 0;
}
