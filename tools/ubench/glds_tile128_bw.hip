// LDS-DMA ceiling for 256x128 tiles, 4 waves per workgroup, 2 workgroups per CU, blk layout (24 KiB per K step).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(256) void tile_stream(const char* __restrict__ A, const char* __restrict__ W, int mtn, int ntn, int ksteps, long long* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, q = nwg >> 3, r = nwg & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int BWN = 8;                                   // band of 8 n-tiles of 128
    const int band = t / (mtn * BWN), rr = t - band * (mtn * BWN);
    const int tm = rr / BWN, tn = band * BWN + (rr % BWN);
    // A: 256 rows x 64 B = 1024 slots -> 4 per thread; W: 128 rows -> 512 slots -> 2 per thread
    long long aoff[4], woff[2];
    for (int i = 0; i < 4; ++i) { const int L = i * 256 + tid, row = L >> 2, c = (L & 3) ^ ((row >> 2) & 3); aoff[i] = (long long)tm * 256 * 2048 + row * 64 + c * 16; }
    for (int i = 0; i < 2; ++i) { const int L = i * 256 + tid, row = L >> 2, c = (L & 3) ^ ((row >> 2) & 3); woff[i] = (long long)(tn >> 1) * 256 * 2048 + ((tn & 1) * 128 + row) * 64 + c * 16; }
    auto stage = [&](int s) {
        const long long kk = (long long)(s % ksteps) * 16384;
        char* st = lds + (s % 3) * 24576;
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(A + aoff[i] + kk), (lptr_t)(st + (i * 256 + wave * 64) * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(W + woff[i] + kk), (lptr_t)(st + 16384 + (i * 256 + wave * 64) * 16), 16, 0, 0);
    };
    const long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < 2; ++s) stage(s);
    for (int s = 0; s < ksteps - 2; ++s) { wait_vmcnt<6>(); __builtin_amdgcn_s_barrier(); stage(s + 2); }
    wait_vmcnt<0>();
    __syncthreads();
    if (tid == 0) { out[blockIdx.x * 2] = __builtin_readcyclecounter() - t0; out[blockIdx.x * 2 + 1] = lds[5]; }
}

int main() {
    const int M = 50432, K = 1024;
    long long* d_out; hipMalloc(&d_out, 1 << 20);
    hipFuncSetAttribute((const void*)tile_stream, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
    for (int N : {3072, 1024, 4096}) {
        char *A, *W; hipMalloc(&A, (size_t)(M / 256 + 1) * 256 * K * 2); hipMalloc(&W, (size_t)N * K * 2);
        hipMemset(A, 1, (size_t)(M / 256 + 1) * 256 * K * 2); hipMemset(W, 1, (size_t)N * K * 2);
        const int mtn = M / 256, ntn = N / 128, blocks = mtn * ntn;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float ms = 0;
        for (int it = 0; it < 2; ++it) {
            hipEventRecord(a);
            hipLaunchKernelGGL(tile_stream, dim3(blocks), dim3(256), 73728, 0, A, W, mtn, ntn, K / 32, d_out);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        }
        const double bytes = (double)blocks * (K / 32) * 24576;
        const double flops = 2.0 * M * N * K;
        printf("256x128 tiles N=%4d: %7.2f TB/s through LDS-DMA (%.3f ms, %d tiles) -> GEMM would need >= %.3f ms => DMA-bound ceiling %.0f TFLOP/s\n",
               N, bytes / ms / 1e9, ms, blocks, ms, flops / ms / 1e9);
        hipFree(A); hipFree(W);
    }
    return 0;
}
