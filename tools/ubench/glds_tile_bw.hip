// Micro-benchmark: LDS-DMA throughput for the GEMM's actual access pattern (256x256 tile, BK=32,
// A panel shared by the n-tiles of a band, W panel shared by the m-tiles), as a function of the row pitch.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(512) void tile_stream(const char* __restrict__ A, const char* __restrict__ W, int pitchA, int pitchW,
                                                   int mtn, int ntn, int ksteps, int reps, long long* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, q = nwg >> 3, r = nwg & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int band = t / (mtn * 4), rr = t - band * (mtn * 4);
    const int tm = rr / 4, tn = band * 4 + (rr - (rr / 4) * 4);
    // per thread: 2 A slots + 2 W slots per step (slot L = r*512+tid -> row L/4, chunk L%4)
    long long aoff[2], woff[2];
    for (int i = 0; i < 2; ++i) {
        const int L = i * 512 + tid, row = L >> 2, c = (L & 3) ^ ((row >> 2) & 3);
        if (pitchA > 0) {
            aoff[i] = (long long)(tm * 256 + row) * pitchA + c * 16;
            woff[i] = (long long)(tn * 256 + row) * pitchW + c * 16;
        } else {   // K-blocked: [tile][kstep][256 rows][64 B]
            aoff[i] = (long long)tm * 256 * 2048 + row * 64 + c * 16;
            woff[i] = (long long)tn * 256 * 2048 + row * 64 + c * 16;
        }
    }
    const long long t0 = __builtin_readcyclecounter();
    const int steps = ksteps * reps;
    auto stage = [&](int s) {
        const int kk = (s % ksteps) * (pitchA > 0 ? 64 : 16384);
        char* st = lds + (s & 3) * 32768;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(A + aoff[i] + kk), (lptr_t)(st + (i * 512 + wave * 64) * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(W + woff[i] + kk), (lptr_t)(st + 16384 + (i * 512 + wave * 64) * 16), 16, 0, 0);
        }
    };
    for (int s = 0; s < 3; ++s) stage(s);
    for (int s = 0; s < steps - 3; ++s) { wait_vmcnt<8>(); __builtin_amdgcn_s_barrier(); stage(s + 3); }
    wait_vmcnt<0>();
    __syncthreads();
    if (tid == 0) { out[blockIdx.x * 2] = __builtin_readcyclecounter() - t0; out[blockIdx.x * 2 + 1] = lds[5]; }
}

int main() {
    const int M = 50432, K = 1024;
    long long* d_out; hipMalloc(&d_out, 1 << 20);
    hipFuncSetAttribute((const void*)tile_stream, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int N : {3072, 1024}) for (int pad : {0, 64, -1}) {
        const int pitch = pad < 0 ? 2048 : K * 2 + pad;
        char *A, *W; hipMalloc(&A, (size_t)M * pitch); hipMalloc(&W, (size_t)N * pitch);
        hipMemset(A, 1, (size_t)M * pitch); hipMemset(W, 1, (size_t)N * pitch);
        const int mtn = M / 256, ntn = N / 256, blocks = mtn * ntn;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float ms = 0;
        for (int it = 0; it < 2; ++it) {
            hipEventRecord(a);
            hipLaunchKernelGGL(tile_stream, dim3(blocks), dim3(512), 131072, 0, A, W, pad < 0 ? -1 : pitch, pad < 0 ? -1 : pitch, mtn, ntn, K / 32, 1, d_out);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        }
        const double bytes = (double)blocks * (K / 32) * 32768;
        printf("N=%4d K=%d pitch=%5d B (pad %4d): %7.2f TB/s aggregate through LDS-DMA  (%.3f ms, %d tiles)\n", N, K, pitch, pad, bytes / ms / 1e9, ms, blocks);
        hipFree(A); hipFree(W);
    }
    return 0;
}
