// What do s_memtime and s_memrealtime count on gfx950?  One wave spins for a fixed number of s_memrealtime ticks; the host times the kernel with HIP events.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/clock_probe.hip -o tools/ubench/clock_probe && tools/ubench/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(long long* out, long long spin_ticks) {
    const long long r0 = (long long)__builtin_amdgcn_s_memrealtime(), c0 = (long long)__builtin_amdgcn_s_memtime();
    long long r1 = r0;
    while (r1 - r0 < spin_ticks) { __builtin_amdgcn_s_sleep(32); r1 = (long long)__builtin_amdgcn_s_memrealtime(); }
    out[0] = (long long)__builtin_amdgcn_s_memtime() - c0; out[1] = r1 - r0;
}
__global__ void burn(float* x, int iters) {      // keeps the chip busy: dependent FMAs on every CU
    float a = x[threadIdx.x], b = 1.0001f;
    for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; a = a * b - 0.5f; }
    x[threadIdx.x + blockIdx.x * blockDim.x] = a;
}
int main() {
    long long* d; hipMalloc(&d, 16); float* x; hipMalloc(&x, 4096 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int load = 0; load < 2; ++load)
        for (long long ticks : {1000000ll, 5000000ll, 20000000ll}) {
            hipStream_t s2; hipStreamCreate(&s2);
            if (load) hipLaunchKernelGGL(burn, dim3(4096), dim3(256), 0, s2, x, 4000000);
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, ticks);
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("load %d: memrealtime ticks %lld, memtime ticks %lld, event %.3f ms -> memrealtime %.1f MHz, memtime %.1f MHz\n", load, h[1], h[0], ms,
                   h[1] / ms / 1e3, h[0] / ms / 1e3);
            hipDeviceSynchronize(); hipStreamDestroy(s2);
        }
    return 0;
}
