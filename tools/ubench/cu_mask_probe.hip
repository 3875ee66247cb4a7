// Which CUs does a stream created with hipExtStreamCreateWithCUMask run on, and what does a GEMM-like / streaming kernel get from a subset?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/cu_mask_probe.hip -o tools/ubench/cu_mask_probe && tools/ubench/cu_mask_probe
// Census: every workgroup records (XCC_ID, HW_ID) and spins for a few microseconds so that the grid spreads over everything it may use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <set>
__global__ void census(unsigned* out, long long spin) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
        out[blockIdx.x * 2] = xcc; out[blockIdx.x * 2 + 1] = hw;
    }
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}
__global__ void stream_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
static void run_census(const char* name, hipStream_t s, int wgs) {
    unsigned* d; hipMalloc(&d, wgs * 8); hipMemset(d, 0xff, wgs * 8);
    hipLaunchKernelGGL(census, dim3(wgs), dim3(64), 0, s, d, 2000);        // 20 us at 100 MHz
    hipStreamSynchronize(s);
    std::vector<unsigned> h(wgs * 2); hipMemcpy(h.data(), d, wgs * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> cus;     // xcc -> set of (se, sh, cu)
    for (int i = 0; i < wgs; ++i) cus[h[2 * i] & 0xf].insert((h[2 * i + 1] >> 8) & 0xff);   // cu_id[11:8], sh_id[12], se_id[15:13]
    printf("%-34s", name);
    int tot = 0;
    for (auto& kv : cus) { printf(" xcc%u:%2zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
    printf("  -> %d distinct CUs\n", tot);
    hipFree(d);
}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("CUs %d\n", prop.multiProcessorCount);
    hipStream_t s0; hipStreamCreate(&s0);
    run_census("unmasked", s0, 4096);
    struct { const char* name; uint32_t m[8]; } masks[] = {
        {"low 224 bits", {0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0}},
        {"high 32 bits", {0, 0, 0, 0, 0, 0, 0, 0xffffffff}},
        {"every 8th bit (32 CUs)", {0x01010101, 0x01010101, 0x01010101, 0x01010101, 0x01010101, 0x01010101, 0x01010101, 0x01010101}},
        {"bits 0-7 (8 CUs)", {0xff, 0, 0, 0, 0, 0, 0, 0}},
        {"bits 0,8,16,..56 (8 CUs)", {0x01010101, 0x01010101, 0, 0, 0, 0, 0, 0}},
        {"all but every 8th (224 CUs)", {0xfefefefe, 0xfefefefe, 0xfefefefe, 0xfefefefe, 0xfefefefe, 0xfefefefe, 0xfefefefe, 0xfefefefe}},
    };
    const size_t n = (size_t)64 << 20;                 // 1 GiB of float4 = 16 B each
    float4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMemset(a, 1, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto copy_rate = [&](const char* name, hipStream_t s, int grid) {
        for (int r = 0; r < 2; ++r) {
            hipEventRecord(e0, s);
            hipLaunchKernelGGL(stream_copy, dim3(grid), dim3(256), 0, s, a, b, n);
            hipEventRecord(e1, s); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("   stream copy on %-28s grid %5d: %7.1f us  %6.0f GB/s (read + write)\n", name, grid, ms * 1e3, 2.0 * n * 16 / ms / 1e6);
    };
    copy_rate("unmasked", s0, 8192);
    for (auto& mk : masks) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mk.m);
        if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", mk.name, hipGetErrorString(e)); continue; }
        run_census(mk.name, s, 4096);
        copy_rate(mk.name, s, 8192);
        hipStreamDestroy(s);
    }
    return 0;
}
