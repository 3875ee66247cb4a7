// Where does the gap between the matrix pipes' power ceiling (tools/ubench/mfma_power.hip: ~1 600-1 700 TFLOP/s on high-entropy fp16 operands) and the K loop of the
// 256x256 GEMM (~1 380 TFLOP/s) go?  The same 8-wave / 2-waves-per-SIMD MFMA stream as mfma_power.hip, with the K loop's other work added one piece at a time:
//   mode 0   MFMAs only, 2 A x 4 B fragments held in registers (the round-4 probe)
//   mode 1   + the fragments of every group of 8 MFMAs re-read from LDS (6 x ds_read_b128 per 8 MFMAs: the 8-wave kernel's ratio, 96 KiB per K step and CU),
//              fresh high-entropy data each time (the LDS holds 64 KiB of it, the read window moves every iteration)
//   mode 2   the 4-wave kernel's ratio (4 reads per 8 MFMAs: 64 KiB per K step and CU)
//   mode 3   mode 1 + the LDS-DMA stream at the K loop's rate (2 x global_load_lds_dwordx4 per wave and 8 MFMAs = 32 KiB per K step and CU) from a region the
//              L2 holds, into the part of the LDS nobody reads (so the MFMAs' operands stay what mode 1 feeds them)
//   mode 4   the same stream from a region no L2 holds (Infinity Cache / HBM)
// No barrier, no s_waitcnt on the DMA (vmcnt is drained every 16 iterations only): this is an energy probe, not a GEMM.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power_ladder.hip -o /tmp/mfma_power_ladder && /tmp/mfma_power_ladder
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE>
__global__ __launch_bounds__(512, 2) void burn(const f16x8* __restrict__ src, const char* __restrict__ stream, float* __restrict__ out, int iters, long long* clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];          // 64 KiB of operand data + 32 KiB DMA landing zone
    f16x8* lds = reinterpret_cast<f16x8*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x * 512 + tid;
    for (int i = tid; i < 4096; i += 512) lds[i] = src[(size_t)((blockIdx.x * 4096 + i) % (256 * 512 * 8))];
    __syncthreads();
    f16x8 a[2], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (i < 2) a[i] = src[(size_t)t * 8 + i]; b[i] = src[(size_t)t * 8 + 4 + i]; }
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x16{};
    const char* sbase = stream + (size_t)blockIdx.x * (MODE == 3 ? 65536u : (4u << 20));
    const long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    f16x8 a2[2] = {a[0], a[1]}, b2[4] = {b[0], b[1], b[2], b[3]};
    // two half-iterations per trip: the fragments one half computes on were fetched during the other (ping-pong register sets, no copies)
    auto half = [&](int it, f16x8 (&ac)[2], f16x8 (&bc)[4], f16x8 (&an)[2], f16x8 (&bn)[4]) {
        if (MODE >= 1) {
            const int base = ((it * 7 + wave * 61) & 63) * 64 + lane;      // a 1 KiB-aligned window of the 64 KiB: conflict-free b128 reads, different data every time
            constexpr int NR = MODE == 2 ? 4 : 6;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const f16x8 v = lds[(base + r * 64 * 9) & 4095];
                if (r < 2) an[r] = v; else bn[r - 2] = v;
            }
        }
        if (MODE >= 3) {
            // 16 KiB per workgroup and half-iteration (= 32 KiB per K step of 16 MFMAs per wave, the K loop's rate): every wave moves its own 2 x 1 KiB;
            // MODE 3: a 64 KiB region per workgroup, walked round and round (2 MiB per XCD: L2 hits); MODE 4: a 4 MiB region per workgroup (128 MiB per XCD: every load comes from
            // the Infinity Cache / HBM)
            unsigned off = (MODE == 3 ? (unsigned)(it & 3) * 16384u : (unsigned)(it & 255) * 16384u) + (unsigned)wave * 2048u + (unsigned)lane * 16u;
            asm volatile("" : "+v"(off));
#pragma unroll
            for (int r = 0; r < 2; ++r)
                __builtin_amdgcn_global_load_lds((gptr_t)(sbase + off + r * 1024), (lptr_t)(smem + 65536 + (wave * 2 + r) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac[i], bc[j], acc[i * 4 + j], 0, 0, 0);
    };
    for (int it = 0; it < iters; it += 2) {
        half(it, a, b, a2, b2);
        half(it + 1, a2, b2, a, b);
        if (MODE >= 3 && (it & 14) == 14) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[t] = s;
    if (tid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE> void run(const char* what, const f16x8* src, const char* stream, float* out, long long* clk, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&burn<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(burn<MODE>, dim3(256), dim3(512), 98304, 0, src, stream, out, iters, clk);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
        const double flop = 256.0 * 8 * iters * 8 * 2.0 * 32 * 32 * 16;
        printf("%-78s %8.2f ms  %7.0f TFLOP/s  (%.3f of 2516.6)   shader clock %5.0f MHz   cycles per 8 MFMAs and wave pair %.0f\n", what, ms, flop / ms / 1e9,
               flop / ms / 1e9 / 2516.6, 100.0 * (double)c[0] / (double)c[1], (double)c[0] / iters);
    }
}
int main() {
    const int iters = 200000;
    const size_t nfrag = (size_t)256 * 512 * 8;
    std::vector<_Float16> h(nfrag * 8);
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) {
        float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
        h[i] = (_Float16)(sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2) * 0.05f);
    }
    f16x8* src; char* stream; float* out; long long* clk;
    hipMalloc(&src, nfrag * 16); hipMalloc(&stream, (size_t)1024 << 20); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 16);
    hipMemcpy(src, h.data(), nfrag * 16, hipMemcpyHostToDevice);
    for (size_t o = 0; o < ((size_t)1024 << 20); o += nfrag * 16) hipMemcpy(stream + o, h.data(), nfrag * 16, hipMemcpyHostToDevice);
    run<0>("MFMAs only (fragments in registers)", src, stream, out, clk, iters);
    run<1>("+ fragments re-read from LDS, 6 ds_read_b128 per 8 MFMAs (8-wave kernel's ratio)", src, stream, out, clk, iters);
    run<2>("+ fragments re-read from LDS, 4 ds_read_b128 per 8 MFMAs (4-wave kernel's ratio)", src, stream, out, clk, iters);
    run<3>("+ 6 reads per 8 MFMAs + the LDS-DMA stream at the K loop's rate, from the L2", src, stream, out, clk, iters);
    run<4>("+ 6 reads per 8 MFMAs + the LDS-DMA stream at the K loop's rate, from beyond the L2", src, stream, out, clk, iters);
    return 0;
}
