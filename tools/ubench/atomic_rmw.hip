// Can the fp32 residual read-modify-write of the proj / fc2 epilogues (x += v, 206 MB per 128-tile launch) be handed to the L2's atomic units?
// A fire-and-forget global_atomic_add_f32 needs no load latency and no registers for the old value in the wave; what it costs in the memory system
// is what this measures.  Three forms over the same [rows, 1024] fp32 buffer, 64-column row segments per wave-instruction as in the GEMM epilogue:
//   rmw      : nontemporal load x4 -> add -> nontemporal store x4        (the epilogue today)
//   atomic   : 4 x global_atomic_add_f32 per lane, lane-contiguous dwords (each instruction covers one 256-byte row segment)
//   atomic16 : the same values, but lane l owns 4 consecutive floats (the 16-byte-per-lane layout of the rmw form): each instruction strides 16 B
//   store    : write-only (what the epilogue would cost if the residual add happened elsewhere)
// and, beside each, the same traffic while every CU also streams MFMA-free LDS-DMA loads (not modelled) -- run alone here.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench/atomic_rmw.hip -o tools/ubench/atomic_rmw && tools/ubench/atomic_rmw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave handles `rows_per_wave` rows of a 64-column slab (the epilogue's unit): 16 lanes x 16 B per row, 4 rows per instruction
template <int MODE>
__global__ __launch_bounds__(512) void k(float* __restrict__ x, const float* __restrict__ v, int rows, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slabs = N / 64;
    const long wid = (long)blockIdx.x * 8 + wave;            // wave -> (row block of 32, slab)
    const long nwaves = (long)(rows / 32) * slabs;
    for (long w = wid; w < nwaves; w += (long)gridDim.x * 8) {
        const int rb = (int)(w / slabs), sl = (int)(w % slabs);
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = rb * 32 + it * 4 + (lane >> 4);
                const long o = (long)r * N + sl * 64 + (lane & 15) * 4;
                f32x4 a = *reinterpret_cast<const f32x4*>(v + o);
                if (MODE == 0) {
                    f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + o));
                    a += b;
                }
                __builtin_nontemporal_store(a, reinterpret_cast<f32x4*>(x + o));
            }
        } else if (MODE == 1) {
            // lane-contiguous dwords: instruction j of row r covers columns [64 sl, 64 sl + 64)
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const long o = (long)(rb * 32 + r) * N + sl * 64 + lane;
                __hip_atomic_fetch_add(x + o, v[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = rb * 32 + it * 4 + (lane >> 4);
                const long o = (long)r * N + sl * 64 + (lane & 15) * 4;
                const f32x4 a = *reinterpret_cast<const f32x4*>(v + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) __hip_atomic_fetch_add(x + o + e, a[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

int main() {
    const int rows = 25216 / 32 * 32, N = 1024;          // one 128-tile lane's residual stream
    const size_t n = (size_t)rows * N;
    float *x, *v;
    hipMalloc(&x, n * 4); hipMalloc(&v, n * 4);
    std::vector<float> hx(n), hv(n);
    for (size_t i = 0; i < n; ++i) { hx[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f; hv[i] = (float)((i * 40503u) % 777) * 1e-4f; }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char* names[4] = {"rmw (nt load + add + nt store, 16 B per lane)", "atomic add f32, lane-contiguous dwords", "atomic add f32, 4 per lane (16-B lane stride)", "store only (16 B per lane)"};
    for (int grid : {256, 512, 1024}) {
        for (int mode = 0; mode < 4; ++mode) {
            hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(v, hv.data(), n * 4, hipMemcpyHostToDevice);
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(a, 0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, x, v, rows, N);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, x, v, rows, N);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, x, v, rows, N);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(512), 0, 0, x, v, rows, N);
                hipEventRecord(b, 0); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep && ms < best) best = ms;
            }
            // check one pass: x after 6 passes = hx + 6 hv for modes 0-2 (fp32 adds in the same order: bit-equal), = hv for mode 3
            std::vector<float> out(4096);
            hipMemcpy(out.data(), x + 12345 * 64, 4096 * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int i = 0; i < 4096; ++i) {
                float want = hx[12345 * 64 + i];
                for (int rpt = 0; rpt < 6; ++rpt) want = want + hv[12345 * 64 + i];
                if (mode == 3) want = hv[12345 * 64 + i];
                if (out[i] != want) ++bad;
            }
            const double bytes = (mode == 3 ? 2.0 : 3.0) * n * 4;      // v read + x read + x written (store-only: v read + x written)
            printf("grid %4d  %-50s %8.1f us  %7.1f GB/s (x traffic alone: %7.1f GB/s)  mismatches %d\n", grid, names[mode], best * 1e3, bytes / best / 1e6,
                   (mode == 3 ? 1.0 : 2.0) * n * 4 / best / 1e6, bad);
        }
    }
    return 0;
}
