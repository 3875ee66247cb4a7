// Layout / rate probe for the block-scaled MX MFMA of gfx950 (v_mfma_scale_f32_32x32x64_f8f6f4) with fp4 (e2m1) operands.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mx_probe.hip -o tools/ubench/mx_probe && tools/ubench/mx_probe
//
// Hypothesis under test (what the GEMM's second phase assumes):
//   A operand, lane l: row i = l % 32, k = 32 * (l / 32) + [0..31], 16 bytes = 32 e2m1 nibbles, low nibble of byte b = k 2b,
//   high nibble = k 2b+1; B operand the same with row -> column; one E8M0 scale per lane (byte `opsel` of the scale VGPR)
//   covering that lane's 32 k; C/D as every 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static const float E2M1[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
static float dec4(unsigned nib) { const float v = E2M1[nib & 7]; return (nib & 8) ? -v : v; }

__global__ void probe(const uint4* a, const uint4* b, const unsigned* sa, const unsigned* sb, f32x16* c) {
    const int l = threadIdx.x;
    const uint4 av = a[l], bv = b[l];
    v8i A = {(int)av.x, (int)av.y, (int)av.z, (int)av.w, 0, 0, 0, 0};
    v8i B = {(int)bv.x, (int)bv.y, (int)bv.z, (int)bv.w, 0, 0, 0, 0};
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 1, (int)sa[l], 2, (int)sb[l]);   // opsel: byte 1 of sa, byte 2 of sb
    c[l] = acc;
}

// rate: NW waves per SIMD each issuing a dependent-free chain of MX fp4 MFMAs
__global__ __launch_bounds__(512) void rate_fp4(float* out, int iters) {
    v8i A = {(int)threadIdx.x, 0x12345678, 0x22222222, 0x31313131, 0, 0, 0, 0}, B = {0x11111111, (int)threadIdx.x, 0x43434343, 0x25252525, 0, 0, 0, 0};
    f32x16 acc[4] = {};
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[j], 4, 4, 0, 127, 0, 127);
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void rate_f16(float* out, int iters) {
    f16x8 A, B;
    for (int e = 0; e < 8; ++e) { A[e] = (_Float16)(threadIdx.x * 0.001f + e); B[e] = (_Float16)(e * 0.5f - threadIdx.x * 0.002f); }
    f32x16 acc[4] = {};
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[j], 0, 0, 0);
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    srand(1);
    std::vector<unsigned char> A(32 * 64), B(32 * 64);                 // nibble codes, A[i][k], B[j][k]
    std::vector<int> ea(32 * 2), eb(32 * 2);                           // E8M0 exponents per (row, k block)
    for (auto& v : A) v = rand() & 15;
    for (auto& v : B) v = rand() & 15;
    for (auto& v : ea) v = 120 + rand() % 12;
    for (auto& v : eb) v = 118 + rand() % 12;
    std::vector<uint4> ha(64), hb(64);
    std::vector<unsigned> hsa(64), hsb(64);
    for (int l = 0; l < 64; ++l) {
        const int r = l % 32, kb = l / 32;
        unsigned wa[4] = {0, 0, 0, 0}, wb[4] = {0, 0, 0, 0};
        for (int k = 0; k < 32; ++k) {
            wa[k / 8] |= (unsigned)A[r * 64 + kb * 32 + k] << (4 * (k % 8));
            wb[k / 8] |= (unsigned)B[r * 64 + kb * 32 + k] << (4 * (k % 8));
        }
        ha[l] = {wa[0], wa[1], wa[2], wa[3]}; hb[l] = {wb[0], wb[1], wb[2], wb[3]};
        hsa[l] = 0x11000033u | ((unsigned)ea[r * 2 + kb] << 8);         // the scale sits in byte 1 (opsel 1); other bytes are decoys
        hsb[l] = 0x55005522u | ((unsigned)eb[r * 2 + kb] << 16);        // byte 2 (opsel 2)
    }
    uint4 *da, *db; unsigned *dsa, *dsb; f32x16* dc;
    hipMalloc(&da, 64 * 16); hipMalloc(&db, 64 * 16); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 64 * 64);
    hipMemcpy(da, ha.data(), 64 * 16, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 64 * 16, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
    std::vector<float> hc(64 * 16);
    hipMemcpy(hc.data(), dc, 64 * 64, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int l = 0; l < 64; ++l)
        for (int reg = 0; reg < 16; ++reg) {
            const int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
            double ref = 0;
            for (int k = 0; k < 64; ++k)
                ref += (double)dec4(A[row * 64 + k]) * std::ldexp(1.0, ea[row * 2 + k / 32] - 127) *
                       (double)dec4(B[col * 64 + k]) * std::ldexp(1.0, eb[col * 2 + k / 32] - 127);
            maxerr = std::fmax(maxerr, std::fabs(ref - hc[l * 16 + reg]));
            maxref = std::fmax(maxref, std::fabs(ref));
        }
    printf("mx fp4 32x32x64 layout probe: max|err| %.3e (max|ref| %.3e) -> %s\n", maxerr, maxref, maxerr <= 1e-5 * maxref ? "LAYOUT OK" : "LAYOUT MISMATCH");

    float* dout; hipMalloc(&dout, 256 * 8 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        const int iters = 4000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(rate_fp4, dim3(256 * 4), dim3(512), 0, 0, dout, iters);
            else hipLaunchKernelGGL(rate_f16, dim3(256 * 4), dim3(512), 0, 0, dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 32 * 32 * (which == 0 ? 64 : 16) * 4.0 * iters * 8 * 256 * 4;
        printf("%s: %.1f TFLOP/s\n", which == 0 ? "mx fp4 32x32x64" : "f16 32x32x16", flop / (ms * 1e-3) / 1e12);
    }
    return 0;
}
