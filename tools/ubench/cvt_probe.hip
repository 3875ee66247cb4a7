// Semantics probe for v_cvt_scalef32_pk_fp4_{f16,f32} (gfx950): scale direction, nibble order, byte select, rounding, saturation.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/cvt_probe.hip -o tools/ubench/cvt_probe && tools/ubench/cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, int n, float scale, unsigned* out16, unsigned* out32) {
    const int i = threadIdx.x;
    if (i >= n) return;
    f16x2 h = {(_Float16)in[2 * i], (_Float16)in[2 * i + 1]};
    out16[i] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(0xAAAAAAAAu, h, scale, 1);
    out32[i] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0x55555555u, in[2 * i], in[2 * i + 1], scale, 2);
}
int main() {
    const float vals[] = {0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f, -6.0f, 0.24f, 0.26f, 0.75f, 1.25f, 1.75f, 2.5f, 3.5f, 5.0f, 7.0f, 100.0f, -0.3f, 0.0f, 2.4f, 2.6f};
    const int n = sizeof(vals) / sizeof(float) / 2;
    float* din; unsigned *d16, *d32;
    hipMalloc(&din, sizeof(vals)); hipMalloc(&d16, 64 * 4); hipMalloc(&d32, 64 * 4);
    hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
    const float E[8] = {0, .5f, 1, 1.5f, 2, 3, 4, 6};
    for (float scale : {1.0f, 2.0f, 0.5f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, n, scale, d16, d32);
        unsigned h16[64], h32[64];
        hipMemcpy(h16, d16, 64 * 4, hipMemcpyDeviceToHost); hipMemcpy(h32, d32, 64 * 4, hipMemcpyDeviceToHost);
        printf("scale %.1f\n", scale);
        for (int i = 0; i < n; ++i) {
            const unsigned b16 = (h16[i] >> 8) & 0xff, b32 = (h32[i] >> 16) & 0xff;
            auto dec = [&](unsigned nib) { return ((nib & 8) ? -1.f : 1.f) * E[nib & 7]; };
            printf("  (%7.3f, %7.3f) f16: word %08x -> lo nibble %5.2f hi nibble %5.2f | f32: word %08x -> lo %5.2f hi %5.2f\n", vals[2 * i], vals[2 * i + 1],
                   h16[i], dec(b16 & 15), dec(b16 >> 4), h32[i], dec(b32 & 15), dec(b32 >> 4));
        }
    }
    return 0;
}
