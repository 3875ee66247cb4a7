// Semantics probe for ds_read_b64_tr_b16 (gfx950): which LDS elements does lane L receive, given each lane's address?
// Measured (ROCm 7.2, MI355X): the wave is four independent groups of 16 lanes; inside a group every lane supplies the address of
// 4 consecutive 16-bit elements, and lane i receives, as its element e, element (i % 4) of source lane (4 e + i / 4) -- a 4x4
// transpose between 4-lane quads and element slots.  With lane s of group g pointing at V[k0 + 4 g + s / 4][d0 + 4 (s % 4) ..+3] of a
// row-major [key][feature] image, lane i gets V[k0 + 4 g + 0..3][d0 + i]: four consecutive keys of ONE feature, i.e. the V^T fragment
// of the P.V MFMA without a transposed copy of V in LDS (not used by attention.hip yet: its V^T image is built through registers).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_probe.hip -o tools/ubench/tr_probe && tools/ubench/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(int pattern, float* out) {
    __shared__ _Float16 lds[32 * 64];
    for (int i = threadIdx.x; i < 32 * 64; i += 64) lds[i] = (_Float16)(float)i;       // value = row * 64 + col (exact up to 2048)
    __syncthreads();
    const int l = threadIdx.x;
    int row, col;
    if (pattern == 0) { row = l % 16; col = 4 * (l / 16); }            // 16 rows, each lane 4 consecutive columns
    else if (pattern == 1) { row = l / 4; col = 4 * (l % 4); }         // lane-linear over a [16][16] block
    else { row = (l % 4) + 4 * (l / 16); col = 4 * ((l / 4) % 4); }
    h16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((h16x4 __attribute__((address_space(3)))*)(lds + row * 64 + col));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (float)v[e];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4 * 4);
    float h[256];
    for (int p = 0; p < 3; ++p) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, d);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("pattern %d: lane -> (row,col) of the 4 elements it received\n", p);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int e = 0; e < 4; ++e) printf(" (%2d,%2d)", (int)h[l * 4 + e] / 64, (int)h[l * 4 + e] % 64);
            if (l % 2) printf("\n");
        }
    }
    return 0;
}
