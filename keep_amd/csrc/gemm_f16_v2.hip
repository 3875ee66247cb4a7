// fp16 MFMA GEMM, large-tile variant: 256 x BN x 64 tiles, 8 wavefronts, direct global->LDS DMA.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]   (same contract and epilogues as gemm_f16.hip)
//
// Structure (MI355X guide: "glds, 2 LDS buffers, BK=64, one barrier per K tile"):
//   * workgroup = 512 threads = 8 waves, WM x WN; each wave owns (256/WM) x (BN/WN) outputs as
//     32x32x16 MFMA tiles (operand swap as in v1: lanes hold 4 consecutive n of one m)
//   * both operands are staged with global_load_lds_dwordx4 (16 B per lane straight into LDS, no
//     VGPR round trip).  The DMA writes LDS linearly (wave base + lane*16), so the XOR swizzle that
//     makes ds_read_b128 conflict-free is applied to the per-lane GLOBAL source address instead:
//     LDS slot (row, s) holds logical 16-B chunk s ^ ((row>>1)&7) of that row
//   * two LDS buffers (2 x (256 + BN) x 128 B); the DMA for K-tile t+1 is issued before the MFMAs
//     of tile t and drained (vmcnt(0)) at the single barrier that ends the step
//   * nseg == 3: hi/lo split product through the same accumulators (strict precision)
#include "gemm_epilogue.h"

namespace keepk {

constexpr int V2_BM = 256, V2_BK = 64, V2_THREADS = 512;

__device__ __forceinline__ int v2_lds_off(int row, int chunk) {
    return row * V2_BK + ((chunk ^ ((row >> 1) & 7)) << 3);
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(V2_THREADS, 2)
void gemm_f16_v2_kernel(GemmParams p) {
    constexpr int BM = V2_BM, BK = V2_BK;
    constexpr int TM = BM / WM / 32;            // MFMA tiles per wave along m
    constexpr int TN = BN / WN / 32;            // along n
    constexpr int A_ROUNDS = BM * 8 / V2_THREADS;   // 16-B slots per thread per K tile
    constexpr int B_ROUNDS = BN * 8 / V2_THREADS;
    constexpr int BUF_ELEMS = (BM + BN) * BK;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f16* lds = reinterpret_cast<f16*>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // tile coordinates: consecutive workgroups share the A panel (same m tile, different n tile)
    const int ntn = p.N / BN;
    const int bid = blockIdx.x;
    const int m0 = (bid / ntn) * BM;
    const int n0 = (bid % ntn) * BN;

    // ---- DMA source offsets (elements) per round; LDS slot L = round*512 + tid -> row L>>3, slot L&7
    int64_t a_off[A_ROUNDS], w_off[B_ROUNDS];
#pragma unroll
    for (int r = 0; r < A_ROUNDS; ++r) {
        const int L = r * V2_THREADS + tid;
        const int row = L >> 3, c = (L & 7) ^ ((row >> 1) & 7);
        int am = m0 + row; am = am < p.M ? am : p.M - 1;
        a_off[r] = (int64_t)am * p.K + c * 8;
    }
#pragma unroll
    for (int r = 0; r < B_ROUNDS; ++r) {
        const int L = r * V2_THREADS + tid;
        const int row = L >> 3, c = (L & 7) ^ ((row >> 1) & 7);
        w_off[r] = (int64_t)(n0 + row) * p.K + c * 8;
    }

    const int ktiles = p.K / BK;
    const int steps = ktiles * p.nseg;

    auto stage = [&](int s, int buf) {
        const int seg = s / ktiles;
        const int kk = (s - seg * ktiles) * BK;
        const f16* ab = ((seg == 1) ? p.a_lo : p.a_hi) + kk;
        const f16* wb = ((seg == 2) ? p.w_lo : p.w_hi) + kk;
        f16* sa = lds + buf * BUF_ELEMS;
        f16* sw = sa + BM * BK;
#pragma unroll
        for (int r = 0; r < A_ROUNDS; ++r)
            __builtin_amdgcn_global_load_lds((gptr_t)(ab + a_off[r]), (lptr_t)(sa + (r * V2_THREADS + wave * 64) * 8), 16, 0, 0);
#pragma unroll
        for (int r = 0; r < B_ROUNDS; ++r)
            __builtin_amdgcn_global_load_lds((gptr_t)(wb + w_off[r]), (lptr_t)(sw + (r * V2_THREADS + wave * 64) * 8), 16, 0, 0);
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    stage(0, 0);
    __syncthreads();

    const int frow = lane & 31, fhi = lane >> 5;
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < steps) stage(s + 1, buf ^ 1);
        const f16* sa = lds + buf * BUF_ELEMS;
        const f16* sw = sa + BM * BK;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 fw[TN], fa[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i)
                fw[i] = *reinterpret_cast<const f16x8*>(sw + v2_lds_off(wn * (TN * 32) + i * 32 + frow, ks * 2 + fhi));
#pragma unroll
            for (int j = 0; j < TM; ++j)
                fa[j] = *reinterpret_cast<const f16x8*>(sa + v2_lds_off(wm * (TM * 32) + j * 32 + frow, ks * 2 + fhi));
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();      // drains the DMA of tile s+1 (vmcnt(0)) and fences the reads of tile s
    }

#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * (TM * 32) + j * 32 + frow;
        if (m >= p.M) continue;
        int prow; int64_t orow;
        gemm_epilogue_row<EPI>(p, m, prow, orow);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int n = n0 + wn * (TN * 32) + i * 32 + 8 * rg + 4 * fhi;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rg * 4 + e];
                gemm_epilogue_store<EPI>(p, orow, prow, n, v);
            }
    }
}

template <int BN, int WM, int WN>
int launch_v2(const GemmParams& p, int epi, hipStream_t s) {
    constexpr size_t lds_bytes = 2 * (size_t)(V2_BM + BN) * V2_BK * sizeof(f16);
    static bool attr_set = false;
    if (!attr_set) {
#define KEEP_SET_ATTR(E) if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_v2_kernel<BN, WM, WN, E>), \
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return -2;
        KEEP_SET_ATTR(EPI_F16) KEEP_SET_ATTR(EPI_GELU_F16) KEEP_SET_ATTR(EPI_RESID_LS) KEEP_SET_ATTR(EPI_PATCH) KEEP_SET_ATTR(EPI_RESID_F32)
#undef KEEP_SET_ATTR
        attr_set = true;
    }
    const int grid = (p.N / BN) * ((p.M + V2_BM - 1) / V2_BM);
    dim3 g(grid), b(V2_THREADS);
    switch (epi) {
        case EPI_F16:      hipLaunchKernelGGL((gemm_f16_v2_kernel<BN, WM, WN, EPI_F16>), g, b, lds_bytes, s, p); break;
        case EPI_GELU_F16: hipLaunchKernelGGL((gemm_f16_v2_kernel<BN, WM, WN, EPI_GELU_F16>), g, b, lds_bytes, s, p); break;
        case EPI_RESID_LS: hipLaunchKernelGGL((gemm_f16_v2_kernel<BN, WM, WN, EPI_RESID_LS>), g, b, lds_bytes, s, p); break;
        case EPI_PATCH:    hipLaunchKernelGGL((gemm_f16_v2_kernel<BN, WM, WN, EPI_PATCH>), g, b, lds_bytes, s, p); break;
        default:           hipLaunchKernelGGL((gemm_f16_v2_kernel<BN, WM, WN, EPI_RESID_F32>), g, b, lds_bytes, s, p); break;
    }
    return 0;
}

}  // namespace keepk

// returns 0 if launched, 1 if the shape is not covered by this variant
int launch_gemm_f16_v2(const GemmParams& p, int epi, int variant, hipStream_t s) {
    using namespace keepk;
    if (p.K % V2_BK) return 1;
    if (variant == 256 && p.N % 256 == 0) return launch_v2<256, 2, 4>(p, epi, s);
    if (variant == 128 && p.N % 128 == 0) return launch_v2<128, 4, 2>(p, epi, s);
    return 1;
}
