// fp16-operand / fp32-accumulate MFMA GEMM with fused epilogues (gfx950).
//
//   C[m][n] = sum_k A[m][k] * W[n][k]          A: activations [M][K], W: torch Linear weight [N][K]
//
// This is the arithmetic behind every nn.Linear / the patch-embed conv of the two
// towers (timm Block: qkv, proj, fc1, fc2 -- SURVEY.md §A.1; HF BertLayer: q/k/v,
// attention.output.dense, intermediate.dense, output.dense -- §A.2).
//
// Tiling (v1, "128x128x64 / 4 waves"):
//   * workgroup = 256 threads = 4 wavefronts (2 along M x 2 along N), tile 128x128, BK = 64
//   * each wave owns 64x64 = 2x2 MFMA 32x32x16 tiles (64 fp32 accumulators / lane)
//   * MFMA operand swap: the "A" operand carries W rows (n) and the "B" operand activation rows
//     (m), so a lane's C fragment is 4 consecutive n of ONE m -> 8/16-byte epilogue stores
//   * global -> VGPR -> LDS staging, LDS double-buffered, one barrier per K step
//   * LDS rows are 128 B; the 16-B slot index is XORed with (row>>1)&7 which makes both the
//     ds_write_b128 (8-lane groups) and the ds_read_b128 (16-lane groups) conflict free
//   * nseg == 3 runs the hi/lo split product  A_hi*W_hi + A_lo*W_hi + A_hi*W_lo  through the same
//     accumulators (strict precision mode)
#include "gemm_epilogue.h"

#ifdef KEEP_EXPERIMENTS
namespace keepk {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int THREADS = 256;
constexpr int TILE_ELEMS = BM * BK;            // 8192 f16 = 16 KiB per operand per buffer

__device__ __forceinline__ int lds_off(int row, int chunk) {
    // element offset of 16-byte chunk `chunk` (0..7) of tile row `row` (0..127)
    return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3);
}

template <int EPI>
__global__ __launch_bounds__(THREADS, 2)
void gemm_f16_nt_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) f16 lds[2 * 2 * TILE_ELEMS];   // [buf][A|W][128*64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * BN;
    const int m0 = blockIdx.y * BM;

    // ---- staging coordinates: 4 chunks of 16 B per thread per operand
    int st_row[4], st_chunk[4];
    const f16* ga[4]; const f16* gw[4];      // row base pointers (hi plane), advanced by k
    int64_t a_row_off[4], w_row_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + THREADS * i;
        st_row[i] = q >> 3;
        st_chunk[i] = q & 7;
        int am = m0 + st_row[i]; if (am > p.M - 1) am = p.M - 1;     // clamp: rows >= M are never stored
        a_row_off[i] = (int64_t)am * p.K + st_chunk[i] * 8;
        w_row_off[i] = (int64_t)(n0 + st_row[i]) * p.K + st_chunk[i] * 8;
    }

    const int ktiles = p.K / BK;
    const int steps = ktiles * p.nseg;

    f16x8 ra[4], rw[4];
    auto load_global = [&](int s) {
        const int seg = s / ktiles;
        const int kk = (s - seg * ktiles) * BK;
        const f16* ab = (seg == 1) ? p.a_lo : p.a_hi;
        const f16* wb = (seg == 2) ? p.w_lo : p.w_hi;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const f16x8*>(ab + a_row_off[i] + kk);
            rw[i] = *reinterpret_cast<const f16x8*>(wb + w_row_off[i] + kk);
        }
    };
    auto store_lds = [&](int buf) {
        f16* sa = lds + buf * 2 * TILE_ELEMS;
        f16* sw = sa + TILE_ELEMS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = lds_off(st_row[i], st_chunk[i]);
            *reinterpret_cast<f16x8*>(sa + o) = ra[i];
            *reinterpret_cast<f16x8*>(sw + o) = rw[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_global(0);
    store_lds(0);
    __syncthreads();

    const int frow = lane & 31;
    const int fhi = lane >> 5;
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < steps) load_global(s + 1);
        const f16* sa = lds + buf * 2 * TILE_ELEMS;
        const f16* sw = sa + TILE_ELEMS;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 fw[2], fa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fw[i] = *reinterpret_cast<const f16x8*>(sw + lds_off(wn * 64 + i * 32 + frow, ks * 2 + fhi));
                fa[i] = *reinterpret_cast<const f16x8*>(sa + lds_off(wm * 64 + i * 32 + frow, ks * 2 + fhi));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < steps) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  acc[i][j][r]: n = n0 + wn*64 + i*32 + (r&3) + 8*(r>>2) + 4*fhi ; m = m0 + wm*64 + j*32 + frow
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + wm * 64 + j * 32 + frow;
        if (m >= p.M) continue;
        int prow; int64_t orow;
        gemm_epilogue_row<EPI>(p, m, prow, orow);
        int nn[8];
        f32x4 vv[8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                nn[i * 4 + rg] = n0 + wn * 64 + i * 32 + 8 * rg + 4 * fhi;
#pragma unroll
                for (int e = 0; e < 4; ++e) vv[i * 4 + rg][e] = acc[i][j][rg * 4 + e];
            }
        gemm_epilogue_batch<EPI, 8>(p, orow, prow, nn, vv);
    }
}

}  // namespace keepk
using namespace keepk;
#endif


int launch_gemm_f16_v2(const GemmParams& p, int epi, int variant, hipStream_t s);
int launch_gemm_f16_v3(const GemmParams& p, int epi, hipStream_t s);
int launch_gemm_f16(const GemmParams& p_in, int epi, hipStream_t s) {
    // Operands in blk layout -> the LDS-DMA kernels (gemm_f16_v2.hip); tune->gemm_impl only picks the tile width.
    static const KeepTune defaults;
    GemmParams p = p_in;
    const KeepTune& t = p.tune ? *p.tune : defaults;
#ifdef KEEP_DIAGNOSTICS
    p.ablate = t.gemm_ablate;
    p.dbg = t.dbg;
#else
    p.ablate = 0;
    p.dbg = nullptr;
#endif
    int impl = t.gemm_impl;
    if (epi == EPI_TOP2) return launch_gemm_f16_v2(p, epi, 256, s) == 0 ? 0 : -1;      // fused prompt screening: 256x256 kernel only
    if (p.comp) {                                    // compensated product: always the 256x256 kernel (callers route small M through nseg = 3)
        return launch_gemm_f16_v2(p, epi, 256, s) == 0 ? 0 : -1;
    }
    if (impl == 0 && p.M <= t.gemm_skinny_m && p.M <= SKINNY_MAX_M && p.splitk_ws) {
        const int rc = launch_gemm_f16_skinny(p, epi, p.splitk_ws, p.splitk_bytes, s);
        if (rc >= 0) return rc;
    }
    if (impl == 0 && t.gemm_splitk_tiles > 0 && p.splitk_ws && p.N % 256 == 0 && p.K % 32 == 0) {
        // Between the small-M kernel and a full machine: ceil(M/256) * N/256 tiles on 256 CUs, each walking all of K alone.
        // Cut K into S slices (>= 8 steps each), fp32 partials, then the shared reduce + epilogue kernel.
        const int tiles = ((p.M + 255) / 256) * (p.N / 256), KT = p.K / 32;
        int S = tiles < t.gemm_splitk_tiles ? 256 / tiles : 1;
        if (S > 8) S = 8;
        if (S > KT / 8) S = KT / 8;
        if (S >= 2 && (size_t)S * p.M * p.N * sizeof(float) <= p.splitk_bytes) {
            GemmParams q = p;
            q.ksplit = S;
            if (launch_gemm_f16_v2(q, EPI_PARTIAL, 256, s) == 0) return launch_gemm_splitk_reduce(p, epi, p.splitk_ws, S, s);
        }
    }
#ifdef KEEP_EXPERIMENTS
    if (impl == 3 && launch_gemm_f16_v3(p, epi, s) == 0) return 0;     // persistent 256x256 variant
    if (impl == 2128 || impl == 3256 || impl == 4256 || impl == 5256) { if (launch_gemm_f16_v2(p, epi, impl, s) == 0) return 0; }
#endif
    if ((impl != 128 && impl != 256) || (impl == 256 && p.N % 256)) impl = (p.N % 256 == 0) ? 256 : 128;
    return launch_gemm_f16_v2(p, epi, impl, s) == 0 ? 0 : -1;      // (-1: shape not covered or the LDS opt-in was refused -- the callers report it)
}

#ifdef KEEP_EXPERIMENTS
// Row-major operands: the register-staged 128x128 kernel above (cross-check variant for the op tests).
void launch_gemm_f16_rowmajor(const GemmParams& p, int epi, hipStream_t s) {
    dim3 grid(p.N / BN, (p.M + BM - 1) / BM), block(THREADS);
    switch (epi) {
        case EPI_F16:       hipLaunchKernelGGL(gemm_f16_nt_kernel<EPI_F16>, grid, block, 0, s, p); break;
        case EPI_GELU_F16:  hipLaunchKernelGGL(gemm_f16_nt_kernel<EPI_GELU_F16>, grid, block, 0, s, p); break;
        case EPI_RESID_LS:  hipLaunchKernelGGL(gemm_f16_nt_kernel<EPI_RESID_LS>, grid, block, 0, s, p); break;
        case EPI_PATCH:     hipLaunchKernelGGL(gemm_f16_nt_kernel<EPI_PATCH>, grid, block, 0, s, p); break;
        default:            hipLaunchKernelGGL(gemm_f16_nt_kernel<EPI_RESID_F32>, grid, block, 0, s, p); break;
    }
}
#else
void launch_gemm_f16_rowmajor(const GemmParams&, int, hipStream_t) {}     // cross-check kernel: experiment builds only
#endif
