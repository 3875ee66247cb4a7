// Exact-fp32 "NT" GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
//   out[m][n] = act(scale * sum_k A[m][k] * B[n][k] + bias[n])
//
// Used where the reference runs small fp32 matmuls whose precision feeds straight into the output:
// visual_head (keep_inference.py:42-46), BERT pooler (tanh), and the tile x prompt similarity
// `img_feature @ text_feature.T` (keep_inference.py:104) / `image_features @ cls`
// (WSI_evaluation/utils.py:128).  The f32 MFMA is an exact fmaf chain (k-ordered), so results match
// an fp32 CPU matmul to summation-order rounding.
//
// Tile: 128x128x16 per 256-thread workgroup (2x2 waves, each 64x64 = 2x2 MFMA 32x32 tiles).
// LDS holds both operands k-major ([16][129] floats) so the one-float-per-lane MFMA operands are
// conflict-free ds_read_b32; 4 lanes cover one 64-byte row segment on the global side.
#include "common.h"

namespace keepk {

constexpr int SB = 128, SK = 16, SLD = 129;

__global__ __launch_bounds__(256)
void sgemm_f32_nt_kernel(SgemmParams p) {
    __shared__ float sA[SK * SLD];
    __shared__ float sB[SK * SLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * SB, n0 = blockIdx.x * SB;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lrow = tid >> 2;          // 0..63
    const int lk = (tid & 3) * 4;       // 0,4,8,12
    const int fi = lane & 31, fk = lane >> 5;

    for (int k0 = 0; k0 < p.K; k0 += SK) {
        f32x4 ra[2], rb[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int am = m0 + lrow + 64 * h; am = am < p.M ? am : p.M - 1;
            int bn = n0 + lrow + 64 * h; bn = bn < p.N ? bn : p.N - 1;
            ra[h] = *reinterpret_cast<const f32x4*>(p.a + (int64_t)am * p.lda + k0 + lk);
            rb[h] = *reinterpret_cast<const f32x4*>(p.b + (int64_t)bn * p.ldb + k0 + lk);
        }
        __syncthreads();                 // previous tile fully consumed
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sA[(lk + e) * SLD + lrow + 64 * h] = ra[h][e];
                sB[(lk + e) * SLD + lrow + 64 * h] = rb[h][e];
            }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < SK; kk += 2) {
            float fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = sA[(kk + fk) * SLD + wm * 64 + i * 32 + fi];
                fb[i] = sB[(kk + fk) * SLD + wn * 64 + i * 32 + fi];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    // D[i][j]: lane holds j(n) = lane&31, i(m) = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + fi;
            if (n >= p.N) continue;
            const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (m >= p.M) continue;
                float v = acc[i][j][r] * p.scale + bias;
                if (p.act == ACT_GELU) v = gelu_erf(v);
                else if (p.act == ACT_TANH) v = tanhf(v);
                p.out[(int64_t)m * p.ldo + n] = v;
            }
        }
}

// Few-row variant (M <= 16: one tile's projection head, one prompt's pooler).  The MFMA kernel above gives such a call
// N/128 workgroups that each walk K alone (76 us for the pooler of one prompt); here one wave owns one output column,
// keeps that B row in registers and reduces across lanes -- N waves, ~3 us.  fp32 FMA throughout.
constexpr int GEMV_MAX_M = 16, GEMV_MAX_K = 1024;
__global__ __launch_bounds__(256)
void sgemv_f32_nt_kernel(SgemmParams p) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= p.N) return;
    constexpr int KV = GEMV_MAX_K / 256;             // float4 per lane
    f32x4 b[KV];
    const float* brow = p.b + (int64_t)n * p.ldb;
#pragma unroll
    for (int i = 0; i < KV; ++i) {
        const int k = (i * 64 + lane) * 4;
        b[i] = k < p.K ? *reinterpret_cast<const f32x4*>(brow + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int m = 0; m < p.M; ++m) {
        const float* arow = p.a + (int64_t)m * p.lda;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < KV; ++i) {
            const int k = (i * 64 + lane) * 4;
            if (k < p.K) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(arow + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = fmaf(a[e], b[i][e], acc);
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            float v = acc * p.scale + (p.bias ? p.bias[n] : 0.f);
            if (p.act == ACT_GELU) v = gelu_erf(v);
            else if (p.act == ACT_TANH) v = tanhf(v);
            p.out[(int64_t)m * p.ldo + n] = v;
        }
    }
}

}  // namespace keepk

int launch_sgemm_f32(const SgemmParams& p, hipStream_t s) {
    const int g_sgemv_m = p.tune ? p.tune->sgemv_m : keepk::GEMV_MAX_M;      // rows up to which the few-row kernel is used (0: never)
    if (p.K % keepk::SK != 0 || p.M < 1 || p.N < 1) return -1;
    if ((p.lda % 4) || (p.ldb % 4)) return -1;
    if (p.M <= g_sgemv_m && p.K <= keepk::GEMV_MAX_K && p.K % 4 == 0) {
        hipLaunchKernelGGL(keepk::sgemv_f32_nt_kernel, dim3((p.N + 3) / 4), dim3(256), 0, s, p);
        return 0;
    }
    dim3 grid((p.N + keepk::SB - 1) / keepk::SB, (p.M + keepk::SB - 1) / keepk::SB);
    hipLaunchKernelGGL(keepk::sgemm_f32_nt_kernel, grid, dim3(256), 0, s, p);
    return 0;
}
