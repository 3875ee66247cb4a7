// Small-M fp16 MFMA GEMM (split-K) for latency-bound calls: one prompt (M = T rows), one tile (M = 197), the
// CLS-only tail of the last ViT block (M = batch).
//
// The 256x256 LDS-DMA kernel (gemm_f16_v2.hip) gives such a call ceil(M/256) * N/256 workgroups -- 3 to 16 of the 256
// CUs -- each walking the whole K dimension alone (K = 3072: 96 steps, ~60 us); the reference's WSI scripts make
// thousands of these calls (1320-7128 encode_text calls at P = 1).  Here the work is cut the
// other way: 32 x 128 output tiles (one 32x32 MFMA tile per wave, fragments loaded straight from the K-blocked
// operands into registers; no LDS, the four waves share their A rows through the L1) times S slices of K, S picked
// so that ~768 workgroups exist.  Every slice writes its fp32 partial tile; a second kernel adds the S partials in a
// fixed order (bit-reproducible, no atomics) and applies the same fused epilogues as the big kernel.
#include "gemm_epilogue.h"

namespace keepk {

constexpr int SK_BM = 32, SK_BN = 128;

// acc layout of v_mfma_f32_32x32x16_f16 with (W fragment, A fragment) operands, as in gemm_f16_v2.hip: lane owns
// output row m = lane & 31 and, in register group rg = 0..3, the four consecutive columns n = 8 rg + 4 (lane >> 5) + 0..3.
__global__ __launch_bounds__(256)
void gemm_skinny_partial_kernel(GemmParams p, float* __restrict__ part, int steps_per_split) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * SK_BN + wave * 32, m0 = blockIdx.y * SK_BM, split = blockIdx.z;
    const int KT = p.K / 32;
    const int kt0 = split * steps_per_split;
    const int kt1 = min(KT, kt0 + steps_per_split);
    const int r = lane & 31, half = lane >> 5;
    // element offsets of this lane's 8-element (16 B) fragment inside a 256 x 32 K-slice of the blk layout
    const int m = m0 + r, n = n0 + r;
    const int64_t a_base = (int64_t)(m >> 8) * KT * 8192 + ((m & 255) << 5) + half * 8;
    const int64_t w_base = (int64_t)(n >> 8) * KT * 8192 + ((n & 255) << 5) + half * 8;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int seg = 0; seg < p.nseg; ++seg) {
        const f16* A = (seg == 1 ? p.a_lo : p.a_hi) + a_base;
        const f16* W = (seg == 2 ? p.w_lo : p.w_hi) + w_base;
        int kt = kt0;
        // two K-slices (4 MFMAs) per iteration: all eight 16-byte loads are in flight before the first MFMA
        for (; kt + 2 <= kt1; kt += 2) {
            const f16x8 a0 = *reinterpret_cast<const f16x8*>(A + (int64_t)kt * 8192);
            const f16x8 w0 = *reinterpret_cast<const f16x8*>(W + (int64_t)kt * 8192);
            const f16x8 a1 = *reinterpret_cast<const f16x8*>(A + (int64_t)kt * 8192 + 16);
            const f16x8 w1 = *reinterpret_cast<const f16x8*>(W + (int64_t)kt * 8192 + 16);
            const f16x8 a2 = *reinterpret_cast<const f16x8*>(A + (int64_t)(kt + 1) * 8192);
            const f16x8 w2 = *reinterpret_cast<const f16x8*>(W + (int64_t)(kt + 1) * 8192);
            const f16x8 a3 = *reinterpret_cast<const f16x8*>(A + (int64_t)(kt + 1) * 8192 + 16);
            const f16x8 w3 = *reinterpret_cast<const f16x8*>(W + (int64_t)(kt + 1) * 8192 + 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, a0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, a1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, a2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3, a3, acc, 0, 0, 0);
        }
        for (; kt < kt1; ++kt) {
            const f16x8 a0 = *reinterpret_cast<const f16x8*>(A + (int64_t)kt * 8192);
            const f16x8 w0 = *reinterpret_cast<const f16x8*>(W + (int64_t)kt * 8192);
            const f16x8 a1 = *reinterpret_cast<const f16x8*>(A + (int64_t)kt * 8192 + 16);
            const f16x8 w1 = *reinterpret_cast<const f16x8*>(W + (int64_t)kt * 8192 + 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, a0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, a1, acc, 0, 0, 0);
        }
    }
    if (m < p.M) {
        float* o = part + ((int64_t)split * p.M + m) * p.N + n0 + half * 4;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[rg * 4 + e];
            *reinterpret_cast<f32x4*>(o + rg * 8) = v;
        }
    }
}

// The same product for 64 <= M (the CLS-row chain of the image tower: M = tiles of a lane, three operand segments): 128 x 128 output tiles, one
// 64 x 64 quadrant (2 x 2 MFMA tiles) per wave.  A wave of the kernel above loads one A and one W fragment per MFMA -- 16 FLOP per operand byte, all of
// it L2 traffic (28 us for [256, 1024] x [1024, 4096] x 3 segments: the L2, not the matrix pipes) --; here every fragment feeds two MFMAs.
constexpr int SK2_BM = 128, SK2_BN = 128;

__global__ __launch_bounds__(256)
void gemm_skinny_partial_wide_kernel(GemmParams p, float* __restrict__ part, int steps_per_split) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * SK2_BN + wn * 64, m0 = blockIdx.y * SK2_BM + wm * 64, split = blockIdx.z;
    const int KT = p.K / 32;
    const int kt0 = split * steps_per_split;
    const int kt1 = min(KT, kt0 + steps_per_split);
    const int r = lane & 31, half = lane >> 5;
    int64_t a_base[2], w_base[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int m = m0 + f * 32 + r, n = n0 + f * 32 + r;       // (rows past M lie inside the operand's 256-row padding: computed, never stored)
        a_base[f] = (int64_t)(m >> 8) * KT * 8192 + ((m & 255) << 5) + half * 8;
        w_base[f] = (int64_t)(n >> 8) * KT * 8192 + ((n & 255) << 5) + half * 8;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // one flat walk over (segment, K slice) with the NEXT step's eight 16-byte fragments in flight under this step's eight MFMAs: a workgroup of this
    // kernel is latency-bound (a few dozen dependent steps, each a round trip to the L2 / HBM), not bandwidth- or pipe-bound
    const int nk = kt1 - kt0, steps = nk > 0 ? nk * p.nseg : 0;
    auto frag = [&](int st, f16x8 (&a)[2][2], f16x8 (&w)[2][2]) {
        const int seg = st / nk, kt = kt0 + (st - seg * nk);
        const f16* A = seg == 1 ? p.a_lo : p.a_hi;
        const f16* W = seg == 2 ? p.w_lo : p.w_hi;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                a[f][k] = *reinterpret_cast<const f16x8*>(A + a_base[f] + (int64_t)kt * 8192 + k * 16);
                w[f][k] = *reinterpret_cast<const f16x8*>(W + w_base[f] + (int64_t)kt * 8192 + k * 16);
            }
    };
    auto fma8 = [&](const f16x8 (&a)[2][2], const f16x8 (&w)[2][2]) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[j][k], a[i][k], acc[i][j], 0, 0, 0);
    };
    f16x8 a0[2][2], w0[2][2], a1[2][2], w1[2][2];
    if (steps > 0) frag(0, a0, w0);
    int st = 0;
    for (; st + 2 <= steps; st += 2) {
        frag(st + 1, a1, w1);
        fma8(a0, w0);
        if (st + 2 < steps) frag(st + 2, a0, w0);
        fma8(a1, w1);
    }
    if (st < steps) fma8(a0, w0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + i * 32 + r;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* o = part + ((int64_t)split * p.M + m) * p.N + n0 + j * 32 + half * 4;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rg * 4 + e];
                *reinterpret_cast<f32x4*>(o + rg * 8) = v;
            }
        }
    }
}

// Sum of the S partials (s = 0, 1, ... in that order) + the fused epilogue of gemm_f16_v2.hip.  One thread per 4 columns.
template <int EPI>
__global__ __launch_bounds__(256)
void gemm_skinny_reduce_kernel(GemmParams p, const float* __restrict__ part, int S) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int nq = p.N / 4;
    if (idx >= (int64_t)p.M * nq) return;
    const int m = (int)(idx / nq), n = (int)(idx - (int64_t)m * nq) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(part + (int64_t)m * p.N + n);
    for (int s = 1; s < S; ++s) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(part + ((int64_t)s * p.M + m) * p.N + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += x[e];
    }
    const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += bias[e];
    int prow; int64_t orow;
    gemm_epilogue_row<EPI>(p, m, prow, orow);
    const int64_t o = orow * p.N + n;
    if (EPI == EPI_F16 || EPI == EPI_GELU_F16) {
        if (EPI == EPI_GELU_F16) {
            f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
            if (p.out_lo) { a = gelu_fast2(a); b = gelu_fast2(b); }
            else { a = gelu_fast2_fp16(a); b = gelu_fast2_fp16(b); }
            v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
        }
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { f16 hh, ll; split_f16(v[e], hh, ll); h[e] = hh; l[e] = ll; }
        const int64_t oo = p.out_kt > 0 ? blk_off((int)orow, n, p.out_kt) : o;
        *reinterpret_cast<f16x4*>(p.out_hi + oo) = h;
        if (p.out_lo) *reinterpret_cast<f16x4*>(p.out_lo + oo) = l;
    } else if (EPI == EPI_RESID_LS) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.ls + n);
        f32x4 r = *reinterpret_cast<const f32x4*>(p.resid + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] += __fmul_rn(g[e], v[e]);     // LayerScale product rounded on its own, as the 256x256 kernel does: torch's x + gamma * y on every path
        *reinterpret_cast<f32x4*>(p.resid + o) = r;
        if (p.resid_copy) *reinterpret_cast<f32x4*>(p.resid_copy + orow * p.resid_copy_ld + n) = r;
    } else if (EPI == EPI_PATCH) {
        const f32x4 pe = *reinterpret_cast<const f32x4*>(p.pos + (int64_t)prow * p.N + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += pe[e];
        *reinterpret_cast<f32x4*>(p.resid + o) = v;
    } else {   // EPI_RESID_F32
        const f32x4 r = *reinterpret_cast<const f32x4*>(p.resid + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r[e];
        *reinterpret_cast<f32x4*>(p.out_f32 + o) = v;
    }
}

// Same sum + residual epilogue for a WHOLE output row per workgroup (N <= 1024: one thread per 4 columns), followed by
// the LayerNorm that consumes the row next (two-pass statistics over the row held in registers): saves the
// LayerNorm launch and its read of the row on the latency path.
template <int EPI>
__global__ __launch_bounds__(256)
void gemm_skinny_reduce_ln_kernel(GemmParams p, const float* __restrict__ part, int S) {
    __shared__ float red[2][4];
    const int m = blockIdx.x, t = threadIdx.x, n = t * 4;
    const bool on = n < p.N;
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (on) {
        f32x4 v = *reinterpret_cast<const f32x4*>(part + (int64_t)m * p.N + n);
        for (int s = 1; s < S; ++s) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(part + ((int64_t)s * p.M + m) * p.N + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += x[e];
        }
        const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + n);
        const int64_t o = (int64_t)m * p.N + n;
        r = *reinterpret_cast<const f32x4*>(p.resid + o);
        if (EPI == EPI_RESID_LS) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.ls + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] += __fmul_rn(g[e], v[e] + bias[e]);     // no fused multiply-add (same rounding as the 256x256 kernel's epilogue)
            *reinterpret_cast<f32x4*>(p.resid + o) = r;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = (v[e] + bias[e]) + r[e];
            if (p.ln_out_f32 != p.out_f32) *reinterpret_cast<f32x4*>(p.out_f32 + o) = r;
        }
    }
    const int wave = t >> 6, lane = t & 63;
    float sum = wave_sum(r[0] + r[1] + r[2] + r[3]);
    if (lane == 0) red[0][wave] = sum;
    __syncthreads();
    const float mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)p.N;
    float sq = 0.f;
    if (on) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { r[e] -= mean; sq += r[e] * r[e]; }
    }
    sq = wave_sum(sq);
    if (lane == 0) red[1][wave] = sq;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)p.N + p.ln_eps);
    if (!on) return;
    const f32x4 g = *reinterpret_cast<const f32x4*>(p.ln_gamma + n);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(p.ln_beta + n);
    f32x4 y; f16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        y[e] = r[e] * rstd * g[e] + bt[e];
        f16 hh, ll; split_f16(y[e], hh, ll); h[e] = hh; l[e] = ll;
    }
    if (p.ln_out_f32) *reinterpret_cast<f32x4*>(p.ln_out_f32 + (int64_t)m * p.N + n) = y;
    const int64_t oo = blk_off(m, n, p.N / 32);
    *reinterpret_cast<f16x4*>(p.ln_out_hi + oo) = h;
    if (p.ln_out_lo) *reinterpret_cast<f16x4*>(p.ln_out_lo + oo) = l;
}

}  // namespace keepk
using namespace keepk;

bool skinny_wide(int M, const KeepTune* t = nullptr) { return M >= 64 && (!t || t->skinny_wide); }       // the 128 x 128 kernel from 64 rows up (below, its half-empty row blocks cost more than the fragment reuse saves)

int skinny_splits(int M, int N, int K, const KeepTune* t = nullptr) {
    const bool wide = skinny_wide(M, t);
    const int tiles = wide ? ((M + SK2_BM - 1) / SK2_BM) * (N / SK2_BN) : ((M + SK_BM - 1) / SK_BM) * (N / SK_BN), ks = K / 32;
    int S = ((wide ? 256 : 768) + tiles - 1) / tiles;      // (wide: one 128 x 128 tile per CU -- fewer, longer K walks and less partial-sum traffic for the reduce)
    if (S > ks / 2) S = ks / 2;
    if (S > 16) S = 16;
    return S < 1 ? 1 : S;
}

size_t skinny_ws_bytes(int M, int N, int K) { return (size_t)skinny_splits(M, N, K) * M * N * sizeof(float); }

// Returns -1 when the shape is not eligible (caller falls back to the big kernel), else 0 or GEMM_DID_LN.
int launch_gemm_f16_skinny(const GemmParams& p, int epi, float* ws, size_t ws_bytes, hipStream_t s) {
    if (!ws || p.N % SK_BN || p.K % 32 || p.M < 1) return -1;
    const int S = skinny_splits(p.M, p.N, p.K, p.tune);
    if ((size_t)S * p.M * p.N * sizeof(float) > ws_bytes) return -1;
    const int KT = p.K / 32, per = (KT + S - 1) / S;
    dim3 block(256);
    if (skinny_wide(p.M, p.tune)) hipLaunchKernelGGL(gemm_skinny_partial_wide_kernel, dim3(p.N / SK2_BN, (p.M + SK2_BM - 1) / SK2_BM, S), block, 0, s, p, ws, per);
    else hipLaunchKernelGGL(gemm_skinny_partial_kernel, dim3(p.N / SK_BN, (p.M + SK_BM - 1) / SK_BM, S), block, 0, s, p, ws, per);
    return launch_gemm_splitk_reduce(p, epi, ws, S, s);
}

// Sum of S fp32 partial planes ws[s][M][N] + epilogue (shared by the small-M path above and the mid-size split-K
// path of the 256x256 kernel).
int launch_gemm_splitk_reduce(const GemmParams& p, int epi, const float* ws, int S, hipStream_t s) {
    dim3 block(256);
    if (p.ln_gamma && p.ln_out_hi && (epi == EPI_RESID_LS || epi == EPI_RESID_F32) && p.N <= 1024) {
        if (epi == EPI_RESID_LS) hipLaunchKernelGGL(gemm_skinny_reduce_ln_kernel<EPI_RESID_LS>, dim3(p.M), block, 0, s, p, ws, S);
        else hipLaunchKernelGGL(gemm_skinny_reduce_ln_kernel<EPI_RESID_F32>, dim3(p.M), block, 0, s, p, ws, S);
        return GEMM_DID_LN;
    }
    const int64_t items = (int64_t)p.M * (p.N / 4);
    dim3 rg((unsigned)((items + 255) / 256));
    switch (epi) {
        case EPI_F16:       hipLaunchKernelGGL(gemm_skinny_reduce_kernel<EPI_F16>, rg, block, 0, s, p, ws, S); break;
        case EPI_GELU_F16:  hipLaunchKernelGGL(gemm_skinny_reduce_kernel<EPI_GELU_F16>, rg, block, 0, s, p, ws, S); break;
        case EPI_RESID_LS:  hipLaunchKernelGGL(gemm_skinny_reduce_kernel<EPI_RESID_LS>, rg, block, 0, s, p, ws, S); break;
        case EPI_PATCH:     hipLaunchKernelGGL(gemm_skinny_reduce_kernel<EPI_PATCH>, rg, block, 0, s, p, ws, S); break;
        default:            hipLaunchKernelGGL(gemm_skinny_reduce_kernel<EPI_RESID_F32>, rg, block, 0, s, p, ws, S); break;
    }
    return 0;
}
