// Fused GEMM epilogues shared by the fp16 MFMA GEMM kernels.
// A lane hands over 4 consecutive output columns n..n+3 of one output row m (fp32 accumulators).
#pragma once
#include "common.h"

namespace keepk {

// Branch-free single-precision erf (libm erff is ~40+ VALU ops with divergent branches, and it sits in
// the epilogue of the largest GEMM).  Two Chebyshev-fitted pieces evaluated unconditionally and selected:
//   |x| <= 0.921875 : erf(x) = x * P6(x^2)
//   |x| >  0.921875 : erf(x) = sign(x) * (1 - exp(-Q9(min(|x|, 4))))      Q(t) = -log(erfc(t))
// max |erf_fast - erf| = 1.13e-7 over [-6, 6] (fit + fp32 Horner emulation in tools/fit_erf.py); the GELU
// built on it is checked against float64 erf in tests/test_ops_gpu.py.
__device__ __forceinline__ float erf_fast(float x) {
    const float t = fminf(fabsf(x), 4.0f);
    const float s = x * x;
    float p = 8.392696327064186e-05f;
    p = fmaf(p, s, -0.0008148506167344749f);
    p = fmaf(p, s, 0.005201591644436121f);
    p = fmaf(p, s, -0.026859646663069725f);
    p = fmaf(p, s, 0.11283700168132782f);
    p = fmaf(p, s, -0.37612634897232056f);
    p = fmaf(p, s, 1.128379225730896f);
    const float r1 = p * x;
    float q = 3.990486874272392e-08f;
    q = fmaf(q, t, -2.4969324385892833e-06f);
    q = fmaf(q, t, 5.4058713431004435e-05f);
    q = fmaf(q, t, -0.0006388962501659989f);
    q = fmaf(q, t, 0.00489716324955225f);
    q = fmaf(q, t, -0.02670992538332939f);
    q = fmaf(q, t, 0.11046823859214783f);
    q = fmaf(q, t, 0.631505012512207f);
    q = fmaf(q, t, 1.1303812265396118f);
    q = fmaf(q, t, -0.00034889878588728607f);
    const float r2 = copysignf(1.0f - __expf(-q), x);
    return fabsf(x) <= 0.921875f ? r1 : r2;
}
__device__ __forceinline__ float gelu_fast(float x) {
    return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f));
}

template <int EPI>
__device__ __forceinline__ void gemm_epilogue_row(const GemmParams& p, int m, int& prow, int64_t& orow) {
    orow = m; prow = 0;
    if (EPI == EPI_PATCH) {
        const int b = m / p.patches_per_img;
        prow = m - b * p.patches_per_img + 1;
        orow = (int64_t)b * (p.patches_per_img + 1) + prow;
    }
}

template <int EPI>
__device__ __forceinline__ void gemm_epilogue_store(const GemmParams& p, int64_t orow, int prow, int n, f32x4 v) {
    const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += bias[e];
    const int64_t o = orow * p.N + n;
    if (EPI == EPI_F16 || EPI == EPI_GELU_F16) {
        if (EPI == EPI_GELU_F16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
        }
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { f16 hh, ll; split_f16(v[e], hh, ll); h[e] = hh; l[e] = ll; }
        *reinterpret_cast<f16x4*>(p.out_hi + o) = h;
        if (p.out_lo) *reinterpret_cast<f16x4*>(p.out_lo + o) = l;
    } else if (EPI == EPI_RESID_LS) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.ls + n);
        f32x4 r = *reinterpret_cast<const f32x4*>(p.resid + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] += g[e] * v[e];
        *reinterpret_cast<f32x4*>(p.resid + o) = r;
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

// Batched form: one output row m, NV column groups (4 consecutive n each).  Every global load the
// epilogue needs (bias, LayerScale, residual / pos-embed rows) is issued BEFORE the first dependent
// store, so a wave pays one memory round trip per row-batch instead of one per 16-byte group (the
// compiler cannot hoist loads above the may-alias stores itself; that serialisation measured ~30 us
// per 256x256 tile).
template <int EPI, int NV>
__device__ __forceinline__ void gemm_epilogue_batch(const GemmParams& p, int64_t orow, int prow, const int (&n)[NV], f32x4 (&v)[NV]) {
    f32x4 bias[NV], aux[NV], res[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        bias[q] = *reinterpret_cast<const f32x4*>(p.bias + n[q]);
        if (EPI == EPI_RESID_LS) {
            aux[q] = *reinterpret_cast<const f32x4*>(p.ls + n[q]);
            res[q] = *reinterpret_cast<const f32x4*>(p.resid + orow * p.N + n[q]);
        } else if (EPI == EPI_PATCH) {
            res[q] = *reinterpret_cast<const f32x4*>(p.pos + (int64_t)prow * p.N + n[q]);
        } else if (EPI == EPI_RESID_F32) {
            res[q] = *reinterpret_cast<const f32x4*>(p.resid + orow * p.N + n[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int64_t o = orow * p.N + n[q];
        f32x4 x = v[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] += bias[q][e];
        if (EPI == EPI_F16 || EPI == EPI_GELU_F16) {
            if (EPI == EPI_GELU_F16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = gelu_fast(x[e]);
            }
            f16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) { f16 hh, ll; split_f16(x[e], hh, ll); h[e] = hh; l[e] = ll; }
            *reinterpret_cast<f16x4*>(p.out_hi + o) = h;
            if (p.out_lo) *reinterpret_cast<f16x4*>(p.out_lo + o) = l;
        } else if (EPI == EPI_RESID_LS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = res[q][e] + aux[q][e] * x[e];
            *reinterpret_cast<f32x4*>(p.resid + o) = x;
        } else if (EPI == EPI_PATCH) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] += res[q][e];
            *reinterpret_cast<f32x4*>(p.resid + o) = x;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] += res[q][e];
            *reinterpret_cast<f32x4*>(p.out_f32 + o) = x;
        }
    }
}

}  // namespace keepk
