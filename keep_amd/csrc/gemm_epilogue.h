// Fused GEMM epilogues shared by the fp16 MFMA GEMM kernels.
// A lane hands over 4 consecutive output columns n..n+3 of one output row m (fp32 accumulators).
#pragma once
#include "common.h"

namespace keepk {

// Branch-free exact-GELU for the fc1 epilogue (libm erff is ~40+ VALU ops with divergent branches and
// sat in the epilogue of the largest GEMM: 13k of 78k cycles per tile).  With a = min(|x|, 6):
//     Phi(-a) = 0.5 * erfc(a / sqrt2) = 2^-P(a)          P: degree-11 polynomial, P(0) = 1
//     gelu(x) = x * (x < 0 ? u : 1 - u),   u = 2^-P(|x|)
// One polynomial, one v_exp_f32, no cancellation for x < 0.  Fitted in tools/fit_gelu.py (weighted
// least squares in absolute-Phi error, fp32 Horner emulation): max |gelu_fast - gelu| = 3.8e-7 over
// [-8, 8] (the fp32 rounding of x*Phi itself); checked on the GPU against float64 erf in
// tests/test_ops_gpu.py::test_gelu_epilogue_accuracy_sweep.
__device__ __forceinline__ float gelu_fast(float x) {
    const float a = fminf(fabsf(x), 6.0f);
    float p = -3.594286202e-09f;
    p = fmaf(p, a, 1.257803746e-07f);
    p = fmaf(p, a, -1.958191660e-06f);
    p = fmaf(p, a, 1.778304431e-05f);
    p = fmaf(p, a, -1.016299357e-04f);
    p = fmaf(p, a, 3.363231954e-04f);
    p = fmaf(p, a, -7.651424676e-05f);
    p = fmaf(p, a, -6.888056640e-03f);
    p = fmaf(p, a, 5.241813138e-02f);
    p = fmaf(p, a, 4.592245221e-01f);
    p = fmaf(p, a, 1.151104093e+00f);
    p = fmaf(p, a, 1.0f);
    const float u = __builtin_amdgcn_exp2f(-p);
    return x * (x < 0.f ? u : 1.0f - u);
}

// Two elements at once on the packed fp32 pipe (v_pk_fma_f32): halves the polynomial's issue slots.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    const f32x2 a = __builtin_elementwise_min(__builtin_elementwise_abs(x), (f32x2){6.0f, 6.0f});
    f32x2 p = {-3.594286202e-09f, -3.594286202e-09f};
    constexpr float c[11] = {1.257803746e-07f, -1.958191660e-06f, 1.778304431e-05f, -1.016299357e-04f, 3.363231954e-04f,
                             -7.651424676e-05f, -6.888056640e-03f, 5.241813138e-02f, 4.592245221e-01f, 1.151104093e+00f, 1.0f};
#pragma unroll
    for (int i = 0; i < 11; ++i) p = __builtin_elementwise_fma(p, a, (f32x2){c[i], c[i]});
    // x * (x < 0 ? u : 1 - u) == max(x, 0) - |x| * u; with a = min(|x|, 6) in place of |x| the difference is < 6e-9
    const f32x2 u = {__builtin_amdgcn_exp2f(-p[0]), __builtin_amdgcn_exp2f(-p[1])};
    const f32x2 m = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    return __builtin_elementwise_fma(-a, u, m);
}

// Degree-6 variant for the single-pass fp16 mode: max |err| 4.8e-7 absolute, 2.7e-5 relative to
// max(|gelu|, 1e-2) -- an order of magnitude under the fp16 rounding (2.4e-4) applied right after it.
__device__ __forceinline__ f32x2 gelu_fast2_fp16(f32x2 x) {
    const f32x2 a = __builtin_elementwise_min(__builtin_elementwise_abs(x), (f32x2){6.0f, 6.0f});
    f32x2 p = {-3.068802471e-05f, -3.068802471e-05f};
    constexpr float c[6] = {7.369693485e-04f, -7.944388315e-03f, 5.316609517e-02f, 4.589701593e-01f, 1.151135921e+00f, 9.999994040e-01f};
#pragma unroll
    for (int i = 0; i < 6; ++i) p = __builtin_elementwise_fma(p, a, (f32x2){c[i], c[i]});
    // x * (x < 0 ? u : 1 - u) == max(x, 0) - |x| * u; with a = min(|x|, 6) in place of |x| the difference is < 6e-9
    const f32x2 u = {__builtin_amdgcn_exp2f(-p[0]), __builtin_amdgcn_exp2f(-p[1])};
    const f32x2 m = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    return __builtin_elementwise_fma(-a, u, m);
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

}  // namespace keepk
