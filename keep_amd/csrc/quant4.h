// MX-fp4 (e2m1 elements, one E8M0 scale per 32 k) side planes of a GEMM operand: the inputs of the compensation phase
// of the 256x256 GEMM (gemm_f16_v2.hip, "phase 2").
//
// An fp16 GEMM operand X carries 11 significant bits; the reference (quick_start/keep_inference.py:54-62) computes in
// fp32.  With X = X_hi + X_lo (X_hi = fp16(X)) the product A W^T = A_hi W_hi^T + A_hi W_lo^T + A_lo W_hi^T (+ 2^-22 terms).
// The first term is the fp16 MFMA pass; the two correction terms are 2^-11 of it, so 2-3 significant bits are enough for
// them and they run on the block-scaled fp4 pipe (4x the fp16 MFMA rate: v_mfma_scale_f32_32x32x64_f8f6f4), i.e. a
// compensated GEMM costs ~1.5 fp16 passes instead of 3 (tools/precision_study.py: same end-to-end error as exact lo parts).
//
// Layout of one operand X[R][K] (R padded to 256 rows, K % 64 == 0), KT = K / 32:
//   q  : [R/256][KT][2 planes: 0 = Q(X_hi), 1 = Q(X_lo)][256 rows][16 B]   16 B = 32 e2m1 nibbles, k 2b in the low
//        nibble of byte b.  One K slice of one row tile is 8 KiB contiguous, a phase-2 chunk (K = 64) 16 KiB.
//   sc : [R/256][KT][2 planes][256 B]   one E8M0 byte per row, permuted so that the four row tiles a wave owns sit in ONE
//        dword: byte ((r / 128) * 32 + r % 32) * 4 + (r % 128) / 32  holds the scale of row r  (r = row % 256)
// The MFMA's fragment is 32 consecutive k of one row per lane, which is exactly one 16-byte element group with its
// scale; since a 32-block never straddles lanes the k order inside a block is the natural one.
#pragma once
#include "common.h"

namespace keepk {

__host__ __device__ __forceinline__ int64_t q4_data_off(int row, int kt, int plane, int KT) {
    return ((int64_t)(row >> 8) * KT + kt) * 8192 + plane * 4096 + ((row & 255) << 4);
}
__host__ __device__ __forceinline__ int64_t q4_scale_off(int row, int kt, int plane, int KT) {
    const int r = row & 255;
    return ((int64_t)(row >> 8) * KT + kt) * 512 + plane * 256 + ((((r >> 7) << 5) + (r & 31)) << 2) + ((r & 127) >> 5);
}
static inline size_t q4_data_bytes(int64_t R, int64_t K) { return (size_t)((R + 255) / 256 * 256) * (size_t)K; }          // both planes
static inline size_t q4_scale_bytes(int64_t R, int64_t K) { return (size_t)((R + 255) / 256 * 256) * (size_t)(K / 32) * 2; }

typedef _Float16 q4_f16x2 __attribute__((ext_vector_type(2)));

// E8M0 exponent byte for a block whose largest magnitude is `amax`: the smallest power of two s with amax / s <= 6
// (6 = largest e2m1 value), so the block maximum never clips.  Returns the biased exponent (scale = 2^(e - 127)).
__device__ __forceinline__ unsigned q4_block_exponent(float amax) {
    unsigned e = (__float_as_uint(amax * (1.0f / 6.0f)) + 0x007fffffu) >> 23;      // ceil to a power of two
    e = e < 1u ? 1u : (e > 254u ? 254u : e);
    return e;
}

// 8 consecutive k of one row (a quarter of a 32-block; the 4 lanes lane&~3 .. lane|3 hold the block) -> 4 bytes of e2m1.
// `e_out` receives the block's E8M0 byte (identical in the 4 lanes).
__device__ __forceinline__ unsigned q4_quantize8(const f16x8& v, unsigned& e_out) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) m = fmaxf(m, fabsf((float)v[i]));
    m = fmaxf(m, __shfl_xor(m, 1));
    m = fmaxf(m, __shfl_xor(m, 2));
    const unsigned e = q4_block_exponent(m);
    const float scale = __uint_as_float(e << 23);
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, q4_f16x2{v[0], v[1]}, scale, 0);     // RNE(v / scale), first source in the low nibble
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, q4_f16x2{v[2], v[3]}, scale, 1);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, q4_f16x2{v[4], v[5]}, scale, 2);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(w, q4_f16x2{v[6], v[7]}, scale, 3);
    e_out = e;
    return w;
}

// Store the hi / lo planes of 8 consecutive k (k0 = kt * 32 + quarter * 8) of row `row`; the calling lanes must be
// arranged so that lanes 4j .. 4j+3 hold quarters 0..3 of one (row, kt).
__device__ __forceinline__ void q4_store8(unsigned char* q, unsigned char* sc, int KT, int row, int kt, int quarter,
                                          const f16x8& hi, const f16x8& lo) {
    unsigned eh, el;
    const unsigned wh = q4_quantize8(hi, eh);
    const unsigned wl = q4_quantize8(lo, el);
    *reinterpret_cast<unsigned*>(q + q4_data_off(row, kt, 0, KT) + quarter * 4) = wh;
    *reinterpret_cast<unsigned*>(q + q4_data_off(row, kt, 1, KT) + quarter * 4) = wl;
    if (quarter == 0) sc[q4_scale_off(row, kt, 0, KT)] = (unsigned char)eh;
    if (quarter == 1) sc[q4_scale_off(row, kt, 1, KT)] = (unsigned char)el;
}

// The same for the hi plane alone (one-term compensation: the consumer reads Q(X_hi) only; the lo plane of the block is left untouched).
__device__ __forceinline__ void q4_store8_hi(unsigned char* q, unsigned char* sc, int KT, int row, int kt, int quarter, const f16x8& hi) {
    unsigned eh;
    const unsigned wh = q4_quantize8(hi, eh);
    *reinterpret_cast<unsigned*>(q + q4_data_off(row, kt, 0, KT) + quarter * 4) = wh;
    if (quarter == 0) sc[q4_scale_off(row, kt, 0, KT)] = (unsigned char)eh;
}

}  // namespace keepk
