// Row-wise / elementwise kernels around the GEMMs (all HBM-bound; 16-byte accesses per lane,
// one 64-lane wavefront per row where a row reduction is involved).
//
//   layernorm        timm nn.LayerNorm(eps=1e-6) / HF BertLayerNorm(eps=1e-12)
//   im2col           the reshape that turns timm PatchEmbed's conv16/s16 into a GEMM
//   bert_embed_ln    HF BertEmbeddings: word + position + token_type -> LayerNorm
//   l2norm_rows      F.normalize(x, dim=-1)                      keep_inference.py:56,61
//   row_argmax / row_softmax / top2_score
//                    argmax of raw cosine, softmax(10*logits) (subtyping_utils.py:72) and
//                    rank_cls_score (WSI_evaluation/utils.py:107-117)
#include "common.h"
#include "quant4.h"

namespace keepk {

// ------------------------------------------------------------------ LayerNorm
template <int NV>   // NV float4 per lane: D = NV * 256
__global__ __launch_bounds__(256)
void layernorm_kernel(LnParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const float* x = p.x + (int64_t)row * p.x_stride;
    f32x4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = *reinterpret_cast<const f32x4*>(x + (i * 64 + lane) * 4);
        if (p.add) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.add + (int64_t)row * p.x_stride + (i * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] += a[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) sum += v[i][e];
    }
    const float mean = wave_sum(sum) * (1.0f / (NV * 256));
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][e] -= mean; sq += v[i][e] * v[i][e]; }
    const float var = wave_sum(sq) * (1.0f / (NV * 256));
    const float rstd = 1.0f / sqrtf(var + p.eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + col);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(p.beta + col);
        f32x4 y;
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[e] = v[i][e] * rstd * g[e] + bt[e];
            f16 hh, ll; split_f16(y[e], hh, ll); h[e] = hh; l[e] = ll;
        }
        if (p.out_f32) *reinterpret_cast<f32x4*>(p.out_f32 + (int64_t)row * p.out_f32_stride + col) = y;
        const int64_t o16 = p.out_kt > 0 ? blk_off(row, col, p.out_kt) : (int64_t)row * (NV * 256) + col;
        if (p.out_hi) *reinterpret_cast<f16x4*>(p.out_hi + o16) = h;
        if (p.out_lo) *reinterpret_cast<f16x4*>(p.out_lo + o16) = l;
    }
}

// LayerNorm with blk-layout fp16 output.  A row's D values are spread over D/32 K-slices that lie 16 KiB apart
// in the blk layout, so a wave that normalises one row can only store 64-byte pieces.  Here a workgroup
// normalises R rows (one per wave), parks the fp16 results in LDS and then writes, per K-slice, the R rows x 64 B
// that ARE contiguous in the blk layout.  R = 8 keeps the LDS footprint (17 KiB) small enough to share a CU with a
// GEMM workgroup of the other stream lane (R = 16 could not, and lost end to end what it gained in isolation).
template <int NV, int R>
__global__ __launch_bounds__(R * 64)
void layernorm_blk_kernel(LnParams p) {
    constexpr int D = NV * 256, PITCH = D + 32, KT = D / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char ln_smem[];
    f16* s_hi = reinterpret_cast<f16*>(ln_smem);
    f16* s_lo = s_hi + R * PITCH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * R;
    {
        const int lr = wave;
        const int row = row0 + lr;
        const int rc = row < p.rows ? row : p.rows - 1;
        const float* x = p.x + (int64_t)rc * p.x_stride;
        f32x4 v[NV];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + (i * 64 + lane) * 4));     // streamed once (+0.25 %; the fp16 stores below measured better WITHOUT the hint)
            if (p.add) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(p.add + (int64_t)rc * p.x_stride + (i * 64 + lane) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] += a[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += v[i][e];
        }
        const float mean = wave_sum(sum) * (1.0f / D);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[i][e] -= mean; sq += v[i][e] * v[i][e]; }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) * (1.0f / D) + p.eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * 4;
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + col);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(p.beta + col);
            f32x4 y; f16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = v[i][e] * rstd * g[e] + bt[e];
                f16 hh, ll; split_f16(y[e], hh, ll); h[e] = hh; l[e] = ll;
            }
            if (p.out_f32 && row < p.rows) *reinterpret_cast<f32x4*>(p.out_f32 + (int64_t)row * p.out_f32_stride + col) = y;
            *reinterpret_cast<f16x4*>(s_hi + lr * PITCH + col) = h;
            if (p.out_lo || (p.out_q && !p.out_q_hi_only)) *reinterpret_cast<f16x4*>(s_lo + lr * PITCH + col) = l;
        }
    }
    __syncthreads();
    // R rows x 4 sixteen-byte chunks per K-slice; a wave covers 64 / (4R) slices per pass
    constexpr int SL = 64 / (4 * R);
    const int orow = (lane >> 2) % R, ochunk = (lane & 3) * 8;
    const int row = row0 + orow;
    if (row < p.rows) {
        for (int kt = wave * SL + lane / (4 * R); kt < KT; kt += R * SL) {
            const int64_t dst = blk_off(row, kt * 32 + ochunk, KT);
            const f16x8 h8 = *reinterpret_cast<const f16x8*>(s_hi + orow * PITCH + kt * 32 + ochunk);
            *reinterpret_cast<f16x8*>(p.out_hi + dst) = h8;
            if (p.out_lo || (p.out_q && !p.out_q_hi_only)) {
                const f16x8 l8 = *reinterpret_cast<const f16x8*>(s_lo + orow * PITCH + kt * 32 + ochunk);
                if (p.out_lo) *reinterpret_cast<f16x8*>(p.out_lo + dst) = l8;
                // the four lanes lane&~3 .. lane|3 hold the four quarters of this (row, K slice): one MX block
                if (p.out_q) q4_store8(p.out_q, p.out_sc, KT, row, kt, lane & 3, h8, l8);
            } else if (p.out_q) {
                q4_store8_hi(p.out_q, p.out_sc, KT, row, kt, lane & 3, h8);          // one-term compensation downstream: Q(x_hi) only
            }
        }
    }
}

// ------------------------------------------------------------------ im2col for the patch embed
__device__ __forceinline__ float bf16_to_f32(unsigned short u) { return __uint_as_float((unsigned)u << 16); }

template <int DT>
__global__ __launch_bounds__(256)
void im2col_kernel(const void* __restrict__ pixels, int B, f16* __restrict__ out_hi, f16* __restrict__ out_lo) {
    // one work item = 8 consecutive pixels of one image row: (b, c, y, xc) with xc in [0,28)
    const int64_t total = (int64_t)B * 3 * 224 * 28;
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
        const int xc = (int)(it % 28);
        int64_t r = it / 28;
        const int y = (int)(r % 224); r /= 224;
        const int c = (int)(r % 3);
        const int b = (int)(r / 3);
        const int64_t src = (((int64_t)b * 3 + c) * 224 + y) * 224 + xc * 8;
        float v[8];
        if (DT == PIX_F32) {
            const f32x4 a = *reinterpret_cast<const f32x4*>((const float*)pixels + src);
            const f32x4 d = *reinterpret_cast<const f32x4*>((const float*)pixels + src + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = d[e]; }
        } else if (DT == PIX_F16) {
            const f16x8 a = *reinterpret_cast<const f16x8*>((const f16*)pixels + src);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
        } else {
            const uint4 a = *reinterpret_cast<const uint4*>((const unsigned short*)pixels + src);
            const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[2 * e] = bf16_to_f32((unsigned short)(w[e] & 0xffffu));
                v[2 * e + 1] = bf16_to_f32((unsigned short)(w[e] >> 16));
            }
        }
        const int py = y >> 4, ph = y & 15, px = xc >> 1, half = xc & 1;
        const int64_t dst = blk_off(b * 196 + py * 14 + px, c * 256 + ph * 16 + half * 8, 24);
        f16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) { f16 hh, ll; split_f16(v[e], hh, ll); h[e] = hh; l[e] = ll; }
        *reinterpret_cast<f16x8*>(out_hi + dst) = h;
        if (out_lo) *reinterpret_cast<f16x8*>(out_lo + dst) = l;
    }
}

// uint8 HWC tiles [B,224,224,3] (what a tile extractor / PIL delivers after resize + crop): ToTensor (/255) and
// Normalize(ImageNet mean/std) of the reference transform (keep_inference.py:91-92) are applied on the fly,
// in the same operation order as torchvision: (x / 255 - mean) / std in fp32.
__global__ __launch_bounds__(256)
void im2col_u8_kernel(const unsigned char* __restrict__ pixels, int B, f16* __restrict__ out_hi, f16* __restrict__ out_lo) {
    // one work item = 8 consecutive pixels (24 bytes) of one image row: (b, y, xc)
    const int64_t total = (int64_t)B * 224 * 28;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
        const int xc = (int)(it % 28);
        const int64_t r = it / 28;
        const int y = (int)(r % 224), b = (int)(r / 224);
        const unsigned char* src = pixels + (((int64_t)b * 224 + y) * 224 + xc * 8) * 3;
        const uint2 w0 = *reinterpret_cast<const uint2*>(src);          // 24 bytes = 3 x 8-byte loads (8-byte aligned: 24 | offset)
        const uint2 w1 = *reinterpret_cast<const uint2*>(src + 8);
        const uint2 w2 = *reinterpret_cast<const uint2*>(src + 16);
        const unsigned wd[6] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y};
        const int py = y >> 4, ph = y & 15, px = xc >> 1, half = xc & 1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f16x8 h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int byte = e * 3 + c;
                const float u = (float)((wd[byte >> 2] >> ((byte & 3) * 8)) & 0xffu);
                const float v = (u / 255.0f - mean[c]) / stdv[c];
                f16 hh, ll; split_f16(v, hh, ll); h[e] = hh; l[e] = ll;
            }
            const int64_t dst = blk_off(b * 196 + py * 14 + px, c * 256 + ph * 16 + half * 8, 24);
            *reinterpret_cast<f16x8*>(out_hi + dst) = h;
            if (out_lo) *reinterpret_cast<f16x8*>(out_lo + dst) = l;
        }
    }
}

__global__ void cls_init_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ resid,
                                int B, int D, int ntok) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * D) return;
    const int b = (int)(i / D), d = (int)(i % D);
    resid[(int64_t)b * ntok * D + d] = cls[d] + pos[d];
}

// ------------------------------------------------------------------ elementwise split (weight prep)
__global__ void split_f16_kernel(const float* __restrict__ src, f16* __restrict__ hi, f16* __restrict__ lo, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        f16 h, l; split_f16(src[i], h, l);
        hi[i] = h;
        if (lo) lo[i] = l;
    }
}

// ------------------------------------------------------------------ blk-layout conversions (weight prep, op tests)
__global__ void split_blockify_kernel(const float* __restrict__ src, f16* __restrict__ hi, f16* __restrict__ lo, int M, int K) {
    const int Mp = (M + 255) / 256 * 256, KT = K / 32;
    const int64_t total = (int64_t)Mp * (K / 8);           // one work item = 8 consecutive k of one row
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(it / (K / 8)), k = (int)(it % (K / 8)) * 8;
        f16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = m < M ? src[(int64_t)m * K + k + e] : 0.f;
            f16 hh, ll; split_f16(x, hh, ll); h[e] = hh; l[e] = ll;
        }
        const int64_t o = blk_off(m, k, KT);
        *reinterpret_cast<f16x8*>(hi + o) = h;
        if (lo) *reinterpret_cast<f16x8*>(lo + o) = l;
    }
}
// split_blockify + the MX-fp4 side planes (quant4.h).  Consecutive work items are consecutive 8-k groups of one row, so
// four neighbouring lanes hold one 32-k block (K % 32 == 0 keeps groups from straddling rows or the grid stride).
__global__ void quant_blockify_kernel(const float* __restrict__ src, f16* __restrict__ hi, f16* __restrict__ lo,
                                      unsigned char* __restrict__ q, unsigned char* __restrict__ sc, int M, int K) {
    const int Mp = (M + 255) / 256 * 256, KT = K / 32;
    const int64_t total = (int64_t)Mp * (K / 8);
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(it / (K / 8)), k = (int)(it % (K / 8)) * 8;
        f16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = m < M ? src[(int64_t)m * K + k + e] : 0.f;
            f16 hh, ll; split_f16(x, hh, ll); h[e] = hh; l[e] = ll;
        }
        const int64_t o = blk_off(m, k, KT);
        *reinterpret_cast<f16x8*>(hi + o) = h;
        if (lo) *reinterpret_cast<f16x8*>(lo + o) = l;
        q4_store8(q, sc, KT, m, k >> 5, (k >> 3) & 3, h, l);
    }
}
__global__ void unblockify_f32_kernel(const f16* __restrict__ hi, const f16* __restrict__ lo, float* __restrict__ out, int M, int K) {
    const int KT = K / 32;
    const int64_t total = (int64_t)M * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / K), k = (int)(i % K);
        const int64_t o = blk_off(m, k, KT);
        out[i] = (float)hi[o] + (lo ? (float)lo[o] : 0.f);
    }
}

// ------------------------------------------------------------------ load-time range check of a GEMM weight: stats[0] = max |w| (as uint bits), stats[1] = sum w^2
__global__ __launch_bounds__(256)
void weight_stats_kernel(const float* __restrict__ w, int64_t n, unsigned* __restrict__ stats_max, float* __restrict__ stats_sq) {
    float mx = 0.f, sq = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = w[i];
        mx = fmaxf(mx, fabsf(v)); sq += v * v;
        if (!(fabsf(v) <= 3.0e38f)) mx = INFINITY;          // NaN / inf in a checkpoint
    }
    mx = wave_max(mx); sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) { atomicMax(stats_max, __float_as_uint(mx)); atomicAdd(stats_sq, sq); }
}

// ------------------------------------------------------------------ row L2 normalise (in place)
__global__ __launch_bounds__(256)
void l2norm_rows_kernel(float* __restrict__ x, int rows, int D, float eps, int* __restrict__ err_flag) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* r = x + (int64_t)row * D;
    float sq = 0.f;
    for (int i = lane; i < D; i += 64) sq += r[i] * r[i];
    const float nrm = sqrtf(wave_sum(sq));
    // a NaN / inf here means an activation left the fp16 range somewhere upstream (conversions do not saturate: common.h split_f16)
    if (err_flag && lane == 0 && !(nrm < INFINITY)) atomicOr(err_flag, 2);
    const float inv = 1.0f / fmaxf(nrm, eps);
    for (int i = lane; i < D; i += 64) r[i] *= inv;
}

// ------------------------------------------------------------------ row argmax (first max wins)
__global__ __launch_bounds__(256)
void row_argmax_kernel(const float* __restrict__ x, int rows, int cols, int32_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* r = x + (int64_t)row * cols;
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int i = lane; i < cols; i += 64) {
        const float v = r[i];
        if (v > bv || bi == 0x7fffffff) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) out[row] = bi;
}

// ------------------------------------------------------------------ row softmax(scale * x)
template <typename OutT>
__global__ __launch_bounds__(256)
void row_softmax_kernel(const float* __restrict__ x, int rows, int cols, float scale, OutT* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* r = x + (int64_t)row * cols;
    float mx = -INFINITY;
    for (int i = lane; i < cols; i += 64) mx = fmaxf(mx, r[i] * scale);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int i = lane; i < cols; i += 64) sum += expf(r[i] * scale - mx);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int i = lane; i < cols; i += 64) out[(int64_t)row * cols + i] = (OutT)(expf(r[i] * scale - mx) * inv);
}

// ------------------------------------------------------------------ rank_cls_score: mean_t[(v1-v2) - |v1+v2-1|]
__global__ __launch_bounds__(256)
void top2_partial_kernel(const float* __restrict__ x, int rows, int cols, float* __restrict__ partial) {
    // one thread per row (cols is the class count, 2..8); block partial sums -> partial[blockIdx.x]
    __shared__ float red[4];
    const int row = blockIdx.x * 256 + threadIdx.x;
    float sc = 0.f;
    if (row < rows) {
        const float* r = x + (int64_t)row * cols;
        float v1 = -INFINITY, v2 = -INFINITY;
        for (int i = 0; i < cols; ++i) {
            const float v = r[i];
            if (v > v1) { v2 = v1; v1 = v; } else if (v > v2) { v2 = v; }
        }
        sc = (v1 - v2) - fabsf(v1 + v2 - 1.0f);
    }
    sc = wave_sum(sc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256)
void top2_final_kernel(const float* __restrict__ partial, int n, int rows, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)rows;
}

// ------------------------------------------------------------------ BERT embeddings + LayerNorm
template <int NV>
__global__ __launch_bounds__(256)
void bert_embed_ln_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ type_ids,
                          const float* __restrict__ wemb, const float* __restrict__ pemb, const float* __restrict__ temb,
                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                          int rows, int T, int vocab, int type_vocab,
                          float* __restrict__ resid, f16* __restrict__ out_hi, f16* __restrict__ out_lo, int* err_flag) {
    constexpr int D = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int64_t id = ids[row];
    int64_t ty = type_ids ? type_ids[row] : 0;
    if (id < 0 || id >= vocab || ty < 0 || ty >= type_vocab) {
        if (lane == 0) atomicOr(err_flag, 1);
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        ty = ty < 0 ? 0 : (ty >= type_vocab ? type_vocab - 1 : ty);
    }
    const int t = row % T;
    f32x4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        const f32x4 a = *reinterpret_cast<const f32x4*>(wemb + id * D + col);
        const f32x4 b = *reinterpret_cast<const f32x4*>(pemb + (int64_t)t * D + col);
        const f32x4 c = *reinterpret_cast<const f32x4*>(temb + ty * D + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][e] = (a[e] + c[e]) + b[e]; sum += v[i][e]; }   // HF BertEmbeddings order: (word + token_type) + position
    }
    const float mean = wave_sum(sum) * (1.0f / D);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][e] -= mean; sq += v[i][e] * v[i][e]; }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) * (1.0f / D) + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + col);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + col);
        f32x4 y; f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[e] = v[i][e] * rstd * g[e] + bt[e];
            f16 hh, ll; split_f16(y[e], hh, ll); h[e] = hh; l[e] = ll;
        }
        *reinterpret_cast<f32x4*>(resid + (int64_t)row * D + col) = y;
        const int64_t o16 = blk_off(row, col, D / 32);
        *reinterpret_cast<f16x4*>(out_hi + o16) = h;
        if (out_lo) *reinterpret_cast<f16x4*>(out_lo + o16) = l;
    }
}

// ------------------------------------------------------------------ Resize(224, bicubic) + CenterCrop on raw uint8 tiles
// Pillow's 8-bit resample (what torchvision's Resize runs on PIL inputs, keep_inference.py:88-90): 22-bit fixed-point weights
// (built on the host in float64 exactly as libImaging does, keep_amd/preprocess.py), horizontal pass -> uint8 -> vertical
// pass -> uint8; only the columns / rows that survive the centre crop are computed.  Bit-identical to PIL.
__device__ __forceinline__ unsigned char clip8_fixed(int v) {
    v >>= 22;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
// tmp[b][y][xx][c] = sum_x src[b][y][x0 + x][c] * k[x]     for xx in the cropped column range
__global__ __launch_bounds__(256)
void resize_h_u8_kernel(const unsigned char* __restrict__ src, int B, int H, int W, const int* __restrict__ bounds,
                        const int* __restrict__ kk, int ksize, int col0, int ncols, unsigned char* __restrict__ tmp) {
    const int64_t total = (int64_t)B * H * ncols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int xx = (int)(i % ncols);
        const int64_t row = i / ncols;                            // b * H + y
        const int x0 = bounds[2 * (col0 + xx)], n = bounds[2 * (col0 + xx) + 1];
        const int* k = kk + (int64_t)(col0 + xx) * ksize;
        const unsigned char* s = src + (row * W + x0) * 3;
        int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
        for (int x = 0; x < n; ++x) { const int w = k[x]; a0 += s[3 * x] * w; a1 += s[3 * x + 1] * w; a2 += s[3 * x + 2] * w; }
        unsigned char* d = tmp + i * 3;
        d[0] = clip8_fixed(a0); d[1] = clip8_fixed(a1); d[2] = clip8_fixed(a2);
    }
}
// out[b][yy][xx][c] = sum_y tmp[b][y0 + y][xx][c] * k[y]    for yy in the cropped row range
__global__ __launch_bounds__(256)
void resize_v_u8_kernel(const unsigned char* __restrict__ tmp, int B, int H, int ncols, const int* __restrict__ bounds,
                        const int* __restrict__ kk, int ksize, int row0, int nrows, unsigned char* __restrict__ out) {
    const int64_t total = (int64_t)B * nrows * ncols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int xx = (int)(i % ncols);
        const int yy = (int)((i / ncols) % nrows);
        const int b = (int)(i / ((int64_t)ncols * nrows));
        const int y0 = bounds[2 * (row0 + yy)], n = bounds[2 * (row0 + yy) + 1];
        const int* k = kk + (int64_t)(row0 + yy) * ksize;
        const unsigned char* s = tmp + (((int64_t)b * H + y0) * ncols + xx) * 3;
        int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
        for (int y = 0; y < n; ++y) { const int w = k[y]; const unsigned char* p = s + (int64_t)y * ncols * 3; a0 += p[0] * w; a1 += p[1] * w; a2 += p[2] * w; }
        unsigned char* d = out + i * 3;
        d[0] = clip8_fixed(a0); d[1] = clip8_fixed(a1); d[2] = clip8_fixed(a2);
    }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, int64_t src_stride, float* __restrict__ dst, int rows, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * D) return;
    const int r = (int)(i / D), d = (int)(i % D);
    dst[i] = src[(int64_t)r * src_stride + d];
}

__global__ void scatter_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t dst_stride, int rows, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * D) return;
    const int r = (int)(i / D), d = (int)(i % D);
    dst[(int64_t)r * dst_stride + d] = src[i];
}

__global__ void gather_rows_blk_kernel(const f16* __restrict__ src, int row_stride, f16* __restrict__ dst, int rows, int D) {
    const int KT = D / 32;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one 16-byte chunk (8 k) per thread
    if (i >= (int64_t)rows * (D / 8)) return;
    const int r = (int)(i / (D / 8)), k = (int)(i % (D / 8)) * 8;
    *reinterpret_cast<f16x8*>(dst + blk_off(r, k, KT)) = *reinterpret_cast<const f16x8*>(src + blk_off(r * row_stride, k, KT));
}

}  // namespace keepk
using namespace keepk;

// ------------------------------------------------------------------ mean-input compensation (keep_calibrate_bias)
// out[k] += sum over rows m < M of x[m][k]  (x in blk layout; out zeroed by the caller).  One workgroup owns one 32-column K slice and walks
// every row tile in a fixed order: no atomics, the sums are bit-reproducible (launches on ONE stream accumulate in issue order)
__global__ __launch_bounds__(256)
void blk_col_sum_kernel(const f16* __restrict__ x, int M, int KT, float* __restrict__ out) {
    __shared__ float red[8][32];
    const int kt = blockIdx.x, k = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int tiles = (M + 255) >> 8;
    float acc = 0.f;
    for (int t = 0; t < tiles; ++t) {
        const f16* base = x + ((int64_t)t * KT + kt) * 8192 + k;
        const int rows = (M - t * 256) < 256 ? (M - t * 256) : 256;
        for (int r = rg; r < rows; r += 8) acc += (float)base[r * 32];
    }
    red[rg][k] = acc;
    __syncthreads();
    if (rg == 0) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) v += red[g][k];
        out[kt * 32 + k] += v;
    }
}
void launch_blk_col_sum(const f16* x, int M, int K, float* out, hipStream_t s) {
    hipLaunchKernelGGL(blk_col_sum_kernel, dim3(K / 32), dim3(256), 0, s, x, M, K / 32, out);
}

// out[n] = bias[n] + sum_k w_lo[n][k] * (col_sum[k] * inv_rows)     (w_lo: the lo plane of a GEMM weight, blk layout [N][K]; one wave per row)
__global__ __launch_bounds__(256)
void bias_mean_corr_kernel(const f16* __restrict__ w_lo, const float* __restrict__ col_sum, float inv_rows, const float* __restrict__ bias,
                           float* __restrict__ out, int N, int K) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    const int KT = K / 32;
    const f16* row = w_lo + (int64_t)(n >> 8) * KT * 8192 + (n & 255) * 32;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc += (float)row[(int64_t)(k >> 5) * 8192 + (k & 31)] * col_sum[k];
    acc = wave_sum(acc);
    if (lane == 0) out[n] = bias[n] + acc * inv_rows;
}
void launch_bias_mean_corr(const f16* w_lo, const float* col_sum, float inv_rows, const float* bias, float* out, int N, int K, hipStream_t s) {
    hipLaunchKernelGGL(bias_mean_corr_kernel, dim3((N + 3) / 4), dim3(256), 0, s, w_lo, col_sum, inv_rows, bias, out, N, K);
}

void launch_gather_rows_blk(const f16* src, int row_stride, f16* dst, int rows, int D, hipStream_t s) {
    const int64_t n = (int64_t)rows * (D / 8);
    hipLaunchKernelGGL(gather_rows_blk_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, row_stride, dst, rows, D);
}

int launch_layernorm(const LnParams& p, hipStream_t s) {
    const int ln_impl = p.tune ? p.tune->ln_impl : 1;
    const bool blk_ok = p.out_kt > 0 && p.out_hi && (p.D == 1024 || p.D == 768) && p.out_kt == p.D / 32;
    if (p.out_q && !blk_ok) return -1;                  // the fp4 planes are only produced by the blk-layout kernel
    if (ln_impl == 2 && blk_ok && p.D == 1024 && !p.out_q) {
        // 4-wave workgroups (4 rows, 256-byte runs): one wave of 56 registers per SIMD fits beside TWO 208-register waves of the other lane's persistent
        // fc1 GEMM (an 8-wave LayerNorm workgroup needs 112 registers per SIMD and only fits beside the 200-register qkv kernel)
        constexpr int R = 4;
        const size_t lds = (size_t)R * (p.D + 32) * 2 * (p.out_lo ? 2 : 1);
        hipLaunchKernelGGL((layernorm_blk_kernel<4, R>), dim3((p.rows + R - 1) / R), dim3(R * 64), lds, s, p);
        return 0;
    }
    if ((ln_impl >= 1 || p.out_q) && blk_ok) {
        constexpr int R = 8;
        const size_t lds = (size_t)R * (p.D + 32) * 2 * ((p.out_lo || (p.out_q && !p.out_q_hi_only)) ? 2 : 1);
        dim3 g((p.rows + R - 1) / R), b(R * 64);
        if (p.D == 1024) hipLaunchKernelGGL((layernorm_blk_kernel<4, R>), g, b, lds, s, p);
        else hipLaunchKernelGGL((layernorm_blk_kernel<3, R>), g, b, lds, s, p);
        return 0;
    }
    dim3 grid((p.rows + 3) / 4), block(256);
    if (p.D == 1024) hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, p);
    else if (p.D == 768) hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, s, p);
    else if (p.D == 512) hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, p);
    else if (p.D == 256) hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, p);
    else return -1;
    return 0;
}

void launch_im2col(const void* pixels, int dtype, int B, f16* out_hi, f16* out_lo,
                   const float* cls, const float* pos, float* resid, int D, hipStream_t s) {
    const int64_t total = (int64_t)B * 3 * 224 * 28;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (dtype == PIX_U8_HWC) {
        const int64_t t8 = (int64_t)B * 224 * 28;
        int b8 = (int)((t8 + 255) / 256); if (b8 > 256 * 16) b8 = 256 * 16;
        hipLaunchKernelGGL(im2col_u8_kernel, dim3(b8), dim3(256), 0, s, (const unsigned char*)pixels, B, out_hi, out_lo);
    } else if (dtype == PIX_F32) hipLaunchKernelGGL(im2col_kernel<PIX_F32>, dim3(blocks), dim3(256), 0, s, pixels, B, out_hi, out_lo);
    else if (dtype == PIX_F16) hipLaunchKernelGGL(im2col_kernel<PIX_F16>, dim3(blocks), dim3(256), 0, s, pixels, B, out_hi, out_lo);
    else hipLaunchKernelGGL(im2col_kernel<PIX_BF16>, dim3(blocks), dim3(256), 0, s, pixels, B, out_hi, out_lo);
    const int64_t n = (int64_t)B * D;
    hipLaunchKernelGGL(cls_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cls, pos, resid, B, D, 197);
}

void launch_resize_crop_u8(const unsigned char* src, int B, int H, int W, const int* xb, const int* xk, int xks, int col0, int ncols,
                           const int* yb, const int* yk, int yks, int row0, int nrows, unsigned char* tmp, unsigned char* out, hipStream_t s) {
    auto blocks = [](int64_t n) { int64_t b = (n + 255) / 256; return (unsigned)(b > 65536 ? 65536 : (b < 1 ? 1 : b)); };
    hipLaunchKernelGGL(resize_h_u8_kernel, dim3(blocks((int64_t)B * H * ncols)), dim3(256), 0, s, src, B, H, W, xb, xk, xks, col0, ncols, tmp);
    hipLaunchKernelGGL(resize_v_u8_kernel, dim3(blocks((int64_t)B * nrows * ncols)), dim3(256), 0, s, tmp, B, H, ncols, yb, yk, yks, row0, nrows, out);
}
void launch_split_f16(const float* src, f16* hi, f16* lo, int64_t n, hipStream_t s) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(split_f16_kernel, dim3(blocks), dim3(256), 0, s, src, hi, lo, n);
}

void launch_split_blockify(const float* src, f16* hi, f16* lo, int M, int K, hipStream_t s) {
    const int64_t total = (int64_t)((M + 255) / 256 * 256) * (K / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(split_blockify_kernel, dim3(blocks), dim3(256), 0, s, src, hi, lo, M, K);
}
void launch_quant_blockify(const float* src, f16* hi, f16* lo, unsigned char* q, unsigned char* sc, int M, int K, hipStream_t s) {
    const int64_t total = (int64_t)((M + 255) / 256 * 256) * (K / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(quant_blockify_kernel, dim3(blocks), dim3(256), 0, s, src, hi, lo, q, sc, M, K);
}
void launch_unblockify_f32(const f16* hi, const f16* lo, float* out, int M, int K, hipStream_t s) {
    const int64_t total = (int64_t)M * K;
    int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(unblockify_f32_kernel, dim3(blocks), dim3(256), 0, s, hi, lo, out, M, K);
}

void launch_weight_stats(const float* w, int64_t n, float* stats2, hipStream_t s) {
    (void)hipMemsetAsync(stats2, 0, 2 * sizeof(float), s);
    int blocks = (int)((n + 255) / 256); if (blocks > 256) blocks = 256;      // one atomic pair per wave: keep the count low (2048 blocks: 200 us per tensor on the atomics alone)
    hipLaunchKernelGGL(weight_stats_kernel, dim3(blocks), dim3(256), 0, s, w, n, (unsigned*)stats2, stats2 + 1);
}
void launch_l2norm_rows(float* x, int rows, int D, float eps, hipStream_t s, int* err_flag) {
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, D, eps, err_flag);
}
void launch_row_argmax(const float* x, int rows, int cols, int32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(row_argmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, cols, out);
}
void launch_row_softmax(const float* x, int rows, int cols, float scale, float* out, hipStream_t s) {
    hipLaunchKernelGGL(row_softmax_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, cols, scale, out);
}
void launch_row_softmax_f16(const float* x, int rows, int cols, float scale, f16* out, hipStream_t s) {
    hipLaunchKernelGGL(row_softmax_kernel<f16>, dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, cols, scale, out);
}
void launch_top2_score(const float* x, int rows, int cols, float* partial, float* out, hipStream_t s) {
    const int nb = (rows + 255) / 256;
    hipLaunchKernelGGL(top2_partial_kernel, dim3(nb), dim3(256), 0, s, x, rows, cols, partial);
    hipLaunchKernelGGL(top2_final_kernel, dim3(1), dim3(256), 0, s, partial, nb, rows, out);
}
void launch_bert_embed_ln(const int64_t* ids, const int64_t* type_ids, const float* wemb, const float* pemb,
                          const float* temb, const float* gamma, const float* beta, float eps,
                          int P, int T, int D, int vocab, int type_vocab,
                          float* resid, f16* out_hi, f16* out_lo, int* err_flag, hipStream_t s) {
    const int rows = P * T;
    dim3 grid((rows + 3) / 4), block(256);
    if (D == 768)
        hipLaunchKernelGGL(bert_embed_ln_kernel<3>, grid, block, 0, s, ids, type_ids, wemb, pemb, temb, gamma, beta, eps,
                           rows, T, vocab, type_vocab, resid, out_hi, out_lo, err_flag);
    else if (D == 1024)
        hipLaunchKernelGGL(bert_embed_ln_kernel<4>, grid, block, 0, s, ids, type_ids, wemb, pemb, temb, gamma, beta, eps,
                           rows, T, vocab, type_vocab, resid, out_hi, out_lo, err_flag);
    else
        hipLaunchKernelGGL(bert_embed_ln_kernel<1>, grid, block, 0, s, ids, type_ids, wemb, pemb, temb, gamma, beta, eps,
                           rows, T, vocab, type_vocab, resid, out_hi, out_lo, err_flag);
}
void launch_scatter_rows_f32(const float* src, float* dst, int64_t dst_stride, int rows, int D, hipStream_t s) {
    const int64_t n = (int64_t)rows * D;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, dst_stride, rows, D);
}
void launch_gather_rows_f32(const float* src, int64_t src_stride, float* dst, int rows, int D, hipStream_t s) {
    const int64_t n = (int64_t)rows * D;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, src_stride, dst, rows, D);
}
