// Slide-level zero-shot kernels: the per-tile reductions the reference does
// in Python loops with one device->host sync per classifier / per tile.
//
//   group_top2_partial   rank_cls_score for EVERY prompt classifier at once
//                        (WSI_evaluation/utils.py:107-117 inside the loop at :127-130):
//                        logits [n, K*C] -> per classifier k the sum over tiles of (v1-v2) - |v1+v2-1|
//   coord_insert / refine_mean
//                        refine_seg (subtyping_utils.py:38-65, detection_utils.py:39-74,
//                        segment_utils.py:63-89): first occurrence of a coordinate wins; each tile's
//                        probabilities are averaged over the existing tiles among
//                        {(x-p,y-p),(x,y-p),(x-p,y),(x,y)} in that order (float32 sum, then / count,
//                        exactly numpy's mean over <= 4 rows)
#include "common.h"
#include "../../include/keep_hip.h"

namespace keepk {

// one thread per classifier k; a block walks `rows_per_block` tiles.  Consecutive threads read consecutive
// C-wide groups of one logits row -> coalesced.  partial[rb][k] is written once (deterministic).
__global__ __launch_bounds__(256)
void group_top2_partial_kernel(const float* __restrict__ logits, int n, int K, int C, int rows_per_block,
                               float* __restrict__ partial) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(n, r0 + rows_per_block);
    if (k >= K) return;
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) {
        const float* p = logits + (int64_t)r * K * C + (int64_t)k * C;
        float v1 = -INFINITY, v2 = -INFINITY;
        for (int c = 0; c < C; ++c) {
            const float v = p[c];
            if (v > v1) { v2 = v1; v1 = v; } else if (v > v2) { v2 = v; }
        }
        acc += (v1 - v2) - fabsf(v1 + v2 - 1.0f);
    }
    partial[(int64_t)blockIdx.y * K + k] = acc;
}
__global__ __launch_bounds__(256)
void group_top2_reduce_kernel(const float* __restrict__ partial, int nrb, int K, float* __restrict__ sums) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    float s = sums[k];
    for (int b = 0; b < nrb; ++b) s += partial[(int64_t)b * K + k];
    sums[k] = s;
}

// ---- coordinate hash: open addressing on packed (x, y), value = smallest tile index with that key
// Coordinates must fit int32 (the Python wrapper checks; slides are < 2^20 pixels wide).  Each half is stored with its sign
// bit flipped, so the all-ones EMPTY_KEY would be (INT32_MAX, INT32_MAX) -- outside what the wrapper admits -- and the
// neighbour (-1, -1) of a tile at (patch-1, patch-1) is an ordinary key.
__device__ __forceinline__ unsigned long long pack_xy(long long x, long long y) {
    return ((unsigned long long)(unsigned)((int)x ^ 0x80000000) << 32) | (unsigned long long)(unsigned)((int)y ^ 0x80000000);
}
__device__ __forceinline__ unsigned hash_xy(unsigned long long k, unsigned mask) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return (unsigned)k & mask;
}
constexpr unsigned long long EMPTY_KEY = 0xffffffffffffffffULL;

__global__ void coord_insert_kernel(const long long* __restrict__ coords, int n, unsigned long long* keys, int* first,
                                    unsigned mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = pack_xy(coords[2 * i], coords[2 * i + 1]);
    unsigned s = hash_xy(key, mask);
    while (true) {
        const unsigned long long prev = atomicCAS(&keys[s], EMPTY_KEY, key);
        if (prev == EMPTY_KEY || prev == key) { atomicMin(&first[s], i); return; }
        s = (s + 1) & mask;
    }
}
__device__ __forceinline__ int coord_lookup(const unsigned long long* keys, const int* first, unsigned mask, long long x, long long y) {
    const unsigned long long key = pack_xy(x, y);
    unsigned s = hash_xy(key, mask);
    while (true) {
        const unsigned long long k = keys[s];
        if (k == EMPTY_KEY) return -1;
        if (k == key) return first[s];
        s = (s + 1) & mask;
    }
}
// one thread per (tile, class); is_first[i] = 1 when tile i is the first with its coordinate
__global__ void refine_mean_kernel(const float* __restrict__ probs, const long long* __restrict__ coords, int n, int C,
                                   long long patch, int overlap, const unsigned long long* keys, const int* first,
                                   unsigned mask, float* __restrict__ out, int* __restrict__ is_first) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * C) return;
    const int i = (int)(t / C), c = (int)(t % C);
    const long long x = coords[2 * i], y = coords[2 * i + 1];
    const int self = coord_lookup(keys, first, mask, x, y);
    if (c == 0) is_first[i] = (self == i);
    if (self != i) { out[t] = probs[t]; return; }
    if (!overlap) { out[t] = probs[t]; return; }
    const long long nx[4] = {x - patch, x, x - patch, x}, ny[4] = {y - patch, y - patch, y, y};
    float sum = 0.f; int cnt = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = q == 3 ? i : coord_lookup(keys, first, mask, nx[q], ny[q]);
        if (j >= 0) { sum = cnt == 0 ? probs[(int64_t)j * C + c] : sum + probs[(int64_t)j * C + c]; ++cnt; }
    }
    out[t] = sum / (float)cnt;
}

// sums the per-(row slot) partial scores of the fused screening GEMM (EPI_TOP2) in a fixed order -> deterministic
__global__ __launch_bounds__(256)
void top2_slots_reduce_kernel(const float* __restrict__ partial, int nslots, int kpad, int K, float scale, float* __restrict__ out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    float s = 0.f;
    for (int b = 0; b < nslots; ++b) s += partial[(int64_t)b * kpad + k];
    out[k] = s * scale;
}
__global__ void scale_vec_kernel(const float* __restrict__ in, int n, float f, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * f;
}

// Retrieval rank of the target column (training/path_training/zero_shot.py:168-171 + retrieval_metrics): how many
// images score above image `target[row]` for text `row`.  `arr.argsort()[-50:][::-1]` lists equal scores with the
// higher index first, so an equal score at a higher index also counts as "above".  One workgroup per text row.
__global__ __launch_bounds__(256)
void diag_rank_kernel(const float* __restrict__ sim, int n_img, const int* __restrict__ target, int row0, int* __restrict__ rank) {
    const int row = blockIdx.x;
    const float* r = sim + (int64_t)row * n_img;
    const int t = target ? target[row0 + row] : row0 + row;
    const float st = r[t];
    int cnt = 0;
    for (int j = threadIdx.x; j < n_img; j += 256) {
        const float v = r[j];
        cnt += (v > st) || (v == st && j > t);
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    __shared__ int part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) rank[row0 + row] = part[0] + part[1] + part[2] + part[3];
}

// Tile x class similarity for a handful of classes (P <= 8: the 2- and 4-column classifiers of the WSI flows and their
// probability maps, subtyping_utils.py:69-72).  The MFMA GEMM pads N to a 128-wide tile and runs at 1.3 TB/s on this
// shape; the job is a stream of the [N, D] features (HBM-bound), so: class vectors in registers,
// one wave per tile row, fp32 FMA + cross-lane sum, the softmax / argmax fused behind it.
template <int PC, int KV>
__global__ __launch_bounds__(256)
void sim_small_kernel(const float* __restrict__ img, const float* __restrict__ txt, int N, int P, float scale, int mode,
                      float* __restrict__ out_f32, f16* __restrict__ out_f16, int32_t* __restrict__ amax) {
    constexpr int D = KV * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 t[PC][KV];
#pragma unroll
    for (int p = 0; p < PC; ++p)
#pragma unroll
        for (int i = 0; i < KV; ++i)
            t[p][i] = p < P ? *reinterpret_cast<const f32x4*>(txt + (int64_t)p * D + (i * 64 + lane) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const int stride = gridDim.x * 4;
    for (int row0 = blockIdx.x * 4 + wave; row0 < N; row0 += 2 * stride) {
        const int row1 = row0 + stride;
        const bool two = row1 < N;
        f32x4 a[2][KV];
#pragma unroll
        for (int i = 0; i < KV; ++i) {
            a[0][i] = *reinterpret_cast<const f32x4*>(img + (int64_t)row0 * D + (i * 64 + lane) * 4);
            a[1][i] = *reinterpret_cast<const f32x4*>(img + (int64_t)(two ? row1 : row0) * D + (i * 64 + lane) * 4);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r == 1 && !two) break;
            const int row = r ? row1 : row0;
            float v[PC];
#pragma unroll
            for (int p = 0; p < PC; ++p) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < KV; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = fmaf(a[r][i][e], t[p][i][e], acc);
                v[p] = wave_sum(acc);
            }
            if (mode == KEEP_SIM_RAW || mode == KEEP_SIM_ARGMAX) {
                float mine = 0.f; float bv = -INFINITY; int bi = 0;
#pragma unroll
                for (int p = 0; p < PC; ++p) {
                    const float x = v[p] * scale;
                    if (lane == p) mine = x;
                    if (p < P && x > bv) { bv = x; bi = p; }
                }
                if (out_f32 && lane < P) out_f32[(int64_t)row * P + lane] = mine;
                if (mode == KEEP_SIM_ARGMAX && lane == 0) amax[row] = bi;
            } else {                                   // softmax(scale * cos), arithmetic as row_softmax_kernel
                float mx = -INFINITY;
#pragma unroll
                for (int p = 0; p < PC; ++p) if (p < P) mx = fmaxf(mx, v[p] * scale);
                float sum = 0.f, mine = 0.f;
#pragma unroll
                for (int p = 0; p < PC; ++p) {
                    if (p < P) {
                        const float e = expf(v[p] * scale - mx);
                        sum += e;
                        if (lane == p) mine = e;
                    }
                }
                const float y = mine * (1.0f / sum);
                if (lane < P) {
                    if (mode == KEEP_SIM_SOFTMAX) out_f32[(int64_t)row * P + lane] = y;
                    else out_f16[(int64_t)row * P + lane] = (f16)y;
                }
            }
        }
    }
}

// ---- tile x prompt similarity for 9..64 prompts (BASELINE config 3: 4096 tiles x 64 prompts, sim matrix + argmax) --------------
// The 128x128 fp32-MFMA GEMM tile leaves most of the chip idle at this shape (64 columns = half a tile wide, 32 workgroups for
// 4096 rows) and needs a second kernel for the argmax.  Here a workgroup owns 64 tile rows x all prompts: each wave 16 rows x NC
// 16-column tiles on v_mfma_f32_16x16x4_f32 (exact f32 FMA chain), the prompt vectors staged through LDS 16 k at a time (double
// buffered, one barrier per block), the tile rows streamed straight from HBM, and the row argmax / softmax taken on the
// accumulators (4 rows x NC columns per lane, then 4 shuffle steps across the 16 lanes of a row).
template <int NC>
__global__ __launch_bounds__(256)
void sim_mid_kernel(const float* __restrict__ img, const float* __restrict__ txt, int N, int P, int D, float scale, int mode,
                    float* __restrict__ out_f32, f16* __restrict__ out_f16, int32_t* __restrict__ amax) {
    __shared__ f32x4 sB[2][4][NC * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int row_base = blockIdx.x * 64 + wave * 16;
    const int arow = min(row_base + j, N - 1);
    const float* ap = img + (int64_t)arow * D + kq * 4;
    const bool stager = tid < NC * 64;                       // one float4 of one prompt per thread and block
    const int bcol = tid >> 2, bq = tid & 3;
    const float* bp = txt + (int64_t)min(bcol, P - 1) * D + bq * 4;
    f32x4 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nblk = D >> 4;
    f32x4 a_next = *reinterpret_cast<const f32x4*>(ap), b_next = f32x4{0.f, 0.f, 0.f, 0.f};
    if (stager) b_next = *reinterpret_cast<const f32x4*>(bp);
    for (int kb = 0; kb < nblk; ++kb) {
        if (stager) sB[kb & 1][bq][bcol] = b_next;
        const f32x4 a = a_next;
        __syncthreads();
        if (kb + 1 < nblk) {
            a_next = *reinterpret_cast<const f32x4*>(ap + (kb + 1) * 16);
            if (stager) b_next = *reinterpret_cast<const f32x4*>(bp + (kb + 1) * 16);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const f32x4 b = sB[kb & 1][kq][c * 16 + j];
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc[c], 0, 0, 0);
        }
    }
    // lane (j, kq) holds rows row_base + kq*4 + r (r = 0..3) at columns c*16 + j
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row_base + kq * 4 + r;
        const bool rok = row < N;
        float x[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) x[c] = acc[c][r] * scale;
        if (mode == KEEP_SIM_RAW || mode == KEEP_SIM_ARGMAX) {
            if (out_f32 && rok) {
#pragma unroll
                for (int c = 0; c < NC; ++c) if (c * 16 + j < P) out_f32[(int64_t)row * P + c * 16 + j] = x[c];
            }
            if (mode == KEEP_SIM_ARGMAX) {
                float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
                for (int c = 0; c < NC; ++c) if (c * 16 + j < P && x[c] > bv) { bv = x[c]; bi = c * 16 + j; }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {                      // first maximum wins, as torch.argmax
                    const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                }
                if (j == 0 && rok) amax[row] = bi;
            }
        } else {                                                        // softmax(scale * cos), arithmetic as row_softmax_kernel
            float mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < NC; ++c) if (c * 16 + j < P) mx = fmaxf(mx, x[c]);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            float e[NC], sum = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) { e[c] = (c * 16 + j < P) ? expf(x[c] - mx) : 0.f; sum += e[c]; }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o);
            const float inv = 1.0f / sum;
            if (rok) {
#pragma unroll
                for (int c = 0; c < NC; ++c) if (c * 16 + j < P) {
                    if (mode == KEEP_SIM_SOFTMAX) out_f32[(int64_t)row * P + c * 16 + j] = e[c] * inv;
                    else out_f16[(int64_t)row * P + c * 16 + j] = (f16)(e[c] * inv);
                }
            }
        }
    }
}


// ---- keep_classify: which tiles need the accurate arithmetic, and moving them ---------------------------------------------
// flags[r] = 1 where the two largest entries of sim row r are closer than `bound` (a label the default arithmetic cannot vouch for)
__global__ __launch_bounds__(256)
void top2_margin_flag_kernel(const float* __restrict__ sim, int N, int P, float bound, int* __restrict__ flags) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    const float* x = sim + (int64_t)r * P;
    float v1 = -INFINITY, v2 = -INFINITY;
    for (int c = 0; c < P; ++c) {
        const float v = x[c];
        if (v > v1) { v2 = v1; v1 = v; } else if (v > v2) v2 = v;
    }
    // NaN-safe: a row with a non-finite entry is always re-examined
    flags[r] = (P > 1 && !(v1 - v2 >= bound)) ? 1 : 0;
}
// ordered compaction of the flagged row indices by ONE workgroup (N is a slide's tile count: a few thousand to a few hundred thousand)
__global__ __launch_bounds__(1024)
void compact_flags_kernel(const int* __restrict__ flags, int N, int* __restrict__ list, int* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < N; i0 += 1024) {
        const int i = i0 + tid;
        const int f = (i < N && flags[i] != 0) ? 1 : 0;
        const unsigned long long b = __ballot(f);
        const int pre = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[w] = __popcll(b);
        __syncthreads();
        int off = base_s;
        for (int k = 0; k < w; ++k) off += wsum[k];
        if (f) list[off + pre] = i;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int k = 0; k < 16; ++k) t += wsum[k]; base_s += t; }
        __syncthreads();
    }
    if (tid == 0) *count = base_s;
}
__global__ __launch_bounds__(256)
void gather_tiles_kernel(const uint4* __restrict__ src, int64_t tile_vec, const int* __restrict__ list, uint4* __restrict__ dst) {
    const int t = blockIdx.y;
    const uint4* s = src + (int64_t)list[t] * tile_vec;
    uint4* d = dst + (int64_t)t * tile_vec;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tile_vec; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ __launch_bounds__(256)
void scatter_rows_kernel(const float* __restrict__ src, const int* __restrict__ list, int D, float* __restrict__ dst) {
    const int t = blockIdx.x;
    for (int c = threadIdx.x; c < D; c += blockDim.x) dst[(int64_t)list[t] * D + c] = src[(int64_t)t * D + c];
}

}  // namespace keepk
using namespace keepk;

// 9..64 prompts: returns 0 when handled, -1 when the shape is left to the GEMM path
int launch_sim_mid(const float* img, const float* txt, int N, int P, int D, float scale, int mode, void* out, int32_t* amax, hipStream_t s) {
    if (P <= 8 || P > 64 || D % 16 || N < 1) return -1;
    if (mode != KEEP_SIM_RAW && mode != KEEP_SIM_ARGMAX && mode != KEEP_SIM_SOFTMAX && mode != KEEP_SIM_SOFTMAX_F16) return -1;
    float* of = (mode == KEEP_SIM_SOFTMAX_F16) ? nullptr : (float*)out;
    f16* oh = (mode == KEEP_SIM_SOFTMAX_F16) ? (f16*)out : nullptr;
    dim3 g((N + 63) / 64), b(256);
    const int nc = (P + 15) / 16;
    if (nc == 1) hipLaunchKernelGGL(sim_mid_kernel<1>, g, b, 0, s, img, txt, N, P, D, scale, mode, of, oh, amax);
    else if (nc == 2) hipLaunchKernelGGL(sim_mid_kernel<2>, g, b, 0, s, img, txt, N, P, D, scale, mode, of, oh, amax);
    else if (nc == 3) hipLaunchKernelGGL(sim_mid_kernel<3>, g, b, 0, s, img, txt, N, P, D, scale, mode, of, oh, amax);
    else hipLaunchKernelGGL(sim_mid_kernel<4>, g, b, 0, s, img, txt, N, P, D, scale, mode, of, oh, amax);
    return 0;
}

// returns 0 when handled, -1 when the shape is left to the GEMM path
int launch_sim_small(const float* img, const float* txt, int N, int P, int D, float scale, int mode, void* out, int32_t* amax,
                     hipStream_t s) {
    if (P > 8 || (D != 768 && D != 1024) || mode == KEEP_SIM_TOP2SCORE) return -1;
    int blocks = (N + 7) / 8;
    if (blocks > 256 * 12) blocks = 256 * 12;
    dim3 g(blocks), b(256);
    float* of = (mode == KEEP_SIM_SOFTMAX_F16) ? nullptr : (float*)out;
    f16* oh = (mode == KEEP_SIM_SOFTMAX_F16) ? (f16*)out : nullptr;
#define KEEP_SIM_LAUNCH(PC, KV) hipLaunchKernelGGL((sim_small_kernel<PC, KV>), g, b, 0, s, img, txt, N, P, scale, mode, of, oh, amax)
    if (D == 768) { if (P <= 2) KEEP_SIM_LAUNCH(2, 3); else if (P <= 4) KEEP_SIM_LAUNCH(4, 3); else KEEP_SIM_LAUNCH(8, 3); }
    else          { if (P <= 2) KEEP_SIM_LAUNCH(2, 4); else if (P <= 4) KEEP_SIM_LAUNCH(4, 4); else KEEP_SIM_LAUNCH(8, 4); }
#undef KEEP_SIM_LAUNCH
    return 0;
}

void launch_diag_rank(const float* sim, int rows, int n_img, const int* target, int row0, int* rank, hipStream_t s) {
    hipLaunchKernelGGL(diag_rank_kernel, dim3(rows), dim3(256), 0, s, sim, n_img, target, row0, rank);
}

void launch_scale_vec(const float* in, int n, float f, float* out, hipStream_t s) {
    hipLaunchKernelGGL(scale_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, n, f, out);
}

void launch_group_top2(const float* logits, int n, int K, int C, float* partial, int max_row_blocks, float* sums, hipStream_t s) {
    int rpb = (n + max_row_blocks - 1) / max_row_blocks; if (rpb < 1) rpb = 1;
    const int nrb = (n + rpb - 1) / rpb;
    dim3 grid((K + 255) / 256, nrb);
    hipLaunchKernelGGL(group_top2_partial_kernel, grid, dim3(256), 0, s, logits, n, K, C, rpb, partial);
    hipLaunchKernelGGL(group_top2_reduce_kernel, dim3((K + 255) / 256), dim3(256), 0, s, partial, nrb, K, sums);
}
void launch_top2_slots_reduce(const float* partial, int nslots, int kpad, int K, float scale, float* out, hipStream_t s) {
    hipLaunchKernelGGL(top2_slots_reduce_kernel, dim3((K + 255) / 256), dim3(256), 0, s, partial, nslots, kpad, K, scale, out);
}
void launch_refine(const float* probs, const long long* coords, int n, int C, long long patch, int overlap,
                   unsigned long long* keys, int* first, unsigned table_size, float* out, int* is_first, hipStream_t s) {
    (void)hipMemsetAsync(keys, 0xff, (size_t)table_size * 8, s);
    (void)hipMemsetAsync(first, 0x7f, (size_t)table_size * 4, s);
    hipLaunchKernelGGL(coord_insert_kernel, dim3((n + 255) / 256), dim3(256), 0, s, coords, n, keys, first, table_size - 1);
    const int64_t total = (int64_t)n * C;
    hipLaunchKernelGGL(refine_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, probs, coords, n, C, patch, overlap,
                       keys, first, table_size - 1, out, is_first);
}

void launch_top2_margin_flags(const float* sim, int N, int P, float bound, int* flags, int* list, int* count, hipStream_t s) {
    hipLaunchKernelGGL(top2_margin_flag_kernel, dim3((N + 255) / 256), dim3(256), 0, s, sim, N, P, bound, flags);
    hipLaunchKernelGGL(compact_flags_kernel, dim3(1), dim3(1024), 0, s, flags, N, list, count);
}
void launch_gather_tiles(const void* src, int64_t tile_bytes, const int* list, int n, void* dst, hipStream_t s) {
    hipLaunchKernelGGL(gather_tiles_kernel, dim3(32, n), dim3(256), 0, s, (const uint4*)src, tile_bytes / 16, list, (uint4*)dst);
}
void launch_scatter_rows(const float* src, const int* list, int n, int D, float* dst, hipStream_t s) {
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(n), dim3(256), 0, s, src, list, D, dst);
}
