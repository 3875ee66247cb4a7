// Whole-sequence softmax attention for short sequences (ViT: 197 tokens, BERT: <=512), gfx950.
//
//   out[b, t, h*64:(h+1)*64] = softmax(q k^T / 8 + key_bias) v        per (b, h)
//
// Reference arithmetic: timm Attention.forward (scaled_dot_product_attention, no mask) -- SURVEY.md
// §A.1; HF BertSelfAttention with the key-padding additive mask -- §A.2.
//
// Layout / mapping:
//   * one workgroup (4 wavefronts) per (batch, head); K (row-major, XOR-swizzled 16-B slots) and
//     V^T (feature-major, padded rows) of that head live in LDS for the whole workgroup
//   * each wavefront owns 16 query rows at a time and computes S^T = K Q^T with
//     mfma_f32_16x16x32_f16 (A operand = K rows from LDS, B operand = Q rows from HBM), so a lane
//     holds, for ONE query (lane&15), the scores of keys {16t + 4*(lane>>4) + r}: the softmax
//     reduction is in-register plus two cross-lane steps (xor 16, xor 32)
//   * the same registers, converted to fp16, ARE the B operand of  O^T = V^T P^T  (the key
//     permutation they imply is applied to the V^T fragment reads instead): no P round trip
//   * normalisation by 1/sum is applied to the fp32 O accumulators
//   * SPLIT mode (strict precision) runs hi/lo products: QhKh+QlKh+QhKl and PhVh+PlVh+PhVl
#include "common.h"

namespace keepk {

constexpr int ATT_MAX_THREADS = 512;
constexpr int HD = 64;

__host__ __device__ constexpr int att_kp2(int NT) { return ((NT + 1) / 2) * 32; }
// V^T row stride (f16).  The PV fragment reads compile to ds_read2_b64 (16-lane groups, 32 banks): with
// KP2 + 4 the stride is 2*odd dwords (mod 32), so the 16 feature rows of a group land on 16 distinct even
// banks -- conflict free (KP2 + 8 measured 35 % of the kernel's LDS cycles as bank conflicts).
__host__ __device__ constexpr int att_vs(int NT) { return att_kp2(NT) + 4; }
__host__ __device__ constexpr size_t att_lds_bytes(int NT, bool split) {
    size_t one = (size_t)NT * 16 * HD * 2 + (size_t)att_kp2(NT) * HD * 2;       // K image + V image, both row-major [key][64]
    return one * (split ? 2 : 1) + (size_t)NT * 16 * 4;
}

__device__ __forceinline__ unsigned sel4(const uint4& x, int i) {
    return i == 0 ? x.x : (i == 1 ? x.y : (i == 2 ? x.z : x.w));
}

typedef __fp16 att_h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
// ds_read_b64_tr_b16: within each group of 16 lanes, lane i receives as element e the element (i % 4) of source lane (4 e + i / 4)
__device__ __forceinline__ f16x4 tr_read4(const f16* p) {
    const att_h16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((att_h16x4 __attribute__((address_space(3)))*)(p));
    return __builtin_bit_cast(f16x4, v);
}

typedef const __attribute__((address_space(1))) void* att_gptr_t;
typedef __attribute__((address_space(3))) void* att_lptr_t;

// Stage one head's K (row-major, swizzled slots) and V^T (feature-major) into LDS.
//   K   : LDS-DMA (global_load_lds, 16 B per lane, no registers): the LDS image is linear in the slot
//         index, so the slot swizzle is applied to the per-lane SOURCE address.
//   V^T : needs a transpose, so it goes through registers -- but ALL loads are issued before the first
//         LDS write (the first version looped load->write and paid one memory round trip per pass:
//         72 % of the kernel's wave-cycles were s_waitcnt).
// Rows >= ntok are filled with a copy of the last valid row (finite values); their scores are masked
// with -inf, so P is exactly 0 there and 0 * finite contributes nothing.
template <int NT, int ATT_THREADS>
__device__ __forceinline__ void stage_kv(const f16* __restrict__ base, int ntok, int D3, int koff, int voff,
                                         f16* sK, f16* sVt, int tid, int wave) {
    constexpr int NKP = NT * 16;
    constexpr int VS = att_vs(NT);
    constexpr int NPAIR = att_kp2(NT) / 2;
    constexpr int K_ITEMS = NKP * 8, V_ITEMS = NPAIR * 8;
    constexpr int K_IT = (K_ITEMS + ATT_THREADS - 1) / ATT_THREADS;
    constexpr int V_IT = (V_ITEMS + ATT_THREADS - 1) / ATT_THREADS;
#pragma unroll
    for (int it = 0; it < K_IT; ++it) {
        const int L = tid + it * ATT_THREADS;              // LDS slot index; K_ITEMS is a multiple of 128,
        if (L < K_ITEMS) {                                 // so this predicate is wave-uniform
            const int row = L >> 3, c = (L & 7) ^ ((row >> 1) & 7);
            const int rc = row < ntok ? row : ntok - 1;
            __builtin_amdgcn_global_load_lds((att_gptr_t)(base + (int64_t)rc * D3 + koff + c * 8),
                                             (att_lptr_t)(sK + (it * ATT_THREADS + wave * 64) * 8), 16, 0, 0);
        }
    }
    // V: row-major as well, by LDS-DMA.  The P.V MFMA wants V^T fragments (4 consecutive keys of ONE feature per lane); they are
    // produced at read time by ds_read_b64_tr_b16 (a 4x4 transpose between the lanes of a quad group and their element slots,
    // semantics measured in tools/ubench/tr_probe.hip), so no transposed copy of V is built -- the register transposition that used to
    // be here (all V loads, select-free repacking, 2-byte-pair stores) was about a quarter of the kernel.  32-byte granules of a row
    // are XOR-swizzled with (row >> 1) & 3 (on the SOURCE address, the LDS image stays lane-linear): the 8 rows x 32 B a half-wave
    // reads then fall on 64 distinct banks.
    constexpr int VROWS = att_kp2(NT);
    constexpr int V_ITEMS2 = VROWS * 8;
    constexpr int V_IT3 = (V_ITEMS2 + ATT_THREADS - 1) / ATT_THREADS;
#pragma unroll
    for (int it = 0; it < V_IT3; ++it) {
        const int L = tid + it * ATT_THREADS;              // V_ITEMS2 is a multiple of 128: wave-uniform predicate
        if (L < V_ITEMS2) {
            const int row = L >> 3, p = L & 7;
            const int c = ((((p >> 1) ^ ((row >> 1) & 3)) << 1) | (p & 1));
            const int rc = row < ntok ? row : ntok - 1;
            __builtin_amdgcn_global_load_lds((att_gptr_t)(base + (int64_t)rc * D3 + voff + c * 8),
                                             (att_lptr_t)(sVt + (it * ATT_THREADS + wave * 64) * 8), 16, 0, 0);
        }
    }
}

// NW wavefronts per workgroup: 4 (two workgroups per CU by LDS) or 8 (the 13 query tiles of a ViT head are
// spread over twice the waves: 16 waves per CU hide the LDS / exp latencies better).
// Waves per SIMD the register allocation is planned for.  Two 8-wave workgroups per CU (what the LDS allows) are 4 waves per SIMD = 128
// registers.  Asking for that bound outright made the NT = 13 kernel SPILL (17 dwords; reloads with s_waitcnt vmcnt(0) in front of every
// output store), while the unconstrained allocation of the same code lands at 126 registers and gets the same occupancy -- so NT = 13 is
// left unconstrained; NT = 16 needs the bound (152 registers without it, 118 and no scratch with it).  Checked in the ISA metadata
// (tests/test_build_artifacts.py: no private segment in any attention / GEMM kernel of the product library).
__host__ __device__ constexpr int att_min_waves(int NT, int NW) { return NW >= 7 && NT != 13 ? 4 : 2; }

template <int NT, bool SPLIT, int NW>
__global__ __launch_bounds__(NW * 64, att_min_waves(NT, NW))
void attention_kernel(AttnParams p) {
    constexpr int ATT_THREADS = NW * 64;
    constexpr int NKP = NT * 16;
    constexpr int VS = att_vs(NT);
    constexpr int NU = (NT + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f16* sK = reinterpret_cast<f16*>(smem);
    f16* sVt = sK + NKP * HD;                              // V, row-major [att_kp2(NT)][64] (see stage_kv)
    float* sBias = reinterpret_cast<float*>(sVt + att_kp2(NT) * HD);
    f16* sKl = reinterpret_cast<f16*>(sBias + NKP);
    f16* sVtl = sKl + NKP * HD;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / p.heads, h = blockIdx.x - b * p.heads;
    const int ntok = p.ntok;
    const int D = p.heads * HD, D3 = 3 * D;
    const int64_t tok0 = (int64_t)b * ntok;
    const f16* base_hi = p.qkv_hi + tok0 * D3;
    const f16* base_lo = SPLIT ? p.qkv_lo + tok0 * D3 : nullptr;

    long long t_start = 0, t_staged = 0;
    if (p.dbg) t_start = __builtin_readcyclecounter();
    // key window of this launch (all keys unless launch_attention split a long sequence in two)
    const int key0 = SPLIT ? p.key0 : 0;
    const int kcount = (SPLIT && p.kcount > 0) ? p.kcount : ntok;
    stage_kv<NT, ATT_THREADS>(base_hi + (int64_t)key0 * D3, kcount, D3, D + h * HD, 2 * D + h * HD, sK, sVt, tid, wave);
    if (SPLIT) stage_kv<NT, ATT_THREADS>(base_lo + (int64_t)key0 * D3, kcount, D3, D + h * HD, 2 * D + h * HD, sKl, sVtl, tid, wave);
    for (int k = tid; k < NKP; k += ATT_THREADS) {
        float bias = 0.f;
        if (k >= kcount) bias = -INFINITY;
        else if (p.mask && p.mask[tok0 + key0 + k] == 0) bias = -1e30f;     // HF adds finfo.min to masked keys
        sBias[k] = bias;                                             // (scores live in the log2 domain below; -1e30 / -inf are scale-free)
    }
    // The K tiles arrive by LDS-DMA: nothing but this wave's own vmcnt orders them before the barrier.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (p.dbg) t_staged = __builtin_readcyclecounter();

    const int qi = lane & 15, g = lane >> 4;
    const int nq = (p.q_rows > 0 && p.q_rows < ntok) ? p.q_rows : ntok;
    const int nqt = (nq + 15) >> 4;
    const float sc2 = p.scale * 1.4426950408889634f;
    // Q fragments of the next query tile are fetched while the current one is computed
    f16x8 qn[2], qln[2];
    // Addresses are a wave-uniform base plus ONE 32-bit lane offset (the launcher refuses buffers of 2^31 elements or more): 64-bit lane
    // addresses for the Q loads and the output stores were loop invariants the compiler hoisted, ran out of registers on (128 per wave at
    // this occupancy) and spilled -- every output store of every query tile then sat behind a scratch reload and an s_waitcnt vmcnt(0).
    // (p.q_hi: the single query row of this image lives in a compact [batch][q_ld] buffer -- every lane of the one query tile then reads that row;
    // only row 0 is stored, q_rows == 1)
    const f16* q_base = (!SPLIT && p.q_hi) ? p.q_hi + (int64_t)b * p.q_ld : base_hi;
    const unsigned q_stride = (!SPLIT && p.q_hi) ? 0u : (unsigned)D3;
    auto load_q = [&](int qt) {
        int q = qt * 16 + qi;
        q = q < ntok ? q : ntok - 1;
        const unsigned qo = (unsigned)q * q_stride + (unsigned)(h * HD + g * 8);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qn[ks] = *reinterpret_cast<const f16x8*>(q_base + (qo + ks * 32));
            if (SPLIT) qln[ks] = *reinterpret_cast<const f16x8*>(base_lo + (qo + ks * 32));
        }
    };
    const bool o_blk = p.out_kt > 0;
    const unsigned o_sa = o_blk ? (unsigned)p.out_kt * 8192u : 256u * (unsigned)D, o_sb = o_blk ? 32u : (unsigned)D;
    const unsigned o_ga = o_blk ? 8192u : 32u, o_g0 = (o_blk ? (unsigned)(h * 2) * 8192u : (unsigned)(h * HD)) + (unsigned)(g * 4);
    if (wave < nqt) load_q(wave);
    asm volatile("" : "+v"(qn[0]), "+v"(qn[1]));          // same as at the bottom of the loop: no pending Q load reaches the loop header on any path
    if (SPLIT) asm volatile("" : "+v"(qln[0]), "+v"(qln[1]));
    for (int qt = wave; qt < nqt; qt += NW) {
        const int q = qt * 16 + qi;
        f16x8 qf[2], ql[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { qf[ks] = qn[ks]; if (SPLIT) ql[ks] = qln[ks]; }
        if (qt + NW < nqt) load_q(qt + NW);
        // The K / V^T fragment reads do not depend on the query tile; without the compiler barriers
        // below LICM hoists ALL of them out of the qt loop (hundreds of VGPRs -> scratch spills).
        // Each loop is software-pipelined one step deep by hand instead.
#define KEEP_MEM_BARRIER() asm volatile("" ::: "memory")
        KEEP_MEM_BARRIER();
        f32x4 s[NT];
        f16x8 kf[2][2], kl[2][2];
        auto load_k = [&](int kt, int slot) {
            const int row = kt * 16 + qi;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = row * HD + (((ks * 4 + g) ^ ((row >> 1) & 7)) << 3);
                kf[slot][ks] = *reinterpret_cast<const f16x8*>(sK + off);
                if (SPLIT) kl[slot][ks] = *reinterpret_cast<const f16x8*>(sKl + off);
            }
        };
        load_k(0, 0);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            if (kt + 1 < NT) load_k(kt + 1, (kt + 1) & 1);
            KEEP_MEM_BARRIER();
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (SPLIT) {
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[kt & 1][ks], qf[ks], s[kt], 0, 0, 0);
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt & 1][ks], ql[ks], s[kt], 0, 0, 0);
                }
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt & 1][ks], qf[ks], s[kt], 0, 0, 0);
            }
        }
        // ---- softmax over keys (registers + lanes {l, l^16, l^32, l^48})
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(sBias + kt * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[kt][r] = s[kt][r] * sc2 + bias[r];           // log2-domain score: exp(x) = exp2(x * log2 e)
                mx = fmaxf(mx, s[kt][r]);
            }
            if ((kt & 3) == 3) KEEP_MEM_BARRIER();
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // exp and the fp16 conversion in one pass, two key tiles (one 32-key MFMA operand) at a time: the fp32 scores die as the packed
        // probabilities are born, so the P.V loop below holds 26 registers of P instead of 52 (the kernel has 128 at this occupancy)
        float sum = 0.f;
        f16x8 ph[NU], pl[SPLIT ? NU : 1];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            float a0[4], a1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a0[r] = __builtin_amdgcn_exp2f(s[2 * u][r] - mx);
                a1[r] = (2 * u + 1 < NT) ? __builtin_amdgcn_exp2f(s[(2 * u + 1 < NT) ? 2 * u + 1 : 0][r] - mx) : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) sum += a0[r];          // summation order: key tile by key tile, as the scores are laid out
            if (2 * u + 1 < NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sum += a1[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // probabilities lie in [0, 1]: no saturation needed (the clamp of split_f16 was 2 of the 3 VALU
                // instructions per score here, ~400 cycles per 16-query tile)
                const f16 h0 = (f16)a0[r], h1 = (f16)a1[r];
                ph[u][r] = h0; ph[u][4 + r] = h1;
                if (SPLIT) { pl[u][r] = (f16)(a0[r] - (float)h0); pl[u][4 + r] = (f16)(a1[r] - (float)h1); }
            }
            __builtin_amdgcn_sched_barrier(0);      // one 32-key unit at a time: interleaving the units keeps all fp32 exps alive next to P
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        float inv = 1.0f / sum;

        // ---- O^T = V^T P^T
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        f16x4 va[2][4][2], vb[2][4][2];      // [slot][dt][half]  hi / lo planes
        // lane (qi, g) points at V[32u + 16 half + 4g + qi/4][16 dt + 4 (qi%4) ..+3]; the transpose read hands it
        // V[32u + 16 half + 4g + 0..3][16 dt + qi]: keys 4g..4g+3 of feature qi, the slots P occupies in the B operand
        auto load_v = [&](int u, int slot) {
            const int vrow = 32 * u + 4 * g + (qi >> 2);
            const int vsw = (2 * g + (qi >> 3)) & 3;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int voff = vrow * HD + ((dt ^ vsw) << 4) + ((qi & 3) << 2);
                va[slot][dt][0] = tr_read4(sVt + voff);
                va[slot][dt][1] = tr_read4(sVt + voff + 16 * HD);
                if (SPLIT) {
                    vb[slot][dt][0] = tr_read4(sVtl + voff);
                    vb[slot][dt][1] = tr_read4(sVtl + voff + 16 * HD);
                }
            }
        };
        KEEP_MEM_BARRIER();
        load_v(0, 0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (u + 1 < NU) load_v(u + 1, (u + 1) & 1);
            KEEP_MEM_BARRIER();
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f16x4 v0 = va[u & 1][dt][0], v1 = va[u & 1][dt][1];
                const f16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if (SPLIT) {
                    const f16x4 w0 = vb[u & 1][dt][0], w1 = vb[u & 1][dt][1];
                    const f16x8 vl = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph[u], o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pl[u], o[dt], 0, 0, 0);
                }
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, ph[u], o[dt], 0, 0, 0);
            }
        }
#undef KEEP_MEM_BARRIER
        // The prefetched Q fragments are "used" here, BEFORE this tile's stores go out: the compiler puts its s_waitcnt for them at their
        // first use, and at the top of the next tile that wait would also drain the stores just issued (one in-order counter).  Here only the Q
        // loads themselves are outstanding, and they were issued a whole tile ago.
        asm volatile("" : "+v"(qn[0]), "+v"(qn[1]));
        if (SPLIT) asm volatile("" : "+v"(qln[0]), "+v"(qln[1]));
        if constexpr (SPLIT) {
            // two key windows (256 < ntok <= 512): the first launch parks (unnormalised O, maximum, sum) per query, the second merges:
            // m = max(m1, m2); O = O1 2^(m1 - m) + O2 2^(m2 - m); l likewise -- exactly the softmax over all keys (scores are in the log2 domain)
            const int64_t prow = ((int64_t)(b * p.heads + h) * ntok + (q < ntok ? q : ntok - 1)) * ATT_PART_FLOATS;
            if (p.part_out) {
                if (q < nq) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(p.part_out + prow + dt * 16 + g * 4) = o[dt];
                    if (g == 0) { p.part_out[prow + 64] = mx; p.part_out[prow + 65] = sum; }
                }
                continue;
            }
            if (p.part_in) {
                const float m1 = p.part_in[prow + 64], l1 = p.part_in[prow + 65];
                const float m = fmaxf(m1, mx);
                const float w1 = __builtin_amdgcn_exp2f(m1 - m), w2 = __builtin_amdgcn_exp2f(mx - m);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const f32x4 o1 = *reinterpret_cast<const f32x4*>(p.part_in + prow + dt * 16 + g * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[dt][r] = o1[r] * w1 + o[dt][r] * w2;
                }
                inv = 1.0f / (l1 * w1 + sum * w2);
            }
        }
        if (q < nq) {
            const int mrow = (int)(tok0 + q);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f16x4 oh, ol;
#pragma unroll
                for (int r = 0; r < 4; ++r) { f16 hh, ll; split_f16(o[dt][r] * inv, hh, ll); oh[r] = hh; ol[r] = ll; }
                // blk layout (common.h blk_off, in 32 bits) or row-major, as ONE form: (m / 256) * o_sa + (m % 256) * o_sb + o_g0 + (dt / 2) * o_ga + (dt % 2) * 16,
                // constants picked before the loop (a per-store "which layout" test is a pair of scalar branches)
                const unsigned oo = (unsigned)(mrow >> 8) * o_sa + (unsigned)(mrow & 255) * o_sb + o_g0 + (unsigned)(dt >> 1) * o_ga + (unsigned)(dt & 1) * 16u;
                *reinterpret_cast<f16x4*>(p.out_hi + oo) = oh;
                if (SPLIT) *reinterpret_cast<f16x4*>(p.out_lo + oo) = ol;
                if (!SPLIT && p.cls_hi && q == 0) {           // the CLS row once more, hi + lo, into row b of the compact operand (same layout constants)
                    const unsigned oc = (unsigned)(b >> 8) * o_sa + (unsigned)(b & 255) * o_sb + o_g0 + (unsigned)(dt >> 1) * o_ga + (unsigned)(dt & 1) * 16u;
                    *reinterpret_cast<f16x4*>(p.cls_hi + oc) = oh;
                    *reinterpret_cast<f16x4*>(p.cls_lo + oc) = ol;
                }
            }
        }
    }
    if (p.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0 && blockIdx.x < 65536) {
            long long* d = p.dbg + (size_t)blockIdx.x * 4;
            d[0] = t_start; d[1] = t_staged; d[2] = __builtin_readcyclecounter(); d[3] = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Persistent variant for the image tower (197 tokens = 13 query tiles, no key mask, single fp16 pass): one 16-wave workgroup per CU walks
// the (image, head) pairs with stride gridDim.x.  Waves 0..12 own ONE query tile each (the 13 tiles of a head over 8 waves left three
// waves with one tile and five with two); waves 13..15 issue the LDS-DMA that stages the NEXT pair's K and V into the other half of a
// double buffer while the thirteen compute, so that staging (1.1 of the 2.6 ms per step of the one-pair-per-workgroup kernel) runs under
// the compute instead of in front of it, and the computing waves issue no DMA at all.  Same arithmetic, same summation order per tile as
// attention_kernel<13, false, 8>: bit-identical results.
template <int NT>
__global__ __launch_bounds__(1024, 4)
void attention_pers_kernel(AttnParams p) {
    constexpr int NW = 16, NCW = NT, NLW = NW - NCW;
    static_assert(NLW >= 1, "needs at least one loader wave");
    constexpr int NKP = NT * 16;
    constexpr int NU = (NT + 1) / 2;
    constexpr int BUF_ELEMS = NKP * HD + att_kp2(NT) * HD;           // K image + V image of one (image, head)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f16* sBuf = reinterpret_cast<f16*>(smem);
    float* sBias = reinterpret_cast<float*>(sBuf + 2 * BUF_ELEMS);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntok = p.ntok, heads = p.heads;
    const int D = heads * HD, D3 = 3 * D;
    const int items = p.batch * heads;
    int item = blockIdx.x;
    if (item >= items) return;

    {   // first pair: every wave stages
        const int b = item / heads, h = item - b * heads;
        stage_kv<NT, NW * 64>(p.qkv_hi + (int64_t)b * ntok * D3, ntok, D3, D + h * HD, 2 * D + h * HD, sBuf, sBuf + NKP * HD, tid, wave);
    }
    for (int k = tid; k < NKP; k += NW * 64) sBias[k] = k >= ntok ? -INFINITY : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int qi = lane & 15, g = lane >> 4;
    const float sc2 = p.scale * 1.4426950408889634f;
    const bool o_blk = p.out_kt > 0;
    const unsigned o_sa = o_blk ? (unsigned)p.out_kt * 8192u : 256u * (unsigned)D, o_sb = o_blk ? 32u : (unsigned)D;
    const unsigned o_ga = o_blk ? 8192u : 32u;
    int qrow = wave * 16 + qi;                                     // this lane's query (compute waves)
    const bool q_valid = qrow < ntok;
    qrow = q_valid ? qrow : ntok - 1;
    f16x8 qn[2];
    auto load_q = [&](int it) {
        const int b = it / heads, h = it - b * heads;
        const f16* base = p.qkv_hi + (int64_t)b * ntok * D3;
        const unsigned qo = (unsigned)qrow * (unsigned)D3 + (unsigned)(h * HD + g * 8);
        qn[0] = *reinterpret_cast<const f16x8*>(base + qo);
        qn[1] = *reinterpret_cast<const f16x8*>(base + (qo + 32));
    };
    if (wave < NCW) load_q(item);
    asm volatile("" : "+v"(qn[0]), "+v"(qn[1]));

    for (int cur = 0;; cur ^= 1) {
        const int nxt = item + (int)gridDim.x;
        const bool more = nxt < items;
        const f16* sK = sBuf + cur * BUF_ELEMS;
        const f16* sVt = sK + NKP * HD;
        if (wave >= NCW) {
            if (more) {
                const int b = nxt / heads, h = nxt - b * heads;
                f16* dK = sBuf + (cur ^ 1) * BUF_ELEMS;
                stage_kv<NT, NLW * 64>(p.qkv_hi + (int64_t)b * ntok * D3, ntok, D3, D + h * HD, 2 * D + h * HD, dK, dK + NKP * HD, tid - NCW * 64, wave - NCW);
            }
        } else {
            const int b = item / heads, h = item - b * heads;
            f16x8 qf[2] = {qn[0], qn[1]};
            if (more) load_q(nxt);
#define KEEP_MEM_BARRIER() asm volatile("" ::: "memory")
            KEEP_MEM_BARRIER();
            f32x4 s[NT];
            f16x8 kf[2][2];
            auto load_k = [&](int kt, int slot) {
                const int row = kt * 16 + qi;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    kf[slot][ks] = *reinterpret_cast<const f16x8*>(sK + row * HD + (((ks * 4 + g) ^ ((row >> 1) & 7)) << 3));
            };
            load_k(0, 0);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                if (kt + 1 < NT) load_k(kt + 1, (kt + 1) & 1);
                KEEP_MEM_BARRIER();
                s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt & 1][ks], qf[ks], s[kt], 0, 0, 0);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const f32x4 bias = *reinterpret_cast<const f32x4*>(sBias + kt * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[kt][r] = s[kt][r] * sc2 + bias[r];
                    mx = fmaxf(mx, s[kt][r]);
                }
                if ((kt & 3) == 3) KEEP_MEM_BARRIER();
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
            f16x8 ph[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                float a0[4], a1[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a0[r] = __builtin_amdgcn_exp2f(s[2 * u][r] - mx);
                    a1[r] = (2 * u + 1 < NT) ? __builtin_amdgcn_exp2f(s[(2 * u + 1 < NT) ? 2 * u + 1 : 0][r] - mx) : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) sum += a0[r];
                if (2 * u + 1 < NT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sum += a1[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { ph[u][r] = (f16)a0[r]; ph[u][4 + r] = (f16)a1[r]; }
                __builtin_amdgcn_sched_barrier(0);
            }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            f32x4 o[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            f16x4 va[2][4][2];
            auto load_v = [&](int u, int slot) {
                const int vrow = 32 * u + 4 * g + (qi >> 2);
                const int vsw = (2 * g + (qi >> 3)) & 3;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int voff = vrow * HD + ((dt ^ vsw) << 4) + ((qi & 3) << 2);
                    va[slot][dt][0] = tr_read4(sVt + voff);
                    va[slot][dt][1] = tr_read4(sVt + voff + 16 * HD);
                }
            };
            KEEP_MEM_BARRIER();
            load_v(0, 0);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + 1 < NU) load_v(u + 1, (u + 1) & 1);
                KEEP_MEM_BARRIER();
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const f16x4 v0 = va[u & 1][dt][0], v1 = va[u & 1][dt][1];
                    const f16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, ph[u], o[dt], 0, 0, 0);
                }
            }
#undef KEEP_MEM_BARRIER
            asm volatile("" : "+v"(qn[0]), "+v"(qn[1]));          // the prefetched Q is waited for here, before this tile's stores go out
            if (q_valid) {
                const int mrow = b * ntok + qrow;
                const unsigned o_g0 = (o_blk ? (unsigned)(h * 2) * 8192u : (unsigned)(h * HD)) + (unsigned)(g * 4);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    f16x4 oh;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { f16 hh, ll; split_f16(o[dt][r] * inv, hh, ll); oh[r] = hh; }
                    const unsigned oo = (unsigned)(mrow >> 8) * o_sa + (unsigned)(mrow & 255) * o_sb + o_g0 + (unsigned)(dt >> 1) * o_ga + (unsigned)(dt & 1) * 16u;
                    *reinterpret_cast<f16x4*>(p.out_hi + oo) = oh;
                }
                if (p.cls_hi && qrow == 0) {                  // KEEP_ATTN_PROJ_CLS: the CLS row once more, hi + lo, into row b of the compact operand
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        f16x4 oh, ol;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { f16 hh, ll; split_f16(o[dt][r] * inv, hh, ll); oh[r] = hh; ol[r] = ll; }
                        const unsigned oc = (unsigned)(b >> 8) * o_sa + (unsigned)(b & 255) * o_sb + o_g0 + (unsigned)(dt >> 1) * o_ga + (unsigned)(dt & 1) * 16u;
                        *reinterpret_cast<f16x4*>(p.cls_hi + oc) = oh;
                        *reinterpret_cast<f16x4*>(p.cls_lo + oc) = ol;
                    }
                }
            }
        }
        if (!more) break;
        // the loader waves' DMA must have landed before anyone reads the other buffer; nobody may still read this one when the next
        // staging (into it, one iteration on) begins: one barrier does both
        if (wave >= NCW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        item = nxt;
    }
}

template <int NT>
int launch_pers(const AttnParams& p, hipStream_t s) {
    constexpr size_t bytes = (size_t)2 * ((size_t)NT * 16 * HD + (size_t)att_kp2(NT) * HD) * 2 + (size_t)NT * 16 * 4;
    if (!keep_lds_opt_in(reinterpret_cast<const void*>(&attention_pers_kernel<NT>), bytes)) return -2;     // refused: the caller falls back to one pair per workgroup
    const int ncu = keep_num_cus();
    const int items = p.batch * p.heads;
    hipLaunchKernelGGL((attention_pers_kernel<NT>), dim3(items < ncu ? items : ncu), dim3(1024), bytes, s, p);
    return 0;
}

template <int NT, bool SPLIT, int NW>
int launch_one(const AttnParams& p, hipStream_t s) {
    constexpr size_t bytes = att_lds_bytes(NT, SPLIT);
    if (!keep_lds_opt_in(reinterpret_cast<const void*>(&attention_kernel<NT, SPLIT, NW>), bytes)) return -2;
    hipLaunchKernelGGL((attention_kernel<NT, SPLIT, NW>), dim3(p.batch * p.heads), dim3(NW * 64), bytes, s, p);
    return 0;
}

}  // namespace keepk

int launch_attention(const AttnParams& p_in, hipStream_t s) {
    using namespace keepk;
    AttnParams p = p_in;
    p.key0 = 0; p.kcount = 0; p.part_out = nullptr; p.part_in = nullptr;      // internal fields: set below for the two-window launches only
    const int g_attn_waves = p.tune ? p.tune->attn_waves : 8;     // wavefronts per workgroup for the unsplit 13/16-tile kernels (4 or 8)
#ifdef KEEP_DIAGNOSTICS
    p.dbg = p.tune ? p.tune->dbg : nullptr;
#else
    p.dbg = nullptr;
#endif
    const int nt = (p.ntok + 15) / 16;
    if (p.ntok < 1 || p.batch < 1) return -1;
    if (p.q_hi && (p.q_rows != 1 || p.split)) return -1;          // the compact query buffer holds one row per image, single-pass only
    if ((p.cls_hi != nullptr) != (p.cls_lo != nullptr) || (p.cls_hi && (p.split || p.mask))) return -1;      // CLS-row hi + lo copy: both planes, single-pass image tower only
    // 32-bit lane offsets inside the kernel: qkv rows of one (batch) slice and the whole output plane (padded to 256 rows) stay below 2^31 elements
    if ((int64_t)p.ntok * 3 * p.heads * HD >= (1ll << 31) || ((int64_t)p.batch * p.ntok + 255) / 256 * 256 * p.heads * HD >= (1ll << 31)) return -1;
    if (p.split) {
        if (nt <= 4) return launch_one<4, true, 4>(p, s);
        if (nt <= 8) return launch_one<8, true, 4>(p, s);
        if (nt <= 13) return launch_one<13, true, 4>(p, s);
        if (nt <= 16) return launch_one<16, true, 4>(p, s);
        if (nt <= 32 && p.part_ws && p.part_bytes >= (size_t)p.batch * p.heads * p.ntok * ATT_PART_FLOATS * sizeof(float)) {
            // K / V hi + lo of 512 keys are 256 KiB: two key windows of <= 256, merged in the second launch (same stream: ordered)
            AttnParams a = p, c = p;
            a.key0 = 0; a.kcount = 256; a.part_out = p.part_ws; a.part_in = nullptr;
            c.key0 = 256; c.kcount = p.ntok - 256; c.part_out = nullptr; c.part_in = p.part_ws;
            const int rc = launch_one<16, true, 4>(a, s);
            return rc ? rc : launch_one<16, true, 4>(c, s);
        }
        return -1;
    }
    if (nt <= 4) return launch_one<4, false, 4>(p, s);
    if (nt <= 8) return launch_one<8, false, 4>(p, s);
    // image tower at full width: the persistent double-buffered kernel (attn_waves = 16) when there are more (image, head) pairs than CUs can hold at once
    if (nt == 13 && g_attn_waves == 16 && !p.mask && p.q_rows <= 0 && p.batch * p.heads >= 512 && launch_pers<13>(p, s) == 0) return 0;
    if (nt <= 13) return g_attn_waves == 4 ? launch_one<13, false, 4>(p, s) : launch_one<13, false, 8>(p, s);
    if (nt <= 16) return g_attn_waves == 4 ? launch_one<16, false, 4>(p, s) : launch_one<16, false, 8>(p, s);
    if (nt <= 32) return launch_one<32, false, 4>(p, s);
    return -1;
}
