// fp16 MFMA GEMM, persistent variant: one 8-wave workgroup per CU walks a list of 256x256 tiles and keeps
// its LDS-DMA ring running ACROSS tile boundaries.
//
// Same contract, operand layout (blk) and epilogues as gemm_f16_v2.hip.  What changes is what the timeline
// (tools/gemm_timeline.py) showed to be pure loss in the one-tile-per-workgroup kernel: every tile started
// with an empty ring (3.5-5.5 k cycles of DMA latency, 6-8 % of a K=1024 tile) and a new workgroup launch.
// Here the DMA for the first K slices of tile i+1 is issued during the last K steps of tile i and lands while
// tile i's epilogue runs; the epilogue bounces through a separate 40 KiB of LDS so the ring (3 x 32 KiB) stays
// untouched.
//
// vmcnt bookkeeping (CDNA counts loads, LDS-DMA and stores in one in-order counter):
//   * in the K loop the usual counted wait: before using step s+1, at most one newer DMA group (4 ops) may be
//     outstanding -> s_waitcnt vmcnt(4) (vmcnt(0) when no newer group exists);
//   * at the last step of a tile with a successor, vmcnt(0): the successor's steps 0 and 1 have landed BEFORE
//     the epilogue issues its loads/stores, so the first barrier of the next tile needs no vmcnt wait at all
//     (a counted wait there would have to drain the epilogue's stores);
//   * from the second barrier of the next tile on, vmcnt(4) again: the epilogue's stores are older than the
//     newest DMA group, so they are (harmlessly) included.
#include "gemm_epilogue.h"

namespace keepk {

typedef const __attribute__((address_space(1))) void* v3_gptr_t;
typedef __attribute__((address_space(3))) void* v3_lptr_t;
template <int N> __device__ __forceinline__ void v3_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int V3_THREADS = 512, V3_BK = 32, V3_NSTAGE = 3, V3_G = 4;
constexpr int V3_BUF = 512 * V3_BK;                       // f16 elements per ring stage (256 A rows + 256 W rows)
constexpr int V3_SLAB_BYTES = 5120;                       // per wave: fp32 32x36 (4608 B) or fp16 hi+lo 2 x 32x40 (5120 B)
constexpr size_t V3_LDS_BYTES = (size_t)V3_NSTAGE * V3_BUF * 2 + 8 * V3_SLAB_BYTES;

__device__ __forceinline__ int v3_lds_off(int row, int chunk) { return row * V3_BK + ((chunk ^ ((row >> 2) & 3)) << 3); }

template <int EPI>
__global__ __launch_bounds__(V3_THREADS, 2)
void gemm_f16_v3_kernel(GemmParams p, int total_tiles) {
    constexpr int BM = 256, BN = 256, BK = V3_BK, TM = 4, TN = 2, NSTAGE = V3_NSTAGE, G = V3_G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f16* lds = reinterpret_cast<f16*>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 31, fhi = lane >> 5;

    const int ntn = p.N / BN, mtn = (p.M + BM - 1) / BM, nwg = total_tiles;
    constexpr int BW = 8;
    auto tile_of = [&](int b, int& tm_, int& tn_) {          // same XCD-aware banded order as gemm_f16_v2.hip
        const int xcd = b & 7, slot = b >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        const int full_tiles = (ntn / BW) * BW * mtn;
        if (t < full_tiles) {
            const int band = t / (mtn * BW), rr = t - band * (mtn * BW);
            tm_ = rr / BW; tn_ = band * BW + (rr - tm_ * BW);
        } else {
            const int remw = ntn % BW, rr = t - full_tiles;
            tm_ = rr / remw; tn_ = (ntn / BW) * BW + (rr - tm_ * remw);
        }
    };

    const int KT = p.K / BK;
    const int steps = KT * p.nseg;
    int a_off[2], w_off[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int L = r * V3_THREADS + tid;
        const int row = L >> 2, c = (L & 3) ^ ((row >> 2) & 3);
        a_off[r] = row * BK + c * 8;
        w_off[r] = a_off[r];
    }
    // DMA of K step `ks` of tile (tm_, tn_) into ring buffer `rb`
    auto stage = [&](int tm_, int tn_, int ks, int rb) {
        const int seg = ks / KT, kt = ks - seg * KT;
        const f16* ab = ((seg == 1) ? p.a_lo : p.a_hi) + ((int64_t)tm_ * KT + kt) * 8192;
        const f16* wb = ((seg == 2) ? p.w_lo : p.w_hi) + ((int64_t)tn_ * KT + kt) * 8192;
        f16* sa = lds + rb * V3_BUF;
        f16* sw = sa + BM * BK;
#pragma unroll
        for (int r = 0; r < 2; ++r)
            __builtin_amdgcn_global_load_lds((v3_gptr_t)(ab + a_off[r]), (v3_lptr_t)(sa + (r * V3_THREADS + wave * 64) * 8), 16, 0, 0);
#pragma unroll
        for (int r = 0; r < 2; ++r)
            __builtin_amdgcn_global_load_lds((v3_gptr_t)(wb + w_off[r]), (v3_lptr_t)(sw + (r * V3_THREADS + wave * 64) * 8), 16, 0, 0);
    };

    f16x8 fw0[TN], fa0[TM], fw1[TN], fa1[TM];
    auto read_frags = [&](int rb, int ks, f16x8 (&fw)[TN], f16x8 (&fa)[TM]) {
        const f16* sa = lds + rb * V3_BUF;
        const f16* sw = sa + BM * BK;
#pragma unroll
        for (int i = 0; i < TN; ++i) fw[i] = *reinterpret_cast<const f16x8*>(sw + v3_lds_off(wn * 64 + i * 32 + frow, ks * 2 + fhi));
#pragma unroll
        for (int j = 0; j < TM; ++j) fa[j] = *reinterpret_cast<const f16x8*>(sa + v3_lds_off(wm * 128 + j * 32 + frow, ks * 2 + fhi));
    };
    f32x16 acc[TN][TM];
    auto mfma_head = [&](const f16x8 (&fw)[TN], const f16x8 (&fa)[TM]) {
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[0], fa[j], acc[0][j], 0, 0, 0);
    };
    auto mfma_tail = [&](const f16x8 (&fw)[TN], const f16x8 (&fa)[TM]) {
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[1], fa[j], acc[1][j], 0, 0, 0);
    };
#define V3_PIN() __builtin_amdgcn_sched_barrier(0)

    // epilogue scratch: one 32 x 32 slab per wave, outside the ring
    unsigned char* slab_raw = smem_raw + (size_t)NSTAGE * V3_BUF * 2 + (size_t)wave * V3_SLAB_BYTES;
    constexpr bool F16_OUT = (EPI == EPI_F16 || EPI == EPI_GELU_F16);

    int b = blockIdx.x;
    if (b >= nwg) return;
    int tm, tn;
    tile_of(b, tm, tn);
    int ring = 0;                              // ring buffer that holds step 0 of the current tile
    // prologue of the FIRST tile only
    stage(tm, tn, 0, 0);
    if (steps > 1) stage(tm, tn, 1, 1);
    bool first = true;

    while (true) {
        const int nb = b + gridDim.x;
        const bool has_next = nb < nwg;
        int tm2 = 0, tn2 = 0;
        if (has_next) tile_of(nb, tm2, tn2);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        if (first) {
            if (steps > 1) v3_wait_vmcnt<G>(); else v3_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (steps > 2) stage(tm, tn, 2, 2);
            else if (has_next && steps == 2) stage(tm2, tn2, 0, 2);
        }
        // (for later tiles steps 0, 1 landed before the previous epilogue and step 2 is already in flight)
        read_frags(ring, 0, fw0, fa0);

        f32x4 ep_bias[TN][4];                  // epilogue constants (bias fragments, or bias + LayerScale), loaded at the last K step
        int rb = ring;                         // ring buffer of step s
        for (int s = 0; s < steps; ++s) {
            const int rb1 = rb + 1 == NSTAGE ? 0 : rb + 1;       // buffer of step s+1
            mfma_head(fw0, fa0);
            V3_PIN();
            read_frags(rb, 1, fw1, fa1);
            V3_PIN();
            mfma_tail(fw0, fa0);
            V3_PIN();
            const bool last = (s + 1 == steps);
            if (last) {
                // epilogue constants are requested BEFORE the final waits: vmcnt retires in order, so a load issued
                // after the successor's DMA could only be consumed once that DMA has landed
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    if (F16_OUT) {
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg)
                            ep_bias[i][rg] = *reinterpret_cast<const f32x4*>(p.bias + tn * BN + wn * 64 + i * 32 + 8 * rg + 4 * fhi);
                    } else {
                        ep_bias[i][0] = *reinterpret_cast<const f32x4*>(p.bias + tn * BN + wn * 64 + i * 32 + (lane & 7) * 4);
                        if (EPI == EPI_RESID_LS) ep_bias[i][1] = *reinterpret_cast<const f32x4*>(p.ls + tn * BN + wn * 64 + i * 32 + (lane & 7) * 4);
                    }
                }
            }
            if (!last || has_next) {
                // the data used after this barrier: step s+1 of this tile, or nothing new (last step: the barrier only
                // frees buffer rb for the successor's step 2)
                if (last) v3_wait_vmcnt<0>();                                     // successor's steps 0, 1 landed (see header)
                else if (!(s == 0 && !first)) {
                    // is there a DMA group newer than step s+1's?  (step s+2 of this tile, or a successor step)
                    const bool newer = (s + 2 < steps) || has_next;
                    if (newer) v3_wait_vmcnt<G>(); else v3_wait_vmcnt<0>();
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                V3_PIN();
                // refill the buffer just freed (rb) with global step s+3
                const int ns = s + NSTAGE;
                if (ns < steps) stage(tm, tn, ns, rb);
                else if (has_next && ns - steps < steps) stage(tm2, tn2, ns - steps, rb);
            }
            mfma_head(fw1, fa1);
            V3_PIN();
            if (!last) read_frags(rb1, 0, fw0, fa0);
            V3_PIN();
            mfma_tail(fw1, fa1);
            rb = rb1;
        }
        // ring position of the successor's step 0 = buffer after the last step's
        ring = rb;

        // ------------------------------------------------------------------ epilogue (32 x 32 slabs)
        const int m0 = tm * BM, n0 = tn * BN;
        if (F16_OUT) {
            constexpr int P16 = 40;                                  // fp16 pitch: 80-byte rows keep ds_read_b128 aligned
            f16* sh = reinterpret_cast<f16*>(slab_raw);
            f16* sl = sh + 32 * P16;
            const int ocol = (lane & 3) * 8, orow = lane >> 2;       // 4 lanes per 32-col row, 16 rows per instruction
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int ncol0 = n0 + wn * 64 + i * 32;
                const f32x4 (&bfrag)[4] = ep_bias[i];
#pragma unroll
                for (int j = 0; j < TM; ++j) {
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        f32x2 a = {acc[i][j][rg * 4 + 0] + bfrag[rg][0], acc[i][j][rg * 4 + 1] + bfrag[rg][1]};
                        f32x2 c = {acc[i][j][rg * 4 + 2] + bfrag[rg][2], acc[i][j][rg * 4 + 3] + bfrag[rg][3]};
                        if (EPI == EPI_GELU_F16) {
                            if (p.out_lo) { a = gelu_fast2(a); c = gelu_fast2(c); }
                            else { a = gelu_fast2_fp16(a); c = gelu_fast2_fp16(c); }
                        }
                        f16x4 h, l;
                        f16 hh, ll;
                        split_f16(a[0], hh, ll); h[0] = hh; l[0] = ll;
                        split_f16(a[1], hh, ll); h[1] = hh; l[1] = ll;
                        split_f16(c[0], hh, ll); h[2] = hh; l[2] = ll;
                        split_f16(c[1], hh, ll); h[3] = hh; l[3] = ll;
                        const int so = frow * P16 + 8 * rg + 4 * fhi;
                        *reinterpret_cast<f16x4*>(sh + so) = h;
                        if (p.out_lo) *reinterpret_cast<f16x4*>(sl + so) = l;
                    }
                    const int mbase = m0 + wm * 128 + j * 32;
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int r = it * 16 + orow, m = mbase + r;
                        const f16x8 h = *reinterpret_cast<const f16x8*>(sh + r * P16 + ocol);
                        if (m < p.M) {
                            const int64_t o = p.out_kt > 0 ? blk_off(m, ncol0 + ocol, p.out_kt) : (int64_t)m * p.N + ncol0 + ocol;
                            *reinterpret_cast<f16x8*>(p.out_hi + o) = h;
                            if (p.out_lo) *reinterpret_cast<f16x8*>(p.out_lo + o) = *reinterpret_cast<const f16x8*>(sl + r * P16 + ocol);
                        }
                    }
                }
            }
        } else {
            constexpr int P32 = 36;
            float* sf = reinterpret_cast<float*>(slab_raw);
            const int ocol = (lane & 7) * 4, orow = lane >> 3;       // 8 lanes per 32-col row, 8 rows per instruction
            f32x4 res[2][4];
            int64_t oo[2][4];
            auto load_res = [&](int i, int j, f32x4 (&rr)[4], int64_t (&o4)[4]) {
                const int mb = m0 + wm * 128 + j * 32, nc = n0 + wn * 64 + i * 32 + ocol;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int m = mb + it * 8 + orow;
                    const int mc = m < p.M ? m : p.M - 1;
                    int prow; int64_t orow_;
                    gemm_epilogue_row<EPI>(p, mc, prow, orow_);
                    o4[it] = orow_ * p.N + nc;
                    if (EPI == EPI_PATCH) rr[it] = *reinterpret_cast<const f32x4*>(p.pos + (int64_t)prow * p.N + nc);
                    else rr[it] = *reinterpret_cast<const f32x4*>(p.resid + o4[it]);
                }
            };
            load_res(0, 0, res[0], oo[0]);
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int nc = n0 + wn * 64 + i * 32 + ocol;
                const f32x4 bias4 = ep_bias[i][0];
                f32x4 ls4 = {1.f, 1.f, 1.f, 1.f};
                if (EPI == EPI_RESID_LS) ls4 = ep_bias[i][1];
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int q = i * TM + j;                              // pass index 0..7, double-buffer by parity
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rg * 4 + e];
                        *reinterpret_cast<f32x4*>(sf + frow * P32 + 8 * rg + 4 * fhi) = v;
                    }
                    if (q + 1 < TN * TM) load_res((q + 1) / TM, (q + 1) % TM, res[(q + 1) & 1], oo[(q + 1) & 1]);
                    const int mbase = m0 + wm * 128 + j * 32;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int r = it * 8 + orow;
                        f32x4 x = *reinterpret_cast<const f32x4*>(sf + r * P32 + ocol);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            x[e] += bias4[e];
                            x[e] = (EPI == EPI_RESID_LS) ? res[q & 1][it][e] + ls4[e] * x[e] : res[q & 1][it][e] + x[e];
                        }
                        if (mbase + r < p.M) {
                            float* dst = (EPI == EPI_RESID_F32) ? p.out_f32 : p.resid;
                            *reinterpret_cast<f32x4*>(dst + oo[q & 1][it]) = x;
                        }
                    }
                }
            }
        }
        if (!has_next) break;
        b = nb; tm = tm2; tn = tn2;
        first = false;
    }
#undef V3_PIN
}

}  // namespace keepk

static int g_v3_cus = 0;

// returns 0 if launched, 1 if the shape is not covered (N % 256, K % 32, fewer than 3 K steps)
int launch_gemm_f16_v3(const GemmParams& p, int epi, hipStream_t s) {
    using namespace keepk;
    if (p.N % 256 || p.K % V3_BK || (p.K / V3_BK) * p.nseg < 3) return 1;
    if (!g_v3_cus) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
        g_v3_cus = prop.multiProcessorCount;
#define KEEP_SET_ATTR(E) if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_v3_kernel<E>), \
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)V3_LDS_BYTES) != hipSuccess) return 1;
        KEEP_SET_ATTR(EPI_F16) KEEP_SET_ATTR(EPI_GELU_F16) KEEP_SET_ATTR(EPI_RESID_LS) KEEP_SET_ATTR(EPI_PATCH) KEEP_SET_ATTR(EPI_RESID_F32)
#undef KEEP_SET_ATTR
    }
    const int tiles = (p.N / 256) * ((p.M + 255) / 256);
    const int grid = tiles < g_v3_cus ? tiles : g_v3_cus;
    dim3 g(grid), b(V3_THREADS);
    switch (epi) {
        case EPI_F16:      hipLaunchKernelGGL(gemm_f16_v3_kernel<EPI_F16>, g, b, V3_LDS_BYTES, s, p, tiles); break;
        case EPI_GELU_F16: hipLaunchKernelGGL(gemm_f16_v3_kernel<EPI_GELU_F16>, g, b, V3_LDS_BYTES, s, p, tiles); break;
        case EPI_RESID_LS: hipLaunchKernelGGL(gemm_f16_v3_kernel<EPI_RESID_LS>, g, b, V3_LDS_BYTES, s, p, tiles); break;
        case EPI_PATCH:    hipLaunchKernelGGL(gemm_f16_v3_kernel<EPI_PATCH>, g, b, V3_LDS_BYTES, s, p, tiles); break;
        default:           hipLaunchKernelGGL(gemm_f16_v3_kernel<EPI_RESID_F32>, g, b, V3_LDS_BYTES, s, p, tiles); break;
    }
    return 0;
}
