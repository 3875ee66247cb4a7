// fp16 MFMA GEMM, large-tile variant: 256 x BN tiles, K step 32, 8 wavefronts, direct global->LDS DMA.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]   (same contract and epilogues as gemm_f16.hip)
//
// Structure (MI355X guide: "glds, 2 LDS buffers, BK=64, one barrier per K tile"):
//   * workgroup = 512 threads = 8 waves, WM x WN; each wave owns (256/WM) x (BN/WN) outputs as
//     32x32x16 MFMA tiles (operand swap as in v1: lanes hold 4 consecutive n of one m)
//   * both operands live in HBM in the K-blocked "blk" layout (common.h) and are staged with
//     global_load_lds_dwordx4 (16 B per lane straight into LDS, no VGPR round trip).  The DMA writes LDS
//     linearly (wave base + lane*16), so the XOR swizzle that makes ds_read_b128 conflict-free is
//     applied to the per-lane GLOBAL source address instead: LDS slot (row, s) holds logical 16-B
//     chunk s ^ ((row>>2)&3) of that 64-byte row
//   * a ring of NSTAGE LDS buffers of BK = 32 (256 + BN rows x 64 B each).  One K tile in flight is
//     latency-bound (a 64 KiB DMA burst takes ~1.9 us to land, measured: that alone set the step
//     time), so NSTAGE-1 tiles are kept in flight: the wait before using tile s is a COUNTED
//     s_waitcnt vmcnt(G*(NSTAGE-2)) followed by a raw s_barrier (a __syncthreads() would drain the
//     DMA queue to zero), and the DMA for tile s+NSTAGE-1 is issued right after that barrier
//   * nseg == 3: hi/lo split product through the same accumulators (strict precision)
//   * COMP = 2: after the fp16 pass, the two first-order correction terms on the MX-fp4 pipe through the same accumulators (phase 2);
//     COMP = 1: the W_lo A_hi term only (half the MFMAs and DMA bytes of phase 2, and the producers of A skip the lo plane)
//   * PERS: one workgroup per CU walks the tile sequence; the next tile's first three K steps are staged from the tail of the K loop
//   * the K loop is written for its ISA: no branch but the back-edge, operand pointers carried from step to step, DMA addresses held in the
//     scalar-base + 32-bit-lane-offset form (empty asm on the offsets), loops aligned to 64 B by the build -- every one of these was measured
//     (DESIGN.md section 4); tests/test_build_artifacts.py checks that no kernel of the library spills
#include "gemm_epilogue.h"
#include "quant4.h"

// Cache-policy bits of the two LDS-DMA streams (aux operand of global_load_lds: 2 = nt).  Tuning knobs for
// tools/ab experiments (KEEP_BUILD_DEFINES); both default to the plain policy.
#ifndef KEEP_A_AUX
#define KEEP_A_AUX 0
#endif
#ifndef KEEP_W_AUX
#define KEEP_W_AUX 0
#endif
namespace keepk {

constexpr int V2_BM = 256, V2_BK = 32;

// 64-byte LDS rows (4 slots of 16 B): slot ^= (row>>2)&3 spreads any 16 consecutive rows over all
// 16 distinct (row&3, slot) bank positions -> conflict-free ds_read_b128
__device__ int g_swz_mask_dummy;
__device__ __forceinline__ int v2_swz(int row, int mask = 3) { return (row >> 2) & mask; }
__device__ __forceinline__ int v2_lds_off(int row, int chunk, int mask = 3) {
    return row * V2_BK + ((chunk ^ v2_swz(row, mask)) << 3);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// <BN, WM, WN, NSTAGE>: 256x256 tile / 8 waves / 4-stage ring = one workgroup per CU (128 KiB LDS);
//                        256x128 tile / 4 waves / 3-stage ring = TWO workgroups per CU (72 KiB LDS each): one
//                        workgroup's prologue/epilogue then runs under the other's MFMA loop.
// waves per SIMD the register budget is planned for: a wave tile of more than 128 accumulators (4 waves x 128x128)
// needs the whole 512-entry file of its SIMD
constexpr int v2_waves_per_simd(int BN, int WM, int WN) { return (V2_BM / WM / 32) * (BN / WN / 32) * 16 > 128 ? 1 : 2; }

typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// phase-2 LDS ring (one K = 64 chunk per stage): the 16 KiB of A planes + 16 KiB of W planes of stage b sit exactly on phase-1 stage b
// (32 KiB each), the 1 KiB + 1 KiB of block scales behind the phase-1 ring -- so the first chunks can be staged into phase-1 stages as
// those retire, under the last fp16 steps.
constexpr int V2_ST2 = 32768;
constexpr int V2_SC2 = 2048;
constexpr int V2_NST2 = 4;
constexpr int V2_SC2_BASE = V2_NST2 * V2_ST2;
constexpr int V2_G2 = 5;             // phase-2 DMA instructions per wave per chunk
constexpr int V2_PERS_LDS = 160 * 1024;   // persistent kernel, fp32 epilogue: 3 prefetched stages (96 KiB) + 8 epilogue slabs of 8 KiB: all of the CU's LDS
constexpr int V2_PERS_LDS_F16 = 96 * 1024 + 8 * 4608;   // fp16 epilogues: one padded fp16 plane per wave (132 KiB: a LayerNorm workgroup of the other lane still fits on the CU)

// PERS ("persistent"): one workgroup per CU walks the tile sequence with stride gridDim.x and stages the first three K steps of its NEXT tile into the
// ring stages that retire during the last steps of the current one, so they land while the epilogue runs (which then bounces through the
// fourth stage and the 32 KiB of LDS behind the ring).  Same arithmetic, same tile order per XCD (tile t and t + 256 map to the same XCD).
template <int BN, int WM, int WN, int NSTAGE, int EPI, int COMP = 0, bool PERS = false>
__global__ __launch_bounds__(WM * WN * 64, v2_waves_per_simd(BN, WM, WN))
void gemm_f16_v2_kernel(GemmParams p) {
    static_assert(!PERS || (BN == 256 && NSTAGE == 4 && COMP == 0 && (EPI == EPI_F16 || EPI == EPI_GELU_F16 || EPI == EPI_RESID_LS)),
                  "the persistent walk is written for the plain 256x256 / 4-stage kernel");
    constexpr int V2_THREADS = WM * WN * 64;
    constexpr int BM = V2_BM, BK = V2_BK;
    constexpr int TM = BM / WM / 32;            // MFMA tiles per wave along m
    constexpr int TN = BN / WN / 32;            // along n
    constexpr int SLOTS = BK / 8;                    // 16-B slots per LDS row
    constexpr int A_ROUNDS = BM * SLOTS / V2_THREADS;   // DMA instructions per thread per K tile
    constexpr int B_ROUNDS = BN * SLOTS / V2_THREADS;
    constexpr int G = A_ROUNDS + B_ROUNDS;
    constexpr int BUF_ELEMS = (BM + BN) * BK;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f16* lds = reinterpret_cast<f16*>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // tile coordinates: consecutive workgroups share the A panel (same m tile, different n tile)
    // XCD-aware, L2-blocked tile order.  Hardware places workgroup b on XCD b % 8 (used for speed only,
    // never for correctness).  Each XCD gets a contiguous run of the tile sequence (bijective for any
    // grid size), and the sequence walks the tile grid in column bands of BW n-tiles, n fastest, then m:
    // the ~32 workgroups resident on one XCD cover a ~4 x 8 patch of tiles, so per K step that XCD's L2
    // fetches 4 A slices + 8 W slices for 64 slice reads (measured LDS-DMA hit rate 79 %).  A band's W
    // (4 MB at K = 1024) does not survive in the 4 MiB L2 from one round to the next; it comes back from
    // the Infinity Cache (profiles/README.md).
    const int ntn = p.N / BN;
    const int mtn = (p.M + BM - 1) / BM;
    // split-K (EPI_PARTIAL): the grid is the tile sequence repeated once per K slice
    const int nwg = (EPI == EPI_PARTIAL || PERS) ? ntn * mtn : (int)gridDim.x;
    const int zsplit = (EPI == EPI_PARTIAL) ? (int)blockIdx.x / nwg : 0;
    const int bidx = (EPI == EPI_PARTIAL) ? (int)blockIdx.x - zsplit * nwg : (int)blockIdx.x;
#ifndef KEEP_BAND_COLS
#define KEEP_BAND_COLS 2048
#endif
    constexpr int BW = KEEP_BAND_COLS / BN;   // band of n-tiles that share an A panel on one XCD (8 tiles of 256: measured 2.5 % better than 4 on fc1)
    auto tile_of = [&](int b, int& tm_, int& tn_) {
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
    int tm, tn;
    tile_of(bidx, tm, tn);
    int m0 = tm * BM;
    int n0 = tn * BN;

    const int swz_mask = (p.ablate & 4) ? 0 : 3;     // diagnostics: ablate&4 disables the swizzle on both sides
    // ---- DMA source offsets.  Operands are in blk layout (common.h): the 256 x 32 slice of one operand
    // for one K step is 16 KiB contiguous, so LDS slot L = round*512 + tid (row L/4, slot L%4) reads
    // byte (L/4)*64 + ((L%4) ^ swz)*16 of the slice: every DMA instruction covers 1 KiB of contiguous HBM.
    const int KT = p.K / BK;
    int a_off[A_ROUNDS], w_off[B_ROUNDS];
#pragma unroll
    for (int r = 0; r < A_ROUNDS; ++r) {
        const int L = r * V2_THREADS + tid;
        const int row = L / SLOTS, c = (L % SLOTS) ^ v2_swz(row, swz_mask);
        a_off[r] = row * BK + c * 8;
    }
#pragma unroll
    for (int r = 0; r < B_ROUNDS; ++r) {
        const int L = r * V2_THREADS + tid;
        const int row = L / SLOTS, c = (L % SLOTS) ^ v2_swz(row, swz_mask);
        w_off[r] = ((n0 & 255) + row) * BK + c * 8;
    }
    int64_t a_tile = (int64_t)(m0 >> 8) * KT * 8192;
    int64_t w_tile = (int64_t)(n0 >> 8) * KT * 8192;
    // persistent walk: the tile after this one (same workgroup), whose first K steps are staged from the tail of the current K loop
    int tile_cur = bidx, m0_n = 0, n0_n = 0;
    int64_t a_tile_n = 0, w_tile_n = 0;
    bool has_next = false, first_tile = true;

    int kt0 = 0, ktiles = KT;
    if (EPI == EPI_PARTIAL) {
        const int per = (KT + p.ksplit - 1) / p.ksplit;
        kt0 = zsplit * per;
        ktiles = (kt0 + per < KT ? kt0 + per : KT) - kt0;
    }
    const int steps = ktiles * p.nseg;

    // stage() is called for s = 0, 1, 2, ... in order, so the operand pointers of "the next step to stage" are carried along (a step moves
    // them 16 KiB; a segment boundary of the split product swaps the planes) instead of being derived from s: the division and selects
    // that cost were ~30 dependent scalar instructions in the middle of every K step, in a wave that issues no MFMA meanwhile.
    const f16* st_a = p.a_hi + a_tile + (int64_t)kt0 * 8192;
    const f16* st_w = p.w_hi + w_tile + (int64_t)kt0 * 8192;
    int st_left = ktiles, st_seg = 0;
    auto stage = [&](int, int buf) {
        const f16* ab = st_a;
        const f16* wb = st_w;
        st_a += 8192; st_w += 8192;
        if (!(PERS || COMP != 0) && --st_left == 0) {             // (persistent and compensated launches are single-pass: no segment boundary, no branch)
            ++st_seg; st_left = ktiles;
            st_a = ((st_seg == 1) ? p.a_lo : p.a_hi) + a_tile + (int64_t)kt0 * 8192;
            st_w = ((st_seg == 2) ? p.w_lo : p.w_hi) + w_tile + (int64_t)kt0 * 8192;
        }
        f16* sa = lds + buf * BUF_ELEMS;
        f16* sw = sa + BM * BK;
        // lane offsets through an empty asm: the address is then formed at the instruction as scalar base + 32-bit lane offset.  Left to itself
        // the compiler strength-reduces the loop to 64-bit lane pointers + a running scalar offset: 4 v_lshl_add_u64 per K step and the
        // vector-address form of the DMA instruction -- measured 3.7 % slower end to end.
        unsigned ao[A_ROUNDS], wo[B_ROUNDS];                 // BYTE offsets, unsigned: zero-extended lane offset + scalar base is the form that maps to the instruction
#pragma unroll
        for (int r = 0; r < A_ROUNDS; ++r) { ao[r] = (unsigned)a_off[r] * 2u; asm volatile("" : "+v"(ao[r])); }
#pragma unroll
        for (int r = 0; r < B_ROUNDS; ++r) { wo[r] = (unsigned)w_off[r] * 2u; asm volatile("" : "+v"(wo[r])); }
        const char* abb = reinterpret_cast<const char*>(ab);
        const char* wbb = reinterpret_cast<const char*>(wb);
#pragma unroll
        for (int r = 0; r < A_ROUNDS; ++r)
            __builtin_amdgcn_global_load_lds((gptr_t)(abb + ao[r]), (lptr_t)(sa + (r * V2_THREADS + wave * 64) * 8), 16, 0, KEEP_A_AUX);
#ifdef KEEP_DIAGNOSTICS
        // what would a kernel gain that did not stream W through LDS?  ablate & 16: same number of DMA instructions, a quarter of the bytes
        if (p.ablate & 16) {
#pragma unroll
            for (int r = 0; r < B_ROUNDS; ++r)
                __builtin_amdgcn_global_load_lds((gptr_t)(wbb + wo[r]), (lptr_t)(sw + (r * V2_THREADS + wave * 64) * 8), 4, 0, KEEP_W_AUX);
            return;
        }
#endif
#pragma unroll
        for (int r = 0; r < B_ROUNDS; ++r)
            __builtin_amdgcn_global_load_lds((gptr_t)(wbb + wo[r]), (lptr_t)(sw + (r * V2_THREADS + wave * 64) * 8), 16, 0, KEEP_W_AUX);
    };
    auto stage_next = [&](int kt, int buf) {             // PERS: K step kt of the NEXT tile (one fp16 pass, no K split)
        const f16* ab = p.a_hi + a_tile_n + (int64_t)kt * 8192;
        const f16* wb = p.w_hi + w_tile_n + (int64_t)kt * 8192;
        f16* sa = lds + buf * BUF_ELEMS;
        f16* sw = sa + BM * BK;
        // the lane offsets go through an empty asm so that the addresses are formed here (scalar base + 32-bit lane offset) and not hoisted out
        // of the K loop as 64-bit lane addresses (they are invariant in it): that costs 32 registers the kernel does not have
#pragma unroll
        for (int r = 0; r < A_ROUNDS; ++r) {
            unsigned o = (unsigned)a_off[r] * 2u;
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(ab) + o), (lptr_t)(sa + (r * V2_THREADS + wave * 64) * 8), 16, 0, KEEP_A_AUX);
        }
#pragma unroll
        for (int r = 0; r < B_ROUNDS; ++r) {
            unsigned o = (unsigned)w_off[r] * 2u;
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(wb) + o), (lptr_t)(sw + (r * V2_THREADS + wave * 64) * 8), 16, 0, KEEP_W_AUX);
        }
    };

    f32x16 acc[TN][TM];
    long long t_start = 0, t_first = 0, t_loop = 0;
#ifdef KEEP_DIAGNOSTICS
    // ablate >> 8 = D: workgroups start (slot & 3) * D * 1024 cycles late -- four phase groups inside every XCD, so that their epilogues do not coincide
    // (is a tile's epilogue shorter when the chip is not in its epilogue all at once?  tools/gemm_timeline.py)
    if (p.ablate >> 8) {
        const long long wait = (long long)((blockIdx.x >> 3) & 3) * (p.ablate >> 8) * 1024, t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
#endif
    if (p.dbg) t_start = __builtin_readcyclecounter();
  for (;;) {                                               // one pass per tile; a non-persistent kernel leaves after the first
    if constexpr (PERS) {
        // The tail of the K loop ALWAYS stages three K steps of "the next tile" (for the last tile of this workgroup: of the tile itself again,
        // 96 KiB nobody reads), so that the K loop has one shape and one set of vmcnt counts: a branch around those steps, with 128
        // accumulators live across it, made the register allocator shuffle accumulators through scratch at the join.
        const int tnx = tile_cur + (int)gridDim.x;
        has_next = tnx < nwg;
        int tm2, tn2;
        tile_of(has_next ? tnx : tile_cur, tm2, tn2);
        m0_n = tm2 * BM; n0_n = tn2 * BN;
        a_tile_n = (int64_t)(m0_n >> 8) * KT * 8192;
        w_tile_n = (int64_t)(n0_n >> 8) * KT * 8192;
    }
    // ---- software-pipelined main loop -------------------------------------------------------------
    // Per K tile (32 deep) a wave runs two groups of TN*TM MFMAs, on fragment sets R0 (k 0..15) and R1
    // (k 16..31).  The single barrier of a step sits BETWEEN the two groups:
    //     MFMA(R0[s]) ; vmcnt: tile s+1 landed ; lgkmcnt(0) ; barrier ;
    //     DMA tile s+NSTAGE -> stage s%NSTAGE ; read R0[s+1] ; MFMA(R1[s]) ; read R1[s+1]
    // so the LDS reads for the next group and the DMA issue run under the other group's MFMAs, and a
    // wave waiting at the barrier waits while its SIMD partner still feeds the matrix pipe.
    const int frow = lane & 31, fhi = lane >> 5;
    f16x8 fw0[TN] = {}, fa0[TM], fw1[TN] = {}, fa1[TM];
    auto read_frags = [&](int st, int ks, f16x8 (&fw)[TN], f16x8 (&fa)[TM]) {
        const f16* sa = lds + st * BUF_ELEMS;
        const f16* sw = sa + BM * BK;
#ifdef KEEP_DIAGNOSTICS
        if (!(p.ablate & 8))         // ablate & 8: the W fragments are not re-read from LDS (a third of the ds_read traffic)
#endif
#pragma unroll
        for (int i = 0; i < TN; ++i)
            fw[i] = *reinterpret_cast<const f16x8*>(sw + v2_lds_off(wn * (TN * 32) + i * 32 + frow, ks * 2 + fhi, swz_mask));
#pragma unroll
        for (int j = 0; j < TM; ++j)
            fa[j] = *reinterpret_cast<const f16x8*>(sa + v2_lds_off(wm * (TM * 32) + j * 32 + frow, ks * 2 + fhi, swz_mask));
    };
    // MFMA group split in a head (first row of tiles) and a tail, so that the LDS reads / DMA issue
    // for the NEXT group can be pinned between them: they then never sit in front of a wait.
    // HM MFMAs of a group before its memory instructions, the rest after.  Swept twice (before and after the K-loop clean-up) on the 8-MFMA
    // groups of the 256x256 tile: 0 / 1 / 2 / 3 / 4 / 5 / 6 / 7 / 8 -> -4.1 / -1.6 / -1.4 / -0.6 / 0 / +0.3 / 0 / -0.6 / -1.4 % end to end.
    constexpr int HM = (TN * TM == 8) ? 5 : TM;
    auto mfma_head = [&](const f16x8 (&fw)[TN], const f16x8 (&fa)[TM]) {
#pragma unroll
        for (int m = 0; m < HM; ++m)
            acc[m / TM][m % TM] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[m / TM], fa[m % TM], acc[m / TM][m % TM], 0, 0, 0);
    };
    auto mfma_tail = [&](const f16x8 (&fw)[TN], const f16x8 (&fa)[TM]) {
#pragma unroll
        for (int m = HM; m < TN * TM; ++m)
            acc[m / TM][m % TM] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[m / TM], fa[m % TM], acc[m / TM][m % TM], 0, 0, 0);
    };
    auto mfma_group = [&](const f16x8 (&fw)[TN], const f16x8 (&fa)[TM]) { mfma_head(fw, fa); mfma_tail(fw, fa); };
    // phase-2 (MX-fp4 correction terms, COMP only) operand stream; defined here because its first chunks are issued from the fp16 loop
    const unsigned char* aq = nullptr; const unsigned char* wq = nullptr; const unsigned char* sq = nullptr;
    if constexpr (COMP != 0) {
        aq = p.a_q + (int64_t)(m0 >> 8) * KT * 8192;
        wq = p.w_q + (int64_t)(n0 >> 8) * KT * 8192;
        // scales, wave-uniform (lanes add 4 * lane).  Two terms: a chunk is two 32-k blocks x two planes of 256 B per operand, wave & 3 = block * 2 + plane.
        // One term: a chunk is four 32-k blocks of ONE plane (A: plane 0 = Q(A_hi), W: plane 1 = Q(W_lo)), wave & 3 = block.
        if constexpr (COMP == 1) sq = (wave < 4 ? p.a_sc + (int64_t)(m0 >> 8) * KT * 512 : p.w_sc + (int64_t)(n0 >> 8) * KT * 512 + 256) + (wave & 3) * 512;
        else sq = (wave < 4 ? p.a_sc + (int64_t)(m0 >> 8) * KT * 512 : p.w_sc + (int64_t)(n0 >> 8) * KT * 512) + (wave & 3) * 256;
    }
    // stage2() is called for chunks 0, 1, 2, ... in order (the first three from the tail of the fp16 loop): running scalar bases, one unsigned
    // 32-bit lane offset per instruction kept opaque to the optimiser -- otherwise the addresses become 64-bit lane pointers (hoisted out of the
    // fp16 loop and spilled, or strength-reduced inside the fp4 loop into vector-address DMA instructions + 64-bit VALU adds)
    const unsigned char* aq_run = aq; const unsigned char* wq_run = wq; const unsigned char* sq_run = sq;
    auto stage2 = [&](int, int buf) {
        unsigned char* sb = smem_raw + buf * V2_ST2;
        unsigned v4 = lane * 4;
        asm volatile("" : "+v"(v4));
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            // one opaque copy of the lane offset per instruction: a shared one gets added to the running base ONCE, as a 64-bit lane pointer
            // two terms: round r = 32-k block r of the chunk, both planes (8 KiB contiguous per operand).  One term: the chunk covers K = 128; round r
            // fetches plane 0 (A) / plane 1 (W) of blocks 2r (threads 0..255) and 2r + 1 (threads 256..511): 2 x 4 KiB pieces, the LDS image stays linear
            unsigned va = COMP == 1 ? (tid >> 8) * 8192 + (tid & 255) * 16 : tid * 16, vw = COMP == 1 ? va + 4096 : va;
            const unsigned char* ab = aq_run + r * (COMP == 1 ? 16384 : 8192);
            const unsigned char* wb = wq_run + r * (COMP == 1 ? 16384 : 8192);
            asm volatile("" : "+v"(va), "+v"(vw), "+s"(ab), "+s"(wb));      // (and the scalar sums stay scalar)
            __builtin_amdgcn_global_load_lds((gptr_t)(ab + va), (lptr_t)(sb + (r * 512 + wave * 64) * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(wb + vw), (lptr_t)(sb + 16384 + (r * 512 + wave * 64) * 16), 16, 0, 0);
        }
        __builtin_amdgcn_global_load_lds((gptr_t)(sq_run + v4), (lptr_t)(smem_raw + V2_SC2_BASE + buf * V2_SC2 + (wave >> 2) * 1024 + (wave & 3) * 256), 4, 0, 0);
        aq_run += COMP == 1 ? 32768 : 16384; wq_run += COMP == 1 ? 32768 : 16384; sq_run += COMP == 1 ? 2048 : 1024;
    };
    // PRE: chunks 0..2 of phase 2 go out from the last fp16 steps, each into the phase-1 stage that step has just retired (chunk c lands in
    // stage c: the launcher admits compensated products only with a step count that is a multiple of the ring depth, K % 128 == 0, and
    // K >= 256) -- the fp4 phase then starts with its ring already full.
    constexpr bool PRE = COMP != 0 && NSTAGE == V2_NST2;
#define KEEP_PIN() __builtin_amdgcn_sched_barrier(0)


    if (PERS && p.dbg && !first_tile) t_start = __builtin_readcyclecounter();      // diagnostics: the stamps describe the workgroup's LAST tile
    if (!PERS || first_tile) {
#pragma unroll
        for (int t = 0; t < NSTAGE - 1; ++t)
            if (t < steps) stage(t, t);
        if (steps >= NSTAGE - 1) wait_vmcnt<G * (NSTAGE - 2)>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (p.dbg) t_first = __builtin_readcyclecounter();
        if (NSTAGE - 1 < steps) stage(NSTAGE - 1, NSTAGE - 1);
    } else {
        // K steps 0, 1 and 2 of this tile were staged from the previous tile's last three steps and landed under its epilogue (waited for and
        // published at the end of it); step 3 goes out now, into the stage the epilogue has just left
        stage(3, 3);
        if (p.dbg) t_first = t_start;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    read_frags(0, 0, fw0, fa0);

    // steady state: every step but the last NSTAGE-1 has a full ring in flight
#define KEEP_STEADY_STEP(ISSUE)                                                                                                    \
    {                                                                                                                              \
        /* group 0 on R0[s]; R1[s] (same stage, already landed) is fetched under it */                                             \
        mfma_head(fw0, fa0);                                                                                                       \
        KEEP_PIN();                                                                                                                \
        read_frags(s % NSTAGE, 1, fw1, fa1);                                                                                       \
        KEEP_PIN();                                                                                                                \
        mfma_tail(fw0, fa0);                                                                                                       \
        KEEP_PIN();                                                                                                                \
        wait_vmcnt<G * (NSTAGE - 2)>();                                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                         \
        __builtin_amdgcn_s_barrier();                                                                                              \
        KEEP_PIN();                                                                                                                \
        /* group 1 on R1[s]; DMA for tile s+NSTAGE and R0[s+1] are issued under it */                                              \
        mfma_head(fw1, fa1);                                                                                                       \
        KEEP_PIN();                                                                                                                \
        ISSUE;                                                                                                                     \
        read_frags((s + 1) % NSTAGE, 0, fw0, fa0);                                                                                 \
        KEEP_PIN();                                                                                                                \
        mfma_tail(fw1, fa1);                                                                                                       \
    }
    int s = 0;
    if constexpr (PRE || PERS) {
        // (the launcher admits these only with steps >= 2 * NSTAGE)  The step that has nothing left to stage for this tile is peeled, so the
        // hot loop carries no "is there a K step left" branch: every scalar branch in it costs the wave tens of cycles in which it issues nothing
        for (; s < steps - NSTAGE; ++s) KEEP_STEADY_STEP(stage(s + NSTAGE, s % NSTAGE))
        if constexpr (PRE) KEEP_STEADY_STEP(stage2(0, 0))              // s == steps - NSTAGE: stage 0 is free -> chunk 0 of the fp4 phase
        else KEEP_STEADY_STEP(stage_next(0, 0))                        //                                      -> the next tile's K step 0
        ++s;
    } else {
        for (; s < steps - (NSTAGE - 1); ++s) KEEP_STEADY_STEP(if (s + NSTAGE < steps) stage(s + NSTAGE, s % NSTAGE))
    }
#undef KEEP_STEADY_STEP
    // drain: fewer tiles in flight
#define KEEP_DRAIN_STEP(WAIT, AFTER)                                                                                               \
    {                                                                                                                              \
        mfma_head(fw0, fa0);                                                                                                       \
        KEEP_PIN();                                                                                                                \
        read_frags(s % NSTAGE, 1, fw1, fa1);                                                                                       \
        KEEP_PIN();                                                                                                                \
        mfma_tail(fw0, fa0);                                                                                                       \
        KEEP_PIN();                                                                                                                \
        wait_vmcnt<WAIT>();                                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                         \
        __builtin_amdgcn_s_barrier();                                                                                              \
        KEEP_PIN();                                                                                                                \
        mfma_head(fw1, fa1);                                                                                                       \
        KEEP_PIN();                                                                                                                \
        AFTER;                                                                                                                     \
        read_frags((s + 1) % NSTAGE, 0, fw0, fa0);                                                                                 \
        KEEP_PIN();                                                                                                                \
        mfma_tail(fw1, fa1);                                                                                                       \
    }
    if constexpr (PRE) {
        // in-order retirement: step steps-3 needs tile steps-2, younger are tile steps-1 (G) and chunk 0; step steps-2 needs tile steps-1,
        // younger are chunks 0 and 1
        KEEP_DRAIN_STEP(G + V2_G2, stage2(1, 1))
        ++s;
        KEEP_DRAIN_STEP(2 * V2_G2, stage2(2, 2))
        ++s;
    } else if constexpr (PERS) {
        // step steps-3 needs tile steps-2: younger are tile steps-1 and the next tile's step 0; step steps-2 needs tile steps-1: younger are the
        // next tile's steps 0 and 1
        KEEP_DRAIN_STEP(2 * G, stage_next(1, 1))
        ++s;
        KEEP_DRAIN_STEP(2 * G, stage_next(2, 2))
        ++s;
    } else {
        for (; s < steps - 1; ++s) KEEP_DRAIN_STEP(0, (void)0)          // wait for everything that is left
    }
#undef KEEP_DRAIN_STEP
    mfma_head(fw0, fa0);
    KEEP_PIN();
    read_frags(s % NSTAGE, 1, fw1, fa1);
    KEEP_PIN();
    mfma_tail(fw0, fa0);
    mfma_group(fw1, fa1);
#undef KEEP_PIN

    // ---- phase 2: the two correction terms  W_lo A_hi^T + W_hi A_lo^T  on the MX-fp4 pipe (quant4.h) ------------------
    // Same accumulators, same wave tiling; K advances 64 per step (one v_mfma_scale_f32_32x32x64_f8f6f4 per 32x32 tile
    // and term: 32 cycles per SIMD against 4 x 32 for the fp16 pass over the same K).  Per step a workgroup streams
    // 34 KiB (fp16 pass: 64 KiB per K = 64) through a 4-stage LDS-DMA ring laid over the fp16 ring's stages: 3 chunks in flight, counted vmcnt,
    // one raw barrier per chunk.
    if constexpr (COMP != 0) {
        static_assert(BN == 256 && WM == 2 && WN == 4, "phase 2 is written for the 2 x 4 wave grid of the 256 x 256 tile");
        constexpr int G2 = V2_G2;
        if (p.dbg) t_first = __builtin_readcyclecounter();       // diagnostics: in a compensated launch stamp 1 marks the end of the fp16 phase
        // One term (COMP == 1): the same loop -- 12 fragment reads, 4 scale dwords, 16 MFMAs, 5 DMA instructions and 34 KiB per chunk -- with a chunk
        // covering K = 128 of the ONE product W_lo A_hi instead of K = 64 of two: half the chunks.  Per lane the two MFMAs of a tile then take the
        // 32-k blocks fhi and 2 + fhi of the chunk (the LDS image holds the four blocks of one plane back to back: 4 KiB each).
        constexpr int FH = COMP == 1 ? 4096 : 8192;              // LDS distance between the blocks of the two lane halves
        constexpr int SEC = COMP == 1 ? 8192 : 4096;             // ... between a lane's first and second fragment set (one term: blocks fhi -> 2 + fhi; two: plane hi -> lo)
        constexpr int SFH = COMP == 1 ? 256 : 512, SSEC = COMP == 1 ? 512 : 256;     // the same for the scale bytes
        const int NC = COMP == 1 ? p.K >> 7 : p.K >> 6;
        if constexpr (PRE) {
            // chunks 0..2 are in flight since the last fp16 steps: one barrier both retires the phase-1 ring and publishes chunk 0
            wait_vmcnt<G2 * (V2_NST2 - 2)>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // every wave is done with the phase-1 ring (its DMA queue is already drained)
#pragma unroll
            for (int t = 0; t < V2_NST2 - 1; ++t)
                if (t < NC) stage2(t, t);
            if (NC >= V2_NST2 - 1) wait_vmcnt<G2 * (V2_NST2 - 2)>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
        const int a_row = (wm * 128 + frow) * 16, w_row = (wn * 64 + frow) * 16;
        const int asc_off = fhi * SFH + (wm * 32 + frow) * 4, wsc_off = 1024 + fhi * SFH + ((wn >> 1) * 32 + frow) * 4;
        const int wsh = (wn & 1) * 16;
        // One step per chunk: fetch the 12 fragments + 4 scale dwords, issue the DMA of chunk c+3, 16 MFMAs, counted wait, barrier.
        // The phase is bound by the LDS-DMA stream, like phase 1 (34 KiB per chunk against 64 KiB per K = 64 there, and it takes
        // 0.53x the time): a two-group software pipeline of the MFMAs (fragments of one group fetched under the other's MFMAs,
        // barrier in between) measured 20 % SLOWER with or without sched_barrier pins, with the DMA issued early or late --
        // what counts is how long the DMA requests are in flight, and this order keeps three chunks outstanding the longest.
#define KEEP_V8(V_) v8i{(int)(V_).x, (int)(V_).y, (int)(V_).z, (int)(V_).w, 0, 0, 0, 0}
        // two terms: W_lo A_hi + W_hi A_lo (set "h" = plane hi, "l" = plane lo); one term: "h" / "l" are k 0..63 / 64..127 of the chunk, both of W_lo A_hi
#define KEEP_MX(I, J) \
            if constexpr (COMP == 1) { \
                acc[I][J] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(KEEP_V8(wh[I]), KEEP_V8(ah[J]), acc[I][J], 4, 4, I, swh, J, sah); \
                acc[I][J] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(KEEP_V8(wl[I]), KEEP_V8(al[J]), acc[I][J], 4, 4, I, swl, J, sal); \
            } else { \
                acc[I][J] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(KEEP_V8(wl[I]), KEEP_V8(ah[J]), acc[I][J], 4, 4, I, swl, J, sah); \
                acc[I][J] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(KEEP_V8(wh[I]), KEEP_V8(al[J]), acc[I][J], 4, 4, I, swh, J, sal); \
            }
        // One macro body, two loops (the last three chunks have nothing left to stage): the hot loop has no branch but its back-edge.
#define KEEP_CHUNK(ISSUE, WAIT)                                                                                                    \
        {                                                                                                                          \
            const unsigned char* sb = smem_raw + (c & (V2_NST2 - 1)) * V2_ST2;                                                     \
            const unsigned char* ss = smem_raw + V2_SC2_BASE + (c & (V2_NST2 - 1)) * V2_SC2;                                       \
            const unsigned char* sa = sb + fhi * FH;        /* this lane's K slice of the chunk: k 32*fhi .. 32*fhi+31 (+ 64 .. in the one-term form) */ \
            const unsigned char* sw = sb + 16384 + fhi * FH;                                                                       \
            /* native vector type, not HIP's uint4 struct: a struct load carries no alias info, and hipcc then puts an */          \
            /* s_waitcnt vmcnt(0) ("may read what an LDS-DMA in flight writes") in front of it: the ring drained on every chunk */  \
            u32x4 ah[4], al[4], wh[2], wl[2];                                                                                      \
            _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                                     \
                ah[jj] = *reinterpret_cast<const u32x4*>(sa + a_row + jj * 512);                                                   \
                al[jj] = *reinterpret_cast<const u32x4*>(sa + SEC + a_row + jj * 512);                                             \
            }                                                                                                                      \
            _Pragma("unroll") for (int ii = 0; ii < 2; ++ii) {                                                                     \
                wh[ii] = *reinterpret_cast<const u32x4*>(sw + w_row + ii * 512);                                                   \
                wl[ii] = *reinterpret_cast<const u32x4*>(sw + SEC + w_row + ii * 512);                                             \
            }                                                                                                                      \
            const int sah = *reinterpret_cast<const int*>(ss + asc_off), sal = *reinterpret_cast<const int*>(ss + asc_off + SSEC); \
            const int swh = (int)(*reinterpret_cast<const unsigned*>(ss + wsc_off) >> wsh),                                        \
                      swl = (int)(*reinterpret_cast<const unsigned*>(ss + wsc_off + SSEC) >> wsh);                                 \
            ISSUE;                                          /* into the stage chunk c-1 was read from */                           \
            KEEP_MX(0, 0) KEEP_MX(0, 1) KEEP_MX(0, 2) KEEP_MX(0, 3)                                                                \
            KEEP_MX(1, 0) KEEP_MX(1, 1) KEEP_MX(1, 2) KEEP_MX(1, 3)                                                                \
            wait_vmcnt<WAIT>();                             /* chunk c+1 has landed */                                             \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                     \
            __builtin_amdgcn_s_barrier();                                                                                          \
            __builtin_amdgcn_sched_barrier(0);              /* the peeled bodies must not be interleaved: 52 operand registers each */ \
        }
        int c = 0;                                          // NC >= 4: compensated products are admitted with K >= 256 only
        for (; c < NC - (V2_NST2 - 1); ++c) KEEP_CHUNK(stage2(c + V2_NST2 - 1, (c + V2_NST2 - 1) & (V2_NST2 - 1)), G2 * (V2_NST2 - 2))
        // the last three chunks: a second (rolled) loop -- three straight-line copies of the body made the allocator spill accumulators
#pragma clang loop unroll(disable)
        for (; c < NC; ++c) KEEP_CHUNK((void)0, 0)
#undef KEEP_CHUNK
#undef KEEP_MX
#undef KEEP_V8
    }

    if (p.dbg) t_loop = __builtin_readcyclecounter();

    if constexpr (EPI == EPI_TOP2) {
        // ---- prompt-screening epilogue: everything stays in the accumulator registers --------------------------------
        // A lane holds, per 32x32 tile (i, j) and register group rg, 4 consecutive columns n = .. + 8 rg + 4 fhi + e of ONE row
        // m = .. + j * 32 + frow: exactly the C = 4 class logits of one classifier (or two C = 2 classifiers).  Score them,
        // sum over this wave's 128 rows (4 tiles in-lane, then the 32 lanes of a half-wave), one store per classifier.
        const int C = p.top2_c;
        const int slot = (m0 >> 8) * 2 + wm;
        float* dst = p.top2_partial + (int64_t)slot * p.top2_kpad;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float sa = 0.f, sb = 0.f;
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const bool valid = m0 + wm * (TM * 32) + j * 32 + frow < p.M;
                    const float v0 = acc[i][j][rg * 4 + 0], v1 = acc[i][j][rg * 4 + 1], v2 = acc[i][j][rg * 4 + 2], v3 = acc[i][j][rg * 4 + 3];
                    const float a = fmaxf(v0, v1), b = fminf(v0, v1), c = fmaxf(v2, v3), d = fminf(v2, v3);
                    if (C == 4) {
                        const float t1 = fmaxf(a, c), t2 = fmaxf(fminf(a, c), fmaxf(b, d));
                        sa += valid ? (t1 - t2) - fabsf(t1 + t2 - 1.0f) : 0.f;
                    } else {
                        sa += valid ? (a - b) - fabsf(a + b - 1.0f) : 0.f;
                        sb += valid ? (c - d) - fabsf(c + d - 1.0f) : 0.f;
                    }
                }
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
                if (frow == 0) {
                    const int n = n0 + wn * (TN * 32) + i * 32 + 8 * rg + 4 * fhi;
                    if (C == 4) dst[n >> 2] = sa;
                    else { dst[n >> 1] = sa; dst[(n >> 1) + 1] = sb; }
                }
            }
        return;
    }
    // ---- epilogue through LDS ---------------------------------------------------------------------
    // An MFMA C fragment gives a lane 4 consecutive n of ONE m, so storing straight from registers makes
    // every store instruction touch 32 different rows (measured 16k-33k cycles per tile, bound by the
    // address path, not by bytes).  Each wave instead bounces one 32(m) x WN_COLS(n) slab at a time through
    // its private 16 KiB of the (now idle) LDS ring and leaves with whole 128/256-byte row segments.
    __syncthreads();
    constexpr int WN_COLS = TN * 32;                 // 64 for the 256x256 variant
    constexpr int PITCH = WN_COLS + 4;               // fp32 elements; +4 keeps ds_write_b128 conflict free
    constexpr int PITCH16 = WN_COLS + 8;             // fp16 elements: 144-byte rows keep ds_read_b128 aligned
    constexpr int SLAB_FLOATS = ((32 * PITCH > 32 * PITCH16 ? 32 * PITCH : 32 * PITCH16) + 63) / 64 * 64 + 96;   // fp32 slab or fp16 hi+lo slabs (2400 for 64 columns)
    static_assert(SLAB_FLOATS * 4 * WM * WN <= NSTAGE * (V2_BM + BN) * V2_BK * 2, "epilogue slabs must fit the LDS ring");
    // persistent: stages 0..2 hold the next tile's first K steps, the slabs live in stage 3 + the 32 KiB of LDS behind the ring: 8 KiB per wave.
    // That is exactly 32 x 64 fp32 without padding -- the 16-byte chunks of row r are rotated by r instead (same bank pattern as the padded
    // rows) -- and one fp16 plane with padding: the persistent kernel is launched for hi-only outputs.
    constexpr int SLAB_BASE = PERS ? 3 * (V2_BM + BN) * V2_BK * 2 : 0;
    constexpr int SLAB_STRIDE = PERS ? ((EPI == EPI_F16 || EPI == EPI_GELU_F16) ? 1152 : 2048) : SLAB_FLOATS;     // floats per wave
    static_assert(!PERS || (SLAB_BASE + SLAB_STRIDE * 4 * WM * WN <= V2_PERS_LDS && 32 * PITCH16 * 2 <= SLAB_STRIDE * 4), "persistent epilogue slabs must fit behind the three prefetched stages");
    float* slab = reinterpret_cast<float*>(smem_raw + SLAB_BASE) + wave * SLAB_STRIDE;
    auto slab_off = [&](int r, int col) { return PERS ? r * 64 + ((((col >> 2) + r) & 15) << 2) : r * PITCH + col; };   // col: multiple of 4
    // (a one-term launch consumes and produces Q(X_hi) planes only: no lo slab, no lo quantisation, unless a caller asks for the fp16 lo plane itself)
    const bool want_lo = !PERS && (p.out_lo || (p.out_q && COMP != 1));
    constexpr bool F16_OUT = (EPI == EPI_F16 || EPI == EPI_GELU_F16);
    constexpr int CPL = F16_OUT ? 8 : 4;             // columns per lane on the way out
    constexpr int LPR = WN_COLS / CPL;               // lanes per row
    constexpr int RPI = 64 / LPR;                    // rows per instruction
    const int ocol = (lane % LPR) * CPL;
    const int orow_in = lane / LPR;
    const int ncol = n0 + wn * WN_COLS + ocol;
    f32x4 bias4[CPL / 4], ls4[CPL / 4];
#pragma unroll
    for (int c = 0; c < CPL / 4; ++c) {
        bias4[c] = (EPI == EPI_PARTIAL) ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(p.bias + ncol + c * 4);
        if (EPI == EPI_RESID_LS) ls4[c] = *reinterpret_cast<const f32x4*>(p.ls + ncol + c * 4);
    }
    // fp16 outputs: bias (+GELU) and the fp16 conversion happen on the accumulator fragments, so only half
    // the bytes cross the LDS (its ds_write rate, ~80 B/clk/CU, was a third of this epilogue)
    f16* slab_hi = reinterpret_cast<f16*>(slab);
    f16* slab_lo = slab_hi + 32 * PITCH16;
    f32x4 bfrag[F16_OUT ? TN * 4 : 1];
    if (F16_OUT) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                bfrag[i * 4 + rg] = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * WN_COLS + i * 32 + 8 * rg + 4 * fhi);
    }
    f32x4 res2[2][32 / RPI];
    int64_t oo2[2][32 / RPI];
    auto load_res = [&](int jj, f32x4 (&res)[32 / RPI], int64_t (&oo)[32 / RPI]) {
        const int mb = m0 + wm * (TM * 32) + jj * 32;
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int m = mb + it * RPI + orow_in;
            const int mc = m < p.M ? m : p.M - 1;
            int prow; int64_t orow;
            gemm_epilogue_row<EPI>(p, mc, prow, orow);
            oo[it] = orow * p.N + ncol;
            if (EPI == EPI_PARTIAL) { oo[it] = ((int64_t)zsplit * p.M + mc) * p.N + ncol; res[it] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            else if (EPI == EPI_PATCH) res[it] = *reinterpret_cast<const f32x4*>(p.pos + (int64_t)prow * p.N + ncol);
            // the residual stream is read once and written once per GEMM (206 MB each way at 256 tiles): non-temporal both ways keeps it from
            // evicting the operand panels the K loops live on (+0.65 % end to end; the load or the store alone: +0.3 % / +0.2 %)
            else res[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.resid + oo[it]));
        }
    };
    if (!F16_OUT) load_res(0, res2[0], oo2[0]);
    // fp16 outputs go to the blk layout (the consumer is a GEMM) or row-major (attention): ONE address form, offset(m) = (m / 256) * out_sa +
    // (m % 256) * out_sb + out_g, with the three constants picked here -- a "which layout" test per store was two scalar branches in front of each
    // of a tile's 16 stores
    const int out_ld = p.out_ld > 0 ? p.out_ld : p.N;
    const int64_t out_sa = p.out_kt > 0 ? (int64_t)p.out_kt * 8192 : (int64_t)256 * out_ld;
    const int out_sb = p.out_kt > 0 ? 32 : out_ld;
    const int out_g = p.out_kt > 0 ? ((ncol >> 5) * 8192 + (ncol & 31)) : ncol + p.out_col0;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int mbase = m0 + wm * (TM * 32) + j * 32;
        if (F16_OUT) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    f32x2 a = {acc[i][j][rg * 4 + 0] + bfrag[i * 4 + rg][0], acc[i][j][rg * 4 + 1] + bfrag[i * 4 + rg][1]};
                    f32x2 b = {acc[i][j][rg * 4 + 2] + bfrag[i * 4 + rg][2], acc[i][j][rg * 4 + 3] + bfrag[i * 4 + rg][3]};
                    if (EPI == EPI_GELU_F16) {
                        // strict / compensated (lo or fp4 planes wanted): full-accuracy polynomial; a persistent launch is hi-only by construction
                        if (COMP != 0 || (!PERS && (p.out_lo || p.out_q))) { a = gelu_fast2(a); b = gelu_fast2(b); }     // (a compensated launch always is)
                        else { a = gelu_fast2_fp16(a); b = gelu_fast2_fp16(b); }
                    }
                    f16x4 h, l;
                    f16 hh, ll;
                    split_f16(a[0], hh, ll); h[0] = hh; l[0] = ll;
                    split_f16(a[1], hh, ll); h[1] = hh; l[1] = ll;
                    split_f16(b[0], hh, ll); h[2] = hh; l[2] = ll;
                    split_f16(b[1], hh, ll); h[3] = hh; l[3] = ll;
                    const int so = frow * PITCH16 + i * 32 + 8 * rg + 4 * fhi;
                    *reinterpret_cast<f16x4*>(slab_hi + so) = h;
                    if (want_lo) *reinterpret_cast<f16x4*>(slab_lo + so) = l;
                }
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int r = it * RPI + orow_in;
                const int m = mbase + r;
                const f16x8 h = *reinterpret_cast<const f16x8*>(slab_hi + r * PITCH16 + ocol);
                if (m < p.M) {
                    const int64_t o = (int64_t)(m >> 8) * out_sa + (m & 255) * out_sb + out_g;
                    __builtin_nontemporal_store(h, reinterpret_cast<f16x8*>(p.out_hi + o));      // written once, read by the next kernel: +0.75 % non-temporal
                    if (want_lo) {
                        const f16x8 l = *reinterpret_cast<const f16x8*>(slab_lo + r * PITCH16 + ocol);
                        if (p.out_lo) *reinterpret_cast<f16x8*>(p.out_lo + o) = l;
                        // this GEMM's N is the consumer's K: 8 lanes cover the wave's 64 columns = two MX blocks of 4 lanes each
                        if (p.out_q) q4_store8(p.out_q, p.out_sc, p.out_kt, m, ncol >> 5, (ocol >> 3) & 3, h, l);
                    } else if (COMP == 1 && !PERS && p.out_q) {
                        q4_store8_hi(p.out_q, p.out_sc, p.out_kt, m, ncol >> 5, (ocol >> 3) & 3, h);
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rg * 4 + e];
                    *reinterpret_cast<f32x4*>(slab + slab_off(frow, i * 32 + 8 * rg + 4 * fhi)) = v;
                }
            // the residual rows of pass j+1 are requested before pass j is finished, so the read latency of
            // the read-modify-write hides behind the previous pass (res2 / oo2 are double-buffered by parity of j)
            if (j + 1 < TM) load_res(j + 1, res2[(j + 1) & 1], oo2[(j + 1) & 1]);
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int r = it * RPI + orow_in;
                f32x4 x = *reinterpret_cast<const f32x4*>(slab + slab_off(r, ocol));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[e] += bias4[0][e];
                    // LayerScale product rounded on its own, then added (torch's `x + gamma * y`, and what the atomic form computes): no fused multiply-add here
                    x[e] = (EPI == EPI_RESID_LS) ? res2[j & 1][it][e] + __fmul_rn(ls4[0][e], x[e]) : res2[j & 1][it][e] + x[e];
                }
                if (mbase + r < p.M) {
                    float* dst = (EPI == EPI_PARTIAL) ? p.splitk_ws : (EPI == EPI_RESID_F32) ? p.out_f32 : p.resid;
                    __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(dst + oo2[j & 1][it]));
                }
            }
        }
    }
    if constexpr (!PERS) break;
    else {
        if (!has_next) { wait_vmcnt<0>(); break; }          // the last tile's (unused) prefetch must not outlive the workgroup's LDS
        // The next tile's K steps 0..2 were issued a whole epilogue ago; vmcnt(0) here waits, at most, for the acknowledgement of this wave's last
        // stores.  The barrier publishes the staged data and says every wave is done with its slab (stage 3 may be overwritten).
        wait_vmcnt<0>();
        __syncthreads();
        m0 = m0_n; n0 = n0_n; a_tile = a_tile_n; w_tile = w_tile_n;
        st_a = p.a_hi + a_tile + 3 * 8192; st_w = p.w_hi + w_tile + 3 * 8192; st_left = ktiles - 3;    // K steps 0..2 of the new tile are in the ring
        tile_cur += (int)gridDim.x;
        first_tile = false;
    }
  }
    if (p.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) {
            long long* d = p.dbg + (size_t)blockIdx.x * 4;
            d[0] = t_start; d[1] = t_first; d[2] = t_loop; d[3] = __builtin_readcyclecounter();
        }
    }
}

template <int EPI>
int launch_v2_pers(const GemmParams& p, hipStream_t s) {
    auto kernel = &gemm_f16_v2_kernel<256, 2, 4, 4, EPI, 0, true>;
    constexpr size_t lds_bytes = (EPI == EPI_F16 || EPI == EPI_GELU_F16) ? V2_PERS_LDS_F16 : V2_PERS_LDS;
    if (!keep_lds_opt_in(reinterpret_cast<const void*>(kernel), lds_bytes)) return -2;
    const int tiles = (p.N / 256) * ((p.M + V2_BM - 1) / V2_BM);
    int cus = keep_num_cus();
    if (p.tune && p.tune->gemm_persistent > 1 && p.tune->gemm_persistent < cus) cus = p.tune->gemm_persistent;     // experiment: fewer workgroups than CUs
    hipLaunchKernelGGL(kernel, dim3(tiles < cus ? tiles : cus), dim3(512), lds_bytes, s, p);
    return 0;
}

template <int BN, int WM, int WN, int NSTAGE, int EPI, int COMP>
int launch_v2_one(const GemmParams& p, hipStream_t s) {
    constexpr size_t ring = (size_t)NSTAGE * (V2_BM + BN) * V2_BK * sizeof(f16);
    constexpr size_t ring2 = (size_t)V2_NST2 * (V2_ST2 + V2_SC2);
    constexpr size_t lds_bytes = COMP != 0 && ring2 > ring ? ring2 : ring;
    auto kernel = &gemm_f16_v2_kernel<BN, WM, WN, NSTAGE, EPI, COMP>;
    if (!keep_lds_opt_in(reinterpret_cast<const void*>(kernel), lds_bytes)) return -2;
    const int grid = (p.N / BN) * ((p.M + V2_BM - 1) / V2_BM) * (EPI == EPI_PARTIAL ? p.ksplit : 1);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(WM * WN * 64), lds_bytes, s, p);
    return 0;
}

template <int BN, int WM, int WN, int NSTAGE>
int launch_v2(const GemmParams& p, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_F16:      return launch_v2_one<BN, WM, WN, NSTAGE, EPI_F16, 0>(p, s);
        case EPI_GELU_F16: return launch_v2_one<BN, WM, WN, NSTAGE, EPI_GELU_F16, 0>(p, s);
        case EPI_RESID_LS: return launch_v2_one<BN, WM, WN, NSTAGE, EPI_RESID_LS, 0>(p, s);
        case EPI_PATCH:    return launch_v2_one<BN, WM, WN, NSTAGE, EPI_PATCH, 0>(p, s);
        case EPI_PARTIAL:  return launch_v2_one<BN, WM, WN, NSTAGE, EPI_PARTIAL, 0>(p, s);
        default:           return launch_v2_one<BN, WM, WN, NSTAGE, EPI_RESID_F32, 0>(p, s);
    }
}

}  // namespace keepk

// returns 0 if launched, 1 if the shape is not covered by this variant
//   variant 256  : 256x256 tiles, 8 waves, one workgroup per CU
//   variant 128  : 256x128 tiles, 8 waves (64x64 per wave)
// p.comp: the 256x256 kernel with the MX-fp4 correction phase (GELU and residual epilogues: the MLP of the image tower)
int launch_gemm_f16_v2(const GemmParams& p, int epi, int variant, hipStream_t s) {
    using namespace keepk;
    if (p.K % V2_BK) return 1;
    if (p.comp) {
        if (p.N % 256 || p.K % 128 || p.K < 256 || p.nseg != 1 || !p.a_q || !p.a_sc || !p.w_q || !p.w_sc) return 1;
        if (p.comp == 1) {                               // the W_lo term only: chunks of K = 128, at least four of them (image-tower MLP shapes)
            if (p.K < 512) return 1;
            if (epi == EPI_GELU_F16) return launch_v2_one<256, 2, 4, 4, EPI_GELU_F16, 1>(p, s);
            if (epi == EPI_RESID_LS) return launch_v2_one<256, 2, 4, 4, EPI_RESID_LS, 1>(p, s);
            if (epi == EPI_F16) return launch_v2_one<256, 2, 4, 4, EPI_F16, 1>(p, s);
            return 1;
        }
        if (epi == EPI_GELU_F16) return launch_v2_one<256, 2, 4, 4, EPI_GELU_F16, 2>(p, s);
        if (epi == EPI_RESID_LS) return launch_v2_one<256, 2, 4, 4, EPI_RESID_LS, 2>(p, s);
        if (epi == EPI_F16) return launch_v2_one<256, 2, 4, 4, EPI_F16, 2>(p, s);
        if (epi == EPI_TOP2) return launch_v2_one<256, 2, 4, 4, EPI_TOP2, 2>(p, s);
        return 1;
    }
    if (epi == EPI_TOP2) return p.N % 256 ? 1 : launch_v2_one<256, 2, 4, 4, EPI_TOP2, 0>(p, s);
    if (p.impl_hint == 2128 && p.nseg == 1 && !p.out_lo && !p.out_q && p.N % 128 == 0 && (p.N / 128) * ((p.M + V2_BM - 1) / V2_BM) > 2 * keep_num_cus()) {
        // (taken only when the launch goes through: a device that refuses the dynamic-LDS opt-in falls through to the selection below, as launch_v2_pers does)
        if (epi == EPI_RESID_LS && launch_v2_one<128, 2, 2, 3, EPI_RESID_LS, 0>(p, s) == 0) return 0;
        if (epi == EPI_F16 && launch_v2_one<128, 2, 2, 3, EPI_F16, 0>(p, s) == 0) return 0;
        if (epi == EPI_GELU_F16 && launch_v2_one<128, 2, 2, 3, EPI_GELU_F16, 0>(p, s) == 0) return 0;
    }
    if (variant == 256 && p.N % 256 == 0 && p.tune && p.tune->gemm_persistent && p.nseg == 1 && !p.out_lo && !p.out_q && p.K % 128 == 0 && p.K >= 256 &&
        (p.N / 256) * ((p.M + V2_BM - 1) / V2_BM) > keep_num_cus()) {
        // (a device that does not grant all 160 KiB of LDS to one workgroup falls through to the one-tile-per-workgroup launch)
        if (epi == EPI_F16 && launch_v2_pers<EPI_F16>(p, s) == 0) return 0;
        if (epi == EPI_GELU_F16 && launch_v2_pers<EPI_GELU_F16>(p, s) == 0) return 0;
        if (epi == EPI_RESID_LS && launch_v2_pers<EPI_RESID_LS>(p, s) == 0) return 0;
    }
    if (variant == 256 && p.N % 256 == 0) return launch_v2<256, 2, 4, 4>(p, epi, s);
    if (variant == 128 && p.N % 128 == 0) return launch_v2<128, 4, 2, 4>(p, epi, s);
    return 1;
}
