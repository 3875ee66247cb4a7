// fp16-operand / fp32-accumulate MFMA GEMM with fused epilogues (gfx950): kernel selection.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]          A: activations [M][K], W: torch Linear weight [N][K]
//
// This is the arithmetic behind every nn.Linear / the patch-embed conv of the two
// towers (timm Block: qkv, proj, fc1, fc2 -- SURVEY.md A.1; HF BertLayer: q/k/v,
// attention.output.dense, intermediate.dense, output.dense -- A.2).  The kernels live in
// gemm_f16_v2.hip (256 x {256,128} LDS-DMA tiles, persistent walk, MX-fp4 correction phase, K-sliced launch) and
// gemm_f16_skinny.hip (calls of a few hundred rows); this file picks one per call.  The measured-negative variants of
// rounds 1-2 (register-staged 128x128, static persistent tiles, W-direct, 4-wave tiles) are kept under tools/experiments/.
#include "gemm_epilogue.h"

int launch_gemm_f16_v2(const GemmParams& p, int epi, int variant, hipStream_t s);
int launch_gemm_f16(const GemmParams& p_in, int epi, hipStream_t s) {
    // Operands in blk layout -> the LDS-DMA kernels (gemm_f16_v2.hip); tune->gemm_impl only picks the tile width.
    static const KeepTune defaults;
    GemmParams p = p_in;
    const KeepTune& t = p.tune ? *p.tune : defaults;
#ifdef KEEP_DIAGNOSTICS
    p.ablate = t.gemm_ablate;
    p.dbg = t.dbg;
#else
    p.ablate = 0;
    p.dbg = nullptr;
#endif
    int impl = t.gemm_impl;
    if (p.out_ld > 0 && (epi != EPI_F16 || p.out_kt > 0 || p.N % 256)) return -1;      // a strided row-major output is an EPI_F16 feature of the 256x256 kernel
    if (p.out_ld > 0) return launch_gemm_f16_v2(p, epi, 256, s) == 0 ? 0 : -1;
    if (epi == EPI_TOP2) return launch_gemm_f16_v2(p, epi, 256, s) == 0 ? 0 : -1;      // fused prompt screening: 256x256 kernel only
    if (p.comp) {                                    // compensated product: always the 256x256 kernel (callers route small M through nseg = 3)
        return launch_gemm_f16_v2(p, epi, 256, s) == 0 ? 0 : -1;
    }
    if (impl == 0 && p.M <= t.gemm_skinny_m && p.M <= SKINNY_MAX_M && p.splitk_ws) {
        const int rc = launch_gemm_f16_skinny(p, epi, p.splitk_ws, p.splitk_bytes, s);
        if (rc >= 0) return rc;
    }
    if (impl == 0 && t.gemm_splitk_tiles > 0 && p.splitk_ws && p.N % 256 == 0 && p.K % 32 == 0) {
        // Between the small-M kernel and a full machine: ceil(M/256) * N/256 tiles on 256 CUs, each walking all of K alone.
        // Cut K into S slices (>= 8 steps each), fp32 partials, then the shared reduce + epilogue kernel.
        const int tiles = ((p.M + 255) / 256) * (p.N / 256), KT = p.K / 32;
        int S = tiles < t.gemm_splitk_tiles ? 256 / tiles : 1;
        if (S > 8) S = 8;
        if (S > KT / 8) S = KT / 8;
        if (S >= 2 && (size_t)S * p.M * p.N * sizeof(float) <= p.splitk_bytes) {
            GemmParams q = p;
            q.ksplit = S;
            if (launch_gemm_f16_v2(q, EPI_PARTIAL, 256, s) == 0) return launch_gemm_splitk_reduce(p, epi, p.splitk_ws, S, s);
        }
    }
    if ((impl != 128 && impl != 256) || (impl == 256 && p.N % 256)) impl = (p.N % 256 == 0) ? 256 : 128;
    if (p.resid_copy) return launch_gemm_f16_v2(p, epi, impl, s) == 0 ? GEMM_NO_RESID_COPY : -1;      // (only the split-K reduce kernels store the second copy)
    return launch_gemm_f16_v2(p, epi, impl, s) == 0 ? 0 : -1;      // (-1: shape not covered or the LDS opt-in was refused -- the callers report it)
}

