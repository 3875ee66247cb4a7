// Shared device/host helpers for libkeep_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define KEEP_WAVE 64

// fp16 has 65504 as its largest finite value: saturate instead of producing inf
// (ViT residual streams can carry outlier channels; the stream itself stays fp32).
// hi/lo split used by the "strict" (3-pass) precision mode: x ~= hi + lo with
// hi = fp16(x), lo = fp16(x - hi).  |x - hi - lo| <= 2^-22 |x| (2^-25 abs floor).
// No saturation: a value beyond the fp16 range (65504) becomes inf on purpose -- it turns into NaN at the next LayerNorm / softmax and the
// final normalisation kernel raises the handle's "non-finite features" flag (the reference computes in fp32 and would not overflow;
// clamping would return plausible garbage instead).
__device__ __forceinline__ void split_f16(float x, f16& hi, f16& lo) {
    // x is pinned as ONE materialised fp32 value.  Without this hipcc (-ffp-contract=fast) folds the multiply / fma that produced x into
    // v_fma_mixlo_f16 for the hi that feeds `lo` (exact product rounded once to fp16) while the STORED hi is v_cvt(v_mul) (rounded twice):
    // the two disagree by an fp16 ulp on a few values per thousand, and hi + lo is then 1e-3 off instead of 2^-22 (found by the split-attention
    // operator test when the saturating clamp, which happened to block the pattern, was removed).
    asm("" : "+v"(x));
    hi = (f16)x;
    lo = (f16)(x - (float)hi);
}

// Exact (erf) GELU: timm nn.GELU and HF hidden_act="gelu".
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---------------------------------------------------------------------------
// K-blocked fp16 operand layout ("blk").  Every fp16 matrix that a GEMM reads as an operand (LN output,
// attention output, MLP hidden, im2col patches, all weights) is stored as
//     X[m][k]  ->  ((m / 256) * KT + k / 32) * 8192 + (m % 256) * 32 + (k % 32),     KT = K / 32
// i.e. tile-major [row tile of 256][K slice of 32][256 rows][32 k]: the 16 KiB that one workgroup needs
// for one K step of one operand is ONE contiguous block.  Measured with tools/ubench/glds_tile_bw.hip:
// the LDS-DMA stream of the 256x256 GEMM runs at 21.5 TB/s on this layout against 13.7 TB/s on
// row-major [M][K] (64-byte pieces at a 2 KiB pitch), which was what bounded the K loop.
// Rows are padded to a multiple of 256 (padding rows are never stored by a consumer).
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ int64_t blk_off(int m, int k, int KT) {
    return ((int64_t)(m >> 8) * KT + (k >> 5)) * 8192 + ((m & 255) << 5) + (k & 31);
}
static inline size_t blk_elems(int64_t M, int64_t K) { return (size_t)((M + 255) / 256 * 256) * (size_t)K; }

// ---------------------------------------------------------------------------
// Kernel launch parameter blocks (plain structs, passed by value)
// ---------------------------------------------------------------------------

// Kernel-selection knobs.  They belong to a handle (keep_set_option) and travel inside the launch parameter blocks:
// there is no process-wide kernel state, so two handles (two GPUs, two threads) never see each other's settings.
struct KeepTune {
    int gemm_impl = 0;           // 0 auto | 128 / 256: LDS-DMA tile width 
    int gemm_skinny_m = 320;     // calls with M <= this take the register-direct split-K kernel (0: never)
    int skinny_wide = 1;         // 1: small-M calls with M >= 64 on the 128x128 / 64x64-per-wave kernel (two MFMAs per fragment loaded); 0: 32x128 tiles always (A/B)
    int gemm_splitk_tiles = 64;  // a 256x256 GEMM with fewer tiles than this is cut into K slices (0: never)
    int sgemv_m = 16;            // rows up to which the few-row fp32 kernel is used (0: never)
    int ln_impl = 2;             // 2 (default): LDS-transposed blk stores from 4-wave workgroups (one 56-register wave per SIMD: fits beside two waves of the other lane's
                                 //    persistent fc1 GEMM; +0.32 % end to end in a six-round rotated A/B, profiles/r05_ab_layernorm_workgroups.txt); 1: the same from 8-wave
                                 //    workgroups (512-byte runs: 2 % faster alone); 0: per-row stores
    int attn_waves = 16;         // 16: image-tower calls with >= 512 (image, head) pairs take the persistent double-buffered kernel (13 compute + 3 loader waves),
                                 //     everything else 8 waves per workgroup; 8 / 4: one (image, head) pair per workgroup of that many waves
    int gemm_persistent = 1;     // 1: plain 256x256 GEMMs with more tiles than CUs run as one workgroup per CU walking the tile list, the next tile's
                                 //    first three K steps prefetched under the epilogue (0: one tile per workgroup; n > 1: n workgroups, experiments)
    int gemm_ablate = 0;         // diagnostics (KEEP_DIAGNOSTICS builds only)
    long long* dbg = nullptr;    // diagnostics: per-workgroup shader-clock stamps
};

// Epilogue selector of the fp16 MFMA GEMM  C[m][n] = sum_k A[m][k] W[n][k].
enum GemmEpi : int {
    EPI_F16      = 0,   // out_f16 = acc + bias                                   (QKV)
    EPI_GELU_F16 = 1,   // out_f16 = gelu(acc + bias)                             (fc1)
    EPI_RESID_LS = 2,   // resid  += ls[n] * (acc + bias)   fp32, in place        (proj / fc2, ViT)
    EPI_PATCH    = 3,   // resid[b*197+1+p] = acc + bias + pos[1+p]               (patch embed)
    EPI_RESID_F32= 4,   // out_f32 = acc + bias + resid      (BERT pre-LN sum; may alias resid)
    EPI_PARTIAL  = 5,   // internal: fp32 partial sums of one K slice -> splitk_ws[slice][M][N] (no bias); the
                        // split-K reduce kernel of gemm_f16_skinny.hip then applies the real epilogue
    EPI_TOP2     = 6,   // prompt screening (WSI_evaluation/utils.py:107-130): columns are K classifiers x C classes (C = 2 or 4,
                        // consecutive); per (row, classifier) top-2 margin score (v1 - v2) - |v1 + v2 - 1| taken in the
                        // accumulator registers and summed over the tile's rows -> top2_partial[row slot][classifier].
                        // No logit is ever written.
};

struct GemmParams {
    const f16* a_hi; const f16* a_lo;     // activations [M][K] in blk layout (row-major for the v1 kernel); lo only when nseg==3
    const f16* w_hi; const f16* w_lo;     // weights [N][K], same layout as A
    int M, N, K;                          // K = per-segment depth; multiples: N%128==0, K%64==0
    int nseg;                             // 1: A_hi*W_hi ; 3: A_hi*W_hi + A_lo*W_hi + A_hi*W_lo
    // comp != 0: after the fp16 pass the correction terms run on the MX-fp4 pipe (quant4.h): needs the fp4 side planes
    // of both operands, K % 128 == 0, nseg == 1, and the 256x256 kernel (the small-M paths use nseg = 3 instead).
    //   2 (or any value but 1): both first-order terms  W_lo A_hi + W_hi A_lo
    //   1: the weight-rounding term W_lo A_hi only (reads plane 0 of a_q and plane 1 of w_q; K >= 512); with out_q set, writes plane 0 only
    int comp;
    const unsigned char* a_q; const unsigned char* a_sc;
    const unsigned char* w_q; const unsigned char* w_sc;
    unsigned char* out_q; unsigned char* out_sc;   // EPI_GELU_F16 with out_kt > 0: also emit the fp4 planes of the output (K = N of this GEMM)
    const float* bias;                    // [N]
    const float* ls;                      // [N]   (EPI_RESID_LS)
    const float* pos;                     // [197][N] (EPI_PATCH)
    float* resid;                         // fp32 [M'][N]
    float* resid_copy; int64_t resid_copy_ld;   // nullable (small-M reduce, EPI_RESID_LS only): row m of the updated residual is ALSO stored at resid_copy + m * resid_copy_ld
                                          // (the CLS-row chain writes its rows back into the token stream from its last reduce: no scatter launch)
    float* out_f32;                       // EPI_RESID_F32
    f16* out_hi; f16* out_lo;             // fp16 outputs (lo optional)
    int out_kt;                           // > 0: fp16 output in blk layout with KT = out_kt (= N/32); 0: row-major [M][N]
    int out_ld, out_col0;                 // row-major fp16 output (EPI_F16, 256x256 kernel only): leading dimension (0 = N) and first column of this GEMM's N columns
                                          // inside a wider buffer (the K | V part of the last block's qkv)
    int patches_per_img;                  // EPI_PATCH: 196
    // optional LayerNorm of the updated row, fused into the split-K reduce (EPI_RESID_LS / EPI_RESID_F32, N <= 1024);
    // launch_gemm_f16 reports through its return value whether it was applied (bit 0) -- the big kernel never does
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    f16* ln_out_hi; f16* ln_out_lo; float* ln_out_f32;   // fp16 in blk layout (KT = N / 32); fp32 row-major [M][N], may alias resid / out_f32
    int top2_c; float* top2_partial; int top2_kpad;   // EPI_TOP2: classes per classifier, [2 * ceil(M/256)][top2_kpad] partial sums
    int impl_hint;                        // per-call kernel choice of the caller (engine option "proj_impl"): 2128 = 256x128 tiles, 4 waves, 3-stage ring, TWO
                                          // workgroups per CU (72 KiB of LDS each) -- one workgroup's epilogue runs under the other's K loop; 0 = the default choice
    int ksplit;                           // internal (EPI_PARTIAL): number of K slices the grid is replicated over
    float* splitk_ws; size_t splitk_bytes; // scratch for the small-M split-K kernel (gemm_f16_skinny.hip); null: never used
    long long* dbg;                       // diagnostics: per-workgroup [start, first tile landed, loop end, end] shader clocks
    int ablate;                           // diagnostics only: 1 = skip staging DMA, 2 = skip MFMA loop (results wrong)
    const KeepTune* tune;                 // kernel selection of the calling handle (null: defaults)
};

int launch_gemm_f16(const GemmParams& p, int epi, hipStream_t s);             // blk-layout operands (product path); returns GEMM_DID_LN or 0
constexpr int GEMM_NO_RESID_COPY = 2;   // launch_gemm_f16: GemmParams.resid_copy was NOT honoured (the call took a kernel without that epilogue: the caller scatters the rows)
constexpr int GEMM_DID_LN = 1;
// split-K scratch: 30 MiB cover every small-M shape (M <= SKINNY_MAX_M: <= 768 + 1024 partial tiles of 32 x 128 fp32);
// the mid-size path (256x256 tiles x K slices when a GEMM has fewer tiles than CUs) needs <= 448 tiles of 256 KiB
constexpr int SKINNY_MAX_M = 1024;
constexpr size_t SKINNY_WS_BYTES = (size_t)448 * 256 * 256 * 4;
int launch_gemm_splitk_reduce(const GemmParams& p, int epi, const float* ws, int S, hipStream_t s);   // returns 0 or GEMM_DID_LN
int launch_gemm_f16_skinny(const GemmParams& p, int epi, float* ws, size_t ws_bytes, hipStream_t s);

// Attention over a fused [M][3*D] qkv buffer (token-major; q|k|v, head-major inside each).
struct AttnParams {
    const f16* qkv_hi; const f16* qkv_lo; // lo only in split mode
    f16* out_hi; f16* out_lo;             // [M][D] row-major, or blk layout when out_kt > 0
    int out_kt;
    const int64_t* mask;                  // [batch][ntok] (1 = attend) or nullptr
    int batch, ntok, heads;               // head_dim fixed at 64
    int q_rows;                           // > 0: only the first q_rows query rows of every image are computed (CLS-only last block)
    const f16* q_hi; int q_ld;            // q_rows == 1 only, nullable: the ONE query row of image b at q_hi + b * q_ld (head-major, like the q part of a qkv row) instead of
                                          // qkv_hi -- the last block computes Q for its CLS rows only
    int split;                            // 0/1
    // nullable (single-pass mode only): the attention output of query row 0 (the CLS row) of image b, from the fp32 accumulators, ALSO as hi + lo
    // planes into row b of a compact [batch][D] operand in the layout of `out_*` (KEEP_ATTN_PROJ_CLS: the CLS rows' proj runs again as a split product)
    f16* cls_hi; f16* cls_lo;
    // split mode with 256 < ntok <= 512: the K / V planes (hi + lo) of more than 256 keys do not fit the LDS, so the keys are processed in two
    // windows of <= 256 (two launches) and merged like an online softmax.  part_ws: caller's scratch, batch * heads * ntok * ATT_PART_FLOATS floats
    // (unnormalised output row + running maximum + sum per query); launch_attention fills key0 / kcount / part_out / part_in itself.
    float* part_ws; size_t part_bytes;
    int key0, kcount;                     // internal: key window [key0, key0 + kcount) of this launch (kcount 0 = all keys)
    float* part_out; const float* part_in;   // internal: first window writes its partial state, second window merges it and stores the result
    float scale;                          // 1/sqrt(64)
    long long* dbg;                       // diagnostics: per-workgroup [start, staged, end] shader clocks (tools/attn_timeline.py)
    const KeepTune* tune;
};
int launch_attention(const AttnParams& p, hipStream_t s);   // returns 0 or -1 (unsupported ntok)
constexpr int ATT_PART_FLOATS = 66;       // 64 output features + maximum + sum

// LayerNorm over rows of D in {768,1024}; fp32 in, fp16 (hi[,lo]) and/or fp32 out.
struct LnParams {
    const float* x; int64_t x_stride;     // row stride in elements
    const float* add;                     // optional second addend (same stride as x)   [unused when null]
    const float* gamma; const float* beta;
    int rows, D; float eps;
    f16* out_hi; f16* out_lo;             // [rows][D] dense (nullable); blk layout when out_kt > 0
    int out_kt;
    float* out_f32; int64_t out_f32_stride;   // nullable
    unsigned char* out_q; unsigned char* out_sc;   // nullable: fp4 side planes of the output (quant4.h), blk path only; needs out_lo semantics internally
    int out_q_hi_only;                    // 1: only plane 0 (Q(x_hi)) and its scales are written (the consumer is a one-term compensated GEMM)
    const KeepTune* tune;
};
int launch_layernorm(const LnParams& p, hipStream_t s);

// fp32 "NT" GEMM on the f32 MFMA: out[m][n] = act(scale * sum_k A[m][k] B[n][k] + bias[n])
enum SgemmAct : int { ACT_NONE = 0, ACT_GELU = 1, ACT_TANH = 2 };
struct SgemmParams {
    const float* a; int64_t lda;
    const float* b; int64_t ldb;
    float* out; int64_t ldo;
    const float* bias;                    // nullable
    int M, N, K;                          // K % 16 == 0
    float scale; int act;
    const KeepTune* tune;
};
int launch_sgemm_f32(const SgemmParams& p, hipStream_t s);

// Row-wise helpers (rowops.hip)
void launch_im2col(const void* pixels, int dtype, int B, f16* out_hi, f16* out_lo,      // out in blk layout (KT = 24)
                   const float* cls, const float* pos, float* resid, int D, hipStream_t s);
// Pillow-exact bicubic Resize + CenterCrop of raw uint8 HWC images (weights / windows from keep_amd/preprocess.py)
void launch_resize_crop_u8(const unsigned char* src, int B, int H, int W, const int* xb, const int* xk, int xks, int col0, int ncols,
                           const int* yb, const int* yk, int yks, int row0, int nrows, unsigned char* tmp, unsigned char* out, hipStream_t s);
void launch_split_f16(const float* src, f16* hi, f16* lo, int64_t n, hipStream_t s);
// row-major fp32 [M][K] -> blk-layout fp16 hi (+lo); rows M..pad are zero-filled
void launch_split_blockify(const float* src, f16* hi, f16* lo, int M, int K, hipStream_t s);
// the same plus the MX-fp4 side planes of quant4.h (q: both planes, sc: scales); K % 32 == 0; lo may be null
void launch_quant_blockify(const float* src, f16* hi, f16* lo, unsigned char* q, unsigned char* sc, int M, int K, hipStream_t s);
// blk-layout fp16 hi (+lo) -> row-major fp32 [M][K]
void launch_unblockify_f32(const f16* hi, const f16* lo, float* out, int M, int K, hipStream_t s);
// stats2[0] = max |w| (read as float), stats2[1] = sum w^2 -- the load-time range check of the fp16 operand planes
void launch_weight_stats(const float* w, int64_t n, float* stats2, hipStream_t s);
void launch_l2norm_rows(float* x, int rows, int D, float eps, hipStream_t s, int* err_flag = nullptr);   // err_flag |= 2 if a row is not finite
void launch_row_argmax(const float* x, int rows, int cols, int32_t* out, hipStream_t s);
void launch_row_softmax(const float* x, int rows, int cols, float scale, float* out, hipStream_t s);
void launch_row_softmax_f16(const float* x, int rows, int cols, float scale, f16* out, hipStream_t s);
void launch_top2_score(const float* x, int rows, int cols, float* partial, float* out, hipStream_t s);
void launch_bert_embed_ln(const int64_t* ids, const int64_t* type_ids, const float* wemb, const float* pemb,
                          const float* temb, const float* gamma, const float* beta, float eps,
                          int P, int T, int D, int vocab, int type_vocab,
                          float* resid, f16* out_hi, f16* out_lo /* blk layout */, int* err_flag, hipStream_t s);
// mean-input compensation (keep_calibrate_bias): column sums of a blk-layout fp16 matrix (accumulating, deterministic), and
// out[n] = bias[n] + sum_k w_lo[n][k] * col_sum[k] * inv_rows for a GEMM weight's lo plane
void launch_blk_col_sum(const f16* x, int M, int K, float* out, hipStream_t s);
void launch_bias_mean_corr(const f16* w_lo, const float* col_sum, float inv_rows, const float* bias, float* out, int N, int K, hipStream_t s);
void launch_gather_rows_f32(const float* src, int64_t src_stride, float* dst, int rows, int D, hipStream_t s);
void launch_scatter_rows_f32(const float* src, float* dst, int64_t dst_stride, int rows, int D, hipStream_t s);   // dst row r * dst_stride <- src row r
// blk-layout fp16 [*, D]: dst row r <- src row r * row_stride
void launch_gather_rows_blk(const f16* src, int row_stride, f16* dst, int rows, int D, hipStream_t s);

// wsi.hip
void launch_group_top2(const float* logits, int n, int K, int C, float* partial, int max_row_blocks, float* sums, hipStream_t s);
void launch_top2_slots_reduce(const float* partial, int nslots, int kpad, int K, float scale, float* out, hipStream_t s);
void launch_scale_vec(const float* in, int n, float f, float* out, hipStream_t s);
int launch_sim_small(const float* img, const float* txt, int N, int P, int D, float scale, int mode, void* out, int32_t* amax,
                     hipStream_t s);            // 0 handled, -1 not eligible (P > 8, D not 768/1024)
int launch_sim_mid(const float* img, const float* txt, int N, int P, int D, float scale, int mode, void* out, int32_t* amax,
                   hipStream_t s);              // 9 <= P <= 64: fused fp32-MFMA similarity + argmax / softmax; 0 handled, -1 not eligible
void launch_diag_rank(const float* sim, int rows, int n_img, const int* target, int row0, int* rank, hipStream_t s);
// keep_classify: ordered list of the rows whose top-2 margin is below `bound` (count on the device), tile / row movers
void launch_top2_margin_flags(const float* sim, int N, int P, float bound, int* flags, int* list, int* count, hipStream_t s);
void launch_gather_tiles(const void* src, int64_t tile_bytes, const int* list, int n, void* dst, hipStream_t s);     // tile_bytes % 16 == 0
void launch_scatter_rows(const float* src, const int* list, int n, int D, float* dst, hipStream_t s);
void launch_refine(const float* probs, const long long* coords, int n, int C, long long patch, int overlap,
                   unsigned long long* keys, int* first, unsigned table_size, float* out, int* is_first, hipStream_t s);

// launch_util.hip: per-(kernel, device) opt-in to more than 64 KiB of dynamic LDS (cached, thread-safe; a refusal leaves no sticky HIP error behind),
// and the CU count of the current device
bool keep_lds_opt_in(const void* kernel, size_t bytes);
int keep_num_cus();

enum PixelDType : int { PIX_F32 = 0, PIX_F16 = 1, PIX_BF16 = 2, PIX_U8_HWC = 3 };
