/*
 * libkeep_hip -- C ABI of the MI355X (gfx950) KEEP zero-shot inference engine.
 *
 * This is the drop-in boundary for the hot path of MAGIC-AI4Med/KEEP: everything below
 * `KEEPModel.encode_image` / `encode_text` / the tile x prompt similarity.  Each entry point names
 * the reference interface it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *   - all functions return 0 on success or a negative KEEP_E* code; keep_last_error(h) gives text
 *   - every data pointer is a raw DEVICE pointer on the handle's GPU unless stated otherwise
 *     (e.g. torch.Tensor.data_ptr()); the caller owns all I/O buffers, the engine owns weights,
 *     repacked fp16 planes and its workspace
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous on
 *     it, with no hidden device synchronisation once the workspace is large enough
 *     (keep_reserve() up front avoids the allocation a first call would otherwise do)
 *   - a handle is not thread-safe: one handle per GPU per host thread
 *   - no torch / C++ types cross this boundary
 *   - multi-GPU: one handle per GPU per process.  The path shards without a data-path collective (tiles are independent through
 *     keep_encode_image and keep_similarity); the one exchange step of a slide -- the all-gather of the per-tile embeddings this
 *     library writes into the caller's buffer -- is the HOST's: ncclAllGather / torch.distributed.all_gather_into_tensor on that
 *     buffer, ordered after the encode by the stream both were given (keep_amd/distributed.py).  There is deliberately no
 *     keep_allgather(): the library neither links RCCL nor owns a communicator.
 */
#ifndef KEEP_HIP_H
#define KEEP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct keep_handle keep_handle;

enum {
    KEEP_OK = 0,
    KEEP_EINVAL = -1,       /* bad argument / shape / dtype                                  */
    KEEP_ESTATE = -2,       /* call out of order (weights not finalised, ...)                */
    KEEP_EKEY = -3,         /* unexpected or missing state_dict key (load_state_dict strict) */
    KEEP_EHIP = -4,         /* HIP runtime error                                             */
    KEEP_EUNSUPPORTED = -5, /* shape outside what the kernels implement                      */
    KEEP_ENOMEM = -6
};

/* pixel dtypes accepted by keep_encode_image */
enum { KEEP_PIX_F32 = 0, KEEP_PIX_F16 = 1, KEEP_PIX_BF16 = 2,   /* [B,3,224,224] NCHW, already ImageNet-normalised */
       KEEP_PIX_U8_HWC = 3 };                                   /* [B,224,224,3] raw uint8 RGB: ToTensor + Normalize fused on the device */

/* similarity modes */
enum {
    KEEP_SIM_RAW = 0,         /* out f32 [N,P] = scale * img @ txt^T          keep_inference.py:104          */
    KEEP_SIM_ARGMAX = 1,      /* RAW + argmax_out int32 [N] (out may be NULL)                                */
    KEEP_SIM_SOFTMAX = 2,     /* out f32 [N,P] = softmax(scale*cos, dim=1)    subtyping_utils.py:72 (scale 10)*/
    KEEP_SIM_SOFTMAX_F16 = 3, /* same, out fp16 [N,P]                          (BASELINE config 5)            */
    KEEP_SIM_TOP2SCORE = 4    /* out f32 [1] = mean_t[(v1-v2)-|v1+v2-1|]      WSI_evaluation/utils.py:107-117 */
};

/* precision modes (keep_set_option "precision") */
enum {
    KEEP_PREC_FP16 = 0,   /* fp16 MFMA operands, fp32 accumulate, fp32 residual/LN/softmax/GELU: fastest, cosines    */
                          /* within ~1.5e-4 of the fp32 reference (outside the 1e-4 north-star tolerance)          */
    KEEP_PREC_STRICT = 1, /* hi/lo split operands, 3 MFMA passes: fp32-class accuracy (4e-7), ~0.4x the speed       */
    KEEP_PREC_COMP = 2    /* DEFAULT.  fp16 pass + first-order correction terms where the error budget needs them: */
                          /* a per-block plan for the image tower (keep_set_block_precision: MLP GEMMs on the       */
                          /* MX-fp4 MFMA pipe, attention side as split products), the whole text tower as split     */
                          /* products.  Cosines within 1e-4 of the fp32 reference.                                  */
};

/* per-block treatment of the image tower in KEEP_PREC_COMP (keep_set_block_precision) */
enum {
    KEEP_ATTN_PLAIN = 0,          /* qkv, q/k/v storage, attention, proj: single fp16 passes                                       */
    KEEP_ATTN_SPLIT = 1,          /* all four as split products (three fp16 passes; q/k/v and the attention output stored hi + lo)  */
    KEEP_ATTN_SPLIT_COMPQKV = 2,  /* the same with the qkv GEMM as a compensated product (fp16 pass + MX-fp4 correction terms)      */
    KEEP_ATTN_COMPQKV = 3,        /* only the qkv GEMM compensated; attention and proj plain                                       */
    KEEP_ATTN_PROJ_CLS = 4        /* single fp16 passes for every row; the attention output of the CLS row of every tile is ALSO kept */
                                  /* hi + lo from the fp32 accumulators and its proj runs again as a split product (B rows): on      */
                                  /* spatially correlated tiles the proj GEMM carries ~70 % of the attention side's rounding error   */
                                  /* and the CLS row's own share of it is what reaches the pooled feature (round 6)                  */
    , KEEP_ATTN_COMPQKV_PROJ_CLS = 5 /* KEEP_ATTN_COMPQKV and KEEP_ATTN_PROJ_CLS together: compensated qkv GEMM, plain attention, plain     */
                                  /* proj + the CLS rows' proj again as a split product -- between "CLS-row proj" and "everything split"   */
};
enum {
    KEEP_MLP_PLAIN = 0,           /* fc1 / fc2: single fp16 passes                                                                  */
    KEEP_MLP_SPLIT = 1,           /* split products (three fp16 passes)                                                             */
    KEEP_MLP_COMP = 2,            /* compensated: fp16 pass + both first-order terms W_lo A_hi + W_hi A_lo on the MX-fp4 pipe       */
    KEEP_MLP_COMP_W = 3,          /* compensated, weight-rounding term W_lo A_hi only (half the fp4 MFMAs and operand bytes; the    */
                                  /* LayerNorm / GELU epilogues write Q(x_hi) only): removes the W half of the fp16 rounding error  */
    KEEP_MLP_CLS = 4              /* single fp16 passes for every row, then the CLS row of every tile AGAIN as split products: B rows */
                                  /* through LayerNorm-2 -> fc1 -> GELU -> fc2 on the small-M kernels, written over the fp16 result.  */
                                  /* The pooled output IS the CLS row (global_pool='token', keep_inference.py:32-40): its own rounding */
                                  /* errors reach the feature directly, the other 196 rows' only through attention averages          */
};

const char* keep_version(void);

/* ---- lifetime ---------------------------------------------------------------------------------
 * Replaces: `AutoModel.from_config(config)` + `.to(device)`  (quick_start/keep_inference.py:81,
 * WSI_evaluation/zeroshot_subtyping_WSI.py:44-46). */
int keep_create(int device_id, keep_handle** out);
int keep_destroy(keep_handle* h);
const char* keep_last_error(keep_handle* h);
/* Notes collected by keep_load_tensor since the last call ('\n'-separated, "" if none; the call clears them): e.g. a GEMM weight whose rms is so
 * small that its entries fall into fp16 subnormals.  Such a tensor LOADS (torch's load_state_dict, keep_inference.py:83, has no such notion);
 * the Python binding turns each line into a warnings.warn.  The pointer is valid until the calling thread's next keep_load_warnings call. */
const char* keep_load_warnings(keep_handle* h);

/* ---- weights ----------------------------------------------------------------------------------
 * Replaces: `model.load_state_dict(state_dict, strict=True)`  (quick_start/keep_inference.py:82-83).
 * Call keep_load_tensor once per state_dict entry using the release key names (SURVEY.md A.3:
 * "visual.blocks.3.attn.qkv.weight", "text.encoder.layer.0.attention.self.query.weight", ...),
 * fp32 data, then keep_finalize_weights(), which fails with KEEP_EKEY if any expected key is missing
 * (strict semantics).  Data is copied / repacked; the caller's buffer may be freed afterwards.
 * `on_device` != 0: `data` is a device pointer on the handle's GPU; 0: host pointer. */
int keep_load_tensor(keep_handle* h, const char* key, const float* data, int ndim, const int64_t* shape,
                     int on_device);
int keep_finalize_weights(keep_handle* h);
/* after finalize: depth / layer counts actually loaded (0 if that tower was not loaded) */
int keep_vit_depth(keep_handle* h);
int keep_bert_layers(keep_handle* h);

/* ---- options ----------------------------------------------------------------------------------
 *   "precision"       KEEP_PREC_COMP (default) | KEEP_PREC_FP16 | KEEP_PREC_STRICT
 *   "strict_blocks"   run the first n ViT blocks (+ patch embed) / BERT layers in split mode (default 0)
 *   "comp_full_blocks" KEEP_PREC_COMP, prefix shorthand: the first n ViT blocks get KEEP_ATTN_SPLIT, the rest KEEP_ATTN_PLAIN (1 unless set)
 *   "comp_mlp_blocks"  KEEP_PREC_COMP, prefix shorthand: the first n ViT blocks get KEEP_MLP_COMP, the rest KEEP_MLP_PLAIN (8 unless set).
 *                     Setting any of the four comp_* shorthands REWRITES the whole per-block plan (keep_set_block_precision below) to that prefix
 *                     family.  A handle STARTS with another plan: block 0 KEEP_ATTN_SPLIT_COMPQKV + KEEP_MLP_COMP, every other block
 *                     KEEP_ATTN_PLAIN + KEEP_MLP_CLS -- until KEEPModel.calibrate / keep_set_block_precision replace it.
 *   "plan_custom"      (read only) 1 if keep_set_block_precision changed the plan since the last shorthand
 *   "comp_min_tiles"   sub-batches with fewer tiles use split products instead of compensated ones (default 32)
 *   "comp_qkv"         KEEP_PREC_COMP: 1 = the qkv GEMM of the split-attention blocks as a compensated product instead of a split one
 *                     (default 0: +0.9 % at equal settings, but calibrate() then needs more compensated MLP blocks -- a net loss)
 *   "comp_qkv_from"    the same for the split-attention blocks with index >= n only (default: none; round 4: 2 / 6 with n = 1 misses the rms target 1 / 8 meets)
 *   "label_margin"     keep_classify: cosine margin below which a tile's label is re-derived in KEEP_PREC_STRICT (default 2.5e-4)
 *   "fused_screening"  keep_prompt_scores with C in {2, 4}: 1 (default) one compensated GEMM with the top-2 score taken in the
 *                     accumulator registers (no logits in HBM) | 2 the same with three fp16 passes | 0 chunked fp32 GEMM + reduction
 *   "max_tiles"       tiles per internal sub-batch of keep_encode_image (default 256)
 *   "max_prompts"     prompts per internal sub-batch of keep_encode_text (default 64)
 *   "streams"         concurrent sub-batches inside keep_encode_image (default 2, 1..4): the batch is split
 *                     into that many lanes on internal HIP streams, issued layer-interleaved, and joined
 *                     back onto the caller's stream with events (no host synchronisation)
 *   "cls_tail"        1 (default): in the last ViT block run proj / MLP for the CLS rows only (exact: the
 *                     pooled output reads nothing else); 0: evaluate every token as the reference does
 *   "cls_qkv"         1: with cls_tail, the last block's qkv GEMM computes K | V for every token and Q for the CLS rows only (exact).  Default 0: measured level
 *                     end to end (-0.09 ms of qkv, +0.02 ms of small launches per 256-tile step)
 *   "proj_impl"       2128 (default): the plain proj GEMMs on the 256x128 / 4-wave / two-workgroups-per-CU kernel (one workgroup's residual epilogue under the
 *                     other's K loop): -8.5 % on the proj launches, +0.57 % end to end (profiles/r05_ab_two_workgroups_per_cu.txt) | 0: the persistent 256x256 kernel.
 *                     Bit-identical results.  "impl2128_mask" (experiments): the same kernel for qkv (1) / fc1 (4) / fc2 (8): measured -1.6 ... -4.3 %
 *   "bias_correction" 1 (default): plain launches use the mean-input-compensated biases once keep_calibrate_bias has run | 0: the checkpoint's biases
 *   "gemm_impl"       0 auto | 128 | 256: LDS-DMA tile width override.  Like every option it belongs to the handle.
 *   "graphs"          1 (default): launch-bound calls -- keep_encode_image of at most 1024 token rows (5 tiles), keep_encode_text of at most
 *                     4096 token rows (e.g. 64 prompts x 64 tokens) -- are captured once per shape and replayed as one hipGraph launch (~100
 *                     dependent kernels of a few microseconds each); 0: always launch kernels
 *   "gemm_skinny_m"   calls with at most this many rows take the small-M split-K GEMM (default 320, 0 never).  The two GEMM paths agree to rounding, each is bit-reproducible
 *   "gemm_splitk_tiles"  a larger call whose 256x256 tiling has fewer tiles than this (default 64, 0 never) is cut into
 *                     K slices with fp32 partials + the same reduce/epilogue kernel (8-16 tiles per call: -16..-26 %)
 *   "gemm_persistent" 1 (default): a plain (one fp16 pass, hi-only output) 256x256 GEMM with more tiles than the device has CUs runs as one
 *                     workgroup per CU walking the tile list, the next tile's first three K steps staged under the epilogue; 0: one
 *                     tile per workgroup.  Bit-identical results either way.
 *   "sgemv_m"         same for the few-row fp32 kernel of the projection head / pooler / similarity (default 16)
 *   "ln_impl"         2 (default) LayerNorm with LDS-transposed K-blocked stores from 4-wave workgroups (co-resident with the other lane's persistent fc1 GEMM:
 *                     +0.3 % end to end) | 1 the same from 8-wave workgroups | 0 per-row stores.  Same results bit for bit.
 *   "attn_waves"      16 (default): image-tower attention of >= 512 (image, head) pairs on the persistent double-buffered kernel (bit-identical
 *                     to 8, the K / V staging of the next pair runs under the current one's compute), 8 waves per workgroup elsewhere | 8 | 4
 *   "lane0_permille", "lane_skew"   experiments with the two-lane schedule (defaults 500 / 0 measured best)
 *   "gemm_ablate", "gemm_dbg", "dbg_skip_ln"   timing diagnostics (results are wrong when ablating): -DKEEP_DIAGNOSTICS builds only
 */
int keep_set_option(keep_handle* h, const char* name, double value);
double keep_get_option(keep_handle* h, const char* name);
/* The per-block plan of KEEP_PREC_COMP: block `block` (0 .. 63) of the image tower gets attention-side treatment `attn_mode` (KEEP_ATTN_*) and
 * MLP treatment `mlp_mode` (KEEP_MLP_*); a negative mode leaves that half as it is.  Replaces nothing in the reference (which computes in fp32,
 * quick_start/keep_inference.py:54-58): it is how this engine spends its precision budget -- where the fp16 rounding of a block matters for the
 * final cosine is a property of the checkpoint, measured by KEEPModel.calibrate / tools/precision_budget.py.  Sub-batches below "comp_min_tiles"
 * run split products wherever a compensated one is planned.  KEEP_PREC_STRICT / "strict_blocks" override the plan; KEEP_PREC_FP16 ignores it. */
int keep_set_block_precision(keep_handle* h, int block, int attn_mode, int mlp_mode);
/* Mean-input compensation of the weight-rounding error (replaces nothing in the reference, which multiplies in fp32: quick_start/keep_inference.py:54-58).
 * A single-pass fp16 GEMM drops the term W_lo A_hi^T.  Part of it is the same for every row: W_lo a_mean, a_mean = the mean input row of that GEMM (GELU
 * outputs are positive, LayerNorm outputs carry their bias, attention outputs are averages) -- 10-50 % of that GEMM's weight-rounding variance on the
 * synthetic checkpoints, and the one part of the rounding error that does not average out over a slide's tiles.  It is a constant vector per GEMM: this call
 * encodes the B calibration tiles (>= 8; KEEP_PIX_* layouts as keep_encode_image) in split products, averages the input rows of the four GEMMs of every ViT
 * block and stores bias + W_lo a_mean; every PLAIN launch (KEEP_ATTN_PLAIN / KEEP_MLP_PLAIN, KEEP_PREC_FP16) then uses that bias -- zero cost per call.  Split and
 * compensated launches compute the term itself and keep the checkpoint's bias.  B = 0 forgets the calibration; loading weights forgets it too.
 * Option "bias_correction" (default 1) switches the use on and off; option "bias_ready" (read only) says whether a calibration is held. */
int keep_calibrate_bias(keep_handle* h, const void* pixels, int pix_dtype, int64_t B, void* stream);
int keep_get_block_precision(keep_handle* h, int block, int* attn_mode, int* mlp_mode);

/* ---- preprocessing on the device -----------------------------------------------------------------
 * Replaces: transforms.Resize(224, BICUBIC) + CenterCrop((224,224))  (quick_start/keep_inference.py:88-90) for raw uint8
 * RGB images [B,H,W,3] of one size.  The fixed-point windows / weights of Pillow's resample are built by the caller
 * (keep_amd/preprocess.py: pil_bicubic_coeffs) -- bounds [out,2] = (first input index, count), weights [out,ksize] -- and the
 * two integer passes run here, so the result is bit-identical to PIL's.  out: uint8 [B,size,size,3], ready for
 * keep_encode_image(..., KEEP_PIX_U8_HWC, ...). */
int keep_resize_crop_u8(keep_handle* h, const unsigned char* src, int64_t B, int64_t H, int64_t W, const int32_t* xbounds,
                        const int32_t* xweights, int xksize, int64_t out_w, const int32_t* ybounds, const int32_t* yweights, int yksize,
                        int64_t out_h, int64_t crop_left, int64_t crop_top, int64_t size, unsigned char* out, void* stream);

/* Pre-allocate workspace for calls of up to `tiles` tiles and `prompts` x `seq` tokens. */
int keep_reserve(keep_handle* h, int64_t tiles, int64_t prompts, int64_t seq);
int64_t keep_workspace_bytes(keep_handle* h);

/* ---- the hot path -----------------------------------------------------------------------------
 * Replaces: KEEPModel.encode_image  (quick_start/keep_inference.py:54-58)
 *   pixels: [B,3,224,224] NCHW, ImageNet-normalised, dtype per `pix_dtype` (or raw uint8 [B,224,224,3] with
 *   KEEP_PIX_U8_HWC: the /255 and mean/std steps of keep_inference.py:91-92 run on the device); out: fp32 [B,768],
 *   L2-normalised (F.normalize semantics). */
int keep_encode_image(keep_handle* h, const void* pixels, int pix_dtype, int64_t B, float* out, void* stream);

/* Replaces: KEEPModel.encode_text  (quick_start/keep_inference.py:60-62)
 *   input_ids / token_type_ids / attention_mask: int64 [P,T] (the tokenizer's return_tensors='pt'
 *   layout, keep_inference.py:99); token_type_ids and attention_mask may be NULL (zeros / ones, as
 *   HF BertModel defaults).  out: fp32 [P,768] L2-normalised.  T <= 512 (BertModel's max_position_embeddings) in every precision mode: above 256 keys
 *   the split-product attention of KEEP_PREC_COMP / KEEP_PREC_STRICT runs over two key windows and merges them (the exact softmax over all keys). */
int keep_encode_text(keep_handle* h, const int64_t* input_ids, const int64_t* token_type_ids,
                     const int64_t* attention_mask, int64_t P, int64_t T, float* out, void* stream);

/* The handle's sticky error bits, set by encode calls on `stream` SINCE THE LAST TIME THIS RETURNED NON-ZERO:
 *   bit 0 (1): a token / type id of a keep_encode_text call was out of range (the kernel clamps it; the reference's nn.Embedding
 *              would raise IndexError);
 *   bit 1 (2): an output feature row of keep_encode_image / keep_encode_text was not finite: an activation left the fp16 range
 *              (|x| > 65504 in a qkv / MLP-hidden store -- conversions do not saturate, so the overflow reaches the output as NaN
 *              instead of as plausible garbage; the fp32 reference would not overflow).
 * Encode calls only ever SET bits; they are cleared here, once the host has seen them, so an error can not be lost between calls.
 * Synchronises `stream`. */
int keep_token_error(keep_handle* h, void* stream);
/* The same without a host synchronisation: enqueues a copy of the (sticky) flag into `host_flag` (pinned host memory owned by the
 * caller, which must stay alive until the stream has passed that point) behind the work already on `stream`.  Does not clear the flag:
 * a caller that reads a 1 calls keep_token_error() to acknowledge it.  This is how the reference's CUDA path reports an out-of-range
 * index too: asynchronously. */
int keep_token_error_async(keep_handle* h, int32_t* host_flag, void* stream);

/* Replaces: `img_feature @ text_feature.T` (keep_inference.py:104), `image_features @ cls`
 * (WSI_evaluation/utils.py:128) and the softmax / top-2 score that follow it in the WSI scripts.
 *   img fp32 [N,D], txt fp32 [P,D] (row-major; a reference classifier [D,C] is passed transposed). */
int keep_similarity(keep_handle* h, const float* img, const float* txt, int64_t N, int64_t P, int64_t D,
                    float scale, int mode, void* out, int32_t* argmax_out, void* stream);

/* Replaces: `img_feature = model.encode_image(img_input)` + `img_feature @ text_feature.T` + the row argmax taken from it
 * (quick_start/keep_inference.py:101,104; BASELINE config 3) when the LABELS must be the fp32 reference's.
 *   The default precision keeps every cosine within 1e-4 of the fp32 reference, which cannot decide a tile whose two best prompts are
 *   closer than that.  keep_classify encodes all B tiles in the handle's precision, takes sim = scale * feats @ txt^T and its row argmax,
 *   then re-encodes ONLY the tiles whose top-2 margin (in cosine units, i.e. / scale) is below `margin` in KEEP_PREC_STRICT (split
 *   products, ~5e-7) and takes those rows again.  margin < 0: the handle's "label_margin" option (default 2.5e-4 = 2 x tolerance + 25 %);
 *   margin == 0: no second look.  One host synchronisation of `stream` (the number of flagged tiles).
 *   The labels are those of the split-product arithmetic provided the first pass is within margin / 2 of it: true in KEEP_PREC_COMP (<= 1e-4 against
 *   2.5e-4); in KEEP_PREC_FP16 (~2e-4) pass a margin of at least twice that mode's error.
 *   pixels / pix_dtype / B as keep_encode_image (16-byte aligned); txt fp32 [P,D] L2-normalised text features (keep_encode_text);
 *   feats_out fp32 [B,D] or NULL; sim_out fp32 [B,P] or NULL; labels_out int32 [B] (first maximum wins, as torch.argmax);
 *   n_rechecked (HOST pointer or NULL): how many tiles were encoded twice. */
int keep_classify(keep_handle* h, const void* pixels, int pix_dtype, int64_t B, const float* txt, int64_t P, float scale, float margin,
                  float* feats_out, float* sim_out, int32_t* labels_out, int64_t* n_rechecked, void* stream);

/* ---- slide-level zero-shot steps ---------------------------------
 * Replaces the loop of `zero_shot_prompt_select` (WSI_evaluation/utils.py:127-130: one GEMM + one
 * rank_cls_score + one .item() sync per prompt classifier).  feats fp32 [N,D] already L2-normalised
 * (utils.py:125), bank fp32 [K*C, D] = the K classifiers stacked, classifier k occupying rows k*C..k*C+C-1
 * (i.e. each [D,C] classifier transposed); scores_out fp32 [K] (device) = rank_cls_score of every k. */
int keep_prompt_scores(keep_handle* h, const float* feats, const float* bank, int64_t N, int64_t K, int64_t C,
                       int64_t D, float* scores_out, void* stream);

/* ---- tile-level zero-shot evaluation ------------------------------------------
 * Replaces the 50-round loop of training/path_training/zero_shot.py:124-136 (per round: one numpy GEMM
 * `image_embeddings.dot(each_round.T)` + a Python argmax per tile).  feats fp32 [N,D] L2-normalised (zero_shot.py:
 * 121-122), bank fp32 [K*C, D]: round k's C normalised class embeddings in rows k*C..k*C+C-1.
 * labels_out int32 [N,K] (device): argmax class of tile n in round k, first maximum wins (numpy.argmax). */
int keep_group_argmax(keep_handle* h, const float* feats, const float* bank, int64_t N, int64_t K, int64_t C,
                      int64_t D, int32_t* labels_out, void* stream);

/* Replaces the retrieval loop of zero_shot.py:168-171 + retrieval_metrics (zeroshot_metrics.py:6-17).
 * txt fp32 [P,D], img fp32 [N,D], both L2-normalised; target int32 [P] = the image each text must retrieve
 * (NULL: text t -> image t, as the reference).  rank_out int32 [P] (device) = position of the target in the
 * descending score list (0 = best; equal scores: higher index first); p@k = mean(rank < k). */
int keep_retrieval_rank(keep_handle* h, const float* txt, const float* img, int64_t P, int64_t N, int64_t D,
                        const int32_t* target, int32_t* rank_out, void* stream);

/* Replaces `refine_seg` (subtyping_utils.py:38-65, detection_utils.py:39-74, segment_utils.py:63-89).
 * probs fp32 [N,C] (softmax(10*cos)), coords int64 [N,2].  out_mean fp32 [N,C]: for the first tile of
 * every distinct coordinate, the float32 mean of the existing tiles among (x-p,y-p),(x,y-p),(x-p,y),(x,y)
 * (or its own row when overlap == 0); is_first int32 [N]: 1 for those tiles, 0 for later duplicates. */
int keep_refine(keep_handle* h, const float* probs, const int64_t* coords, int64_t N, int64_t C, int64_t patch,
                int overlap, float* out_mean, int32_t* is_first, void* stream);

/* ---- profiling (HIP events on the launch stream) ----------------------------------------------
 * tag names: "vit.im2col" "vit.patch" "vit.ln" "vit.qkv" "vit.attn" "vit.proj" "vit.fc1" "vit.fc2"
 * "vit.head" "text.embed" "text.ln" "text.qkv" "text.attn" "text.out" "text.ffn1" "text.ffn2"
 * "text.pool" "sim", and -- so that each plain image-tower tag times ONE kernel instantiation -- "vit.qkv.x" "vit.attn.x" "vit.proj.x"
 * "vit.fc1.x" "vit.fc2.x" (the launches that carry extra passes: split / compensated products of the blocks the precision setting
 * names) and "vit.tail" (the CLS-rows-only operators of the last block).  keep_profile_enable(h, NULL) times every tag, a name (or
 * several separated by commas) times only those, "" disables.  keep_profile_read synchronises the recorded events and returns the accumulated
 * milliseconds, launch count and (for GEMM tags) executed FLOPs 2*M*N*K for `tag` since the last
 * keep_profile_reset.  Lanes on different internal streams overlap, so per-tag times can sum to more than
 * the wall time. */
int keep_profile_enable(keep_handle* h, const char* tag_or_null);
int keep_profile_read(keep_handle* h, const char* tag, double* total_ms, int64_t* launches, double* flops);
int keep_profile_reset(keep_handle* h);

/* ---- single-operator entry points (used by the parity tests; fp32 in/out on device) -----------
 * keep_op_linear: out = epilogue(A[M,K] @ W[N,K]^T + bias) through the fp16 MFMA GEMM.
 *   epi 0: out[M,N] = acc+bias            1: gelu(acc+bias)
 *       2: out = resid + ls*(acc+bias)    4: out = resid + acc + bias      (resid, ls fp32)
 *   split 1 runs the 3-pass hi/lo product, split 2 the compensated product (fp16 pass + MX-fp4 correction terms;
 *   N % 256 == 0, K % 128 == 0, K >= 256, epi 0..2), split 3 the compensated product with the W_lo A_hi term only (K >= 512).  Outputs of epi 0/1 are the fp16-rounded values (hi, or hi+lo when
 *   split != 0) converted back to fp32. */
int keep_op_linear(keep_handle* h, const float* a, const float* w, const float* bias, const float* ls,
                   const float* resid, int64_t M, int64_t N, int64_t K, int epi, int split, float* out,
                   void* stream);
/* One MLP half of a ViT block through the tower's own kernels (timm Block: x + ls2 * fc2(gelu(fc1(norm2(x))))):
 *   LayerNorm (writes the fp16 operand and, per mode, its lo plane / MX-fp4 side planes) -> fc1 + GELU -> fc2 + LayerScale + residual.
 *   mode = KEEP_MLP_* 0..3: 0 plain fp16 | 1 split | 2 compensated | 3 compensated, W_lo term only.  x, out fp32 [M, D]; D in {768, 1024}; F % 256 == 0. */
int keep_op_mlp(keep_handle* h, const float* x, const float* ln_w, const float* ln_b, const float* fc1_w, const float* fc1_b,
                const float* fc2_w, const float* fc2_b, const float* ls, int64_t M, int64_t D, int64_t F, int mode, float* out,
                void* stream);
/* qkv fp32 [B*T, 3*heads*64] (q|k|v), mask int64 [B,T] or NULL -> out fp32 [B*T, heads*64] */
int keep_op_attention(keep_handle* h, const float* qkv, const int64_t* mask, int64_t B, int64_t T, int heads,
                      int split, float* out, void* stream);
int keep_op_layernorm(keep_handle* h, const float* x, const float* add, const float* gamma, const float* beta,
                      int64_t rows, int64_t D, float eps, float* out, void* stream);
/* out[M,N] = act(scale * A[M,K] @ B[N,K]^T + bias); act 0 none, 1 gelu, 2 tanh (exact fp32 MFMA) */
int keep_op_sgemm(keep_handle* h, const float* a, const float* b, const float* bias, int64_t M, int64_t N,
                  int64_t K, float scale, int act, float* out, void* stream);
int keep_op_l2norm(keep_handle* h, float* x, int64_t rows, int64_t D, void* stream);
/* measurement aid: one wavefront spins for ~spin_us microseconds and writes {shader-clock cycles, 100 MHz reference ticks} to device_out2
 * (two int64 on the device): effective shader clock = cycles / ticks * 100 MHz.  Launched on a side stream next to a running workload it
 * reads the clock the part actually sustains under that load (bench.py records it; the MFMA peak is quoted at 2.4 GHz). */
int keep_clock_probe(keep_handle* h, int spin_us, long long* device_out2, void* stream);
/* measurement aid: what the matrix pipes alone deliver on this part with the caller's operand values.  One 8-wave workgroup per CU; every wave loads
 * 6 fp16 fragments (6 x 16 B per lane: operands_f16 holds CUs x 512 x 48 fp16 values) and issues `iters` x 8 independent v_mfma_f32_32x32x16_f16 with no
 * memory access in the loop; sink (fp32, CUs x 512) receives one value per lane.  *flops_out (host) = the FLOPs of the launch; time it with events on
 * `stream`.  With N(0, 1)-like operands the socket's power cap holds the pipes at ≈1.6 GHz: ≈1 590 TFLOP/s = 0.63 of the nominal dense peak that the roofline
 * fractions are quoted against; with zeros 2 480 (profiles/r04_mfma_power_ceiling.txt).  bench.py reports it as context next to `roofline`. */
int keep_mfma_probe(keep_handle* h, const void* operands_f16, float* sink, int iters, double* flops_out, void* stream);
/* diagnostics: with option "gemm_dbg"=1 every GEMM launch records, per workgroup, four shader-clock
 * stamps [start, first K tile landed, main loop end, end]; this copies them to host memory. */
int keep_debug_read(keep_handle* h, void* host_dst, int64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* KEEP_HIP_H */
