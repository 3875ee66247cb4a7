// Host side of libkeep_hip: handle, weight store (release state_dict key layout), workspace arena,
// tower orchestration and the extern "C" boundary declared in include/keep_hip.h.
//
// Orchestration mirrors the reference's call order, not its code:
//   encode_image  quick_start/keep_inference.py:54-58  -> timm VisionTransformer.forward
//   encode_text   quick_start/keep_inference.py:60-62  -> HF BertModel.forward           
#include "common.h"
#include "quant4.h"
#include "../../include/keep_hip.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define HIPCHK(h, expr)                                                                      \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) return (h)->fail(KEEP_EHIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

namespace {

// encode_text calls of up to this many token rows (64 prompts x 64 tokens after padding trim: a whole classifier bank chunk) are captured once per shape
// and replayed as one graph launch: ~90 dependent kernels of 10-25 us whose host-side issue (1.4 ms) otherwise runs next to them
constexpr int64_t TXT_GRAPH_ROWS = 4096;

// Every entry point runs on the handle's device and puts the caller's current device back (torch keeps its own notion of
// the current device per thread; changing it behind its back redirects the caller's next allocation).
struct DevGuard {
    int prev = -1; bool ok = true;
    explicit DevGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess; else prev = -1;
    }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define KEEP_ON_DEVICE(h) DevGuard _guard((h)->device); if (!_guard.ok) return (h)->fail(KEEP_EHIP, "hipSetDevice(%d) failed", (h)->device)

struct WTensor {
    std::vector<int64_t> shape;
    int64_t numel = 0;
    float* f32 = nullptr;     // kept for vectors / embeddings / head / pooler
    f16* hi = nullptr;        // GEMM weights: fp16 planes
    f16* lo = nullptr;
    unsigned char* q = nullptr;   // MX-fp4 side planes of (hi, lo) and their scales (quant4.h); K % 128 == 0 weights only
    unsigned char* sc = nullptr;
    float prescale = 1.f;     // the planes hold prescale * W (a power of two; proj / fc2 of the image tower only): folded back through LayerScale and bias at finalize
};

enum Tag {
    T_VIT_IM2COL, T_VIT_PATCH, T_VIT_LN, T_VIT_QKV, T_VIT_ATTN, T_VIT_PROJ, T_VIT_FC1, T_VIT_FC2, T_VIT_HEAD,
    T_TXT_EMBED, T_TXT_LN, T_TXT_QKV, T_TXT_ATTN, T_TXT_OUT, T_TXT_FFN1, T_TXT_FFN2, T_TXT_POOL, T_SIM,
    // image-tower launches that are NOT the plain single-pass kernel of their operator: split products / compensated (MX-fp4) products of the
    // blocks the precision setting names ("x" = extra passes), and the CLS-rows-only operators of the last block ("tail": small-M kernels).
    // The plain tags above then time one kernel instantiation each (bench.py's roofline block needs a per-kernel figure).
    T_VIT_QKV_X, T_VIT_ATTN_X, T_VIT_PROJ_X, T_VIT_FC1_X, T_VIT_FC2_X, T_VIT_TAIL, T_COUNT
};
const char* kTagNames[T_COUNT] = {
    "vit.im2col", "vit.patch", "vit.ln", "vit.qkv", "vit.attn", "vit.proj", "vit.fc1", "vit.fc2", "vit.head",
    "text.embed", "text.ln", "text.qkv", "text.attn", "text.out", "text.ffn1", "text.ffn2", "text.pool", "sim",
    "vit.qkv.x", "vit.attn.x", "vit.proj.x", "vit.fc1.x", "vit.fc2.x", "vit.tail"};

struct VitBlock {
    const float *n1w, *n1b, *n2w, *n2b, *qkv_b, *proj_b, *fc1_b, *fc2_b, *ls1, *ls2;
    const WTensor *qkv, *proj, *fc1, *fc2;
};
struct BertLayer {
    WTensor qkv;                 // fused [3H, H]
    float* qkv_b = nullptr;      // fused [3H]
    const float *o_b, *ln1w, *ln1b, *i_b, *d_b, *ln2w, *ln2b;
    const WTensor *o, *i, *d;
};

}  // namespace

struct keep_handle {
    int device = 0;
    std::string err;
    std::string load_warnings;   // '\n'-separated notes of keep_load_tensor calls (a weight the fp16 planes resolve poorly); read and cleared by keep_load_warnings
    std::map<std::string, WTensor> w;
    bool finalized = false;

    // dims (filled at finalize)
    int vit_depth = 0, vit_D = 0, vit_heads = 0, vit_F = 0, proj_dim = 0;
    int bert_layers = 0, bert_H = 0, bert_heads = 0, bert_F = 0, bert_vocab = 0, bert_maxpos = 0, bert_types = 0;
    std::vector<VitBlock> vblocks;
    std::vector<BertLayer> blayers;
    std::vector<float*> owned_vecs;   // LayerScale / bias vectors re-derived for pre-scaled weights (finalize_vit)

    // options
    KeepTune tune;               // kernel selection (travels in the launch parameter blocks; nothing is process-wide)
    int precision = KEEP_PREC_COMP;
    int strict_blocks = 0;       // first n blocks / layers as full hi/lo split products (any mode)
    // The prefix shorthands (state of the last keep_set_option; they only take effect once one of them is set -- see plan_default below)
    int comp_full_blocks = 1;    // KEEP_PREC_COMP: first n ViT blocks run qkv / attention / proj as split products as well
    int comp_mlp_blocks = 8;     // KEEP_PREC_COMP: first n ViT blocks run fc1 / fc2 as compensated (fp16 + MX-fp4) products
    int fused_screening = 1;     // keep_prompt_scores: 1 fused compensated GEMM (default) | 2 fused 3-pass split GEMM | 0 logits through HBM (any C)
    int comp_min_tiles = 32;     // lanes with fewer tiles take the split product where a compensated one is asked for (small-M kernels)
    int comp_qkv = 0;            // 1: KEEP_PREC_COMP, blocks < comp_full_blocks: the qkv GEMM as a compensated product (x1.5) instead of a split one (x3);
                                 // q / k / v still stored as hi + lo planes, attention still a split product.  Measured (round 3): +0.9 % at equal settings,
                                 // but block 0's qkv is where the error budget is tightest (the 3 % of the rounding variance the fp4 terms leave is
                                 // amplified by all 24 blocks): calibrate() then needs 10 compensated MLP blocks instead of 6 -- a net loss.  Off.
    int comp_qkv_from = 1 << 20; // the same for the split-attention blocks with index >= this only (block 0 keeps its three-pass qkv)
    // The per-block plan of KEEP_PREC_COMP (keep_set_block_precision; the four options above are prefix shorthands that rewrite it):
    //   attn_mode[i]  attention side of block i: KEEP_ATTN_PLAIN | KEEP_ATTN_SPLIT (qkv, q/k/v storage, attention, proj as split products) |
    //                 KEEP_ATTN_SPLIT_COMPQKV (the same with the qkv GEMM as a compensated product) | KEEP_ATTN_COMPQKV (compensated qkv only) |
    //                 KEEP_ATTN_PROJ_CLS (plain for every row + the CLS rows' proj again as a split product on their fp32-grade attention output) |
    //                 KEEP_ATTN_COMPQKV_PROJ_CLS (both of the last two)
    //   mlp_mode[i]   fc1 / fc2 of block i: KEEP_MLP_PLAIN | KEEP_MLP_SPLIT | KEEP_MLP_COMP (both MX-fp4 correction terms) | KEEP_MLP_COMP_W (the W_lo term only) |
    //                 KEEP_MLP_CLS (plain for every row + the CLS rows again as split products)
    // Which block gets what is a measured, per-checkpoint decision (tools/precision_budget.py, KEEPModel.calibrate).
    static constexpr int MAX_BLOCKS = 64;
    unsigned char attn_mode[MAX_BLOCKS] = {}, mlp_mode[MAX_BLOCKS] = {};
    bool plan_custom = false;    // keep_set_block_precision was called since the last prefix option
    void plan_from_prefix() {
        for (int i = 0; i < MAX_BLOCKS; ++i) {
            attn_mode[i] = i < comp_full_blocks ? ((comp_qkv || i >= comp_qkv_from) ? KEEP_ATTN_SPLIT_COMPQKV : KEEP_ATTN_SPLIT) : KEEP_ATTN_PLAIN;
            mlp_mode[i] = i < comp_mlp_blocks ? KEEP_MLP_COMP : KEEP_MLP_PLAIN;
        }
        plan_custom = false;
    }
    // The plan a handle starts with (no calibration has seen the weights yet): block 0's attention side as split products with a compensated qkv
    // and its MLP compensated -- the first block's rounding errors, in EVERY row, are amplified by all the attention layers that follow: 45-48 % of the
    // all-fp16 error variance on the synthetic checkpoints --, every other block plain with the CLS rows' MLP redone as split products (the pooled
    // feature is a CLS row).  profiles/r05_precision_budget.md: cosine rms 8.5e-6 on the bench weights, a quarter of what the 1e-4 tolerance allows a
    // 100 000-tile slide, 3 % slower than what KEEPModel.calibrate picks for them.
    void plan_default() {
        for (int i = 0; i < MAX_BLOCKS; ++i) { attn_mode[i] = KEEP_ATTN_PLAIN; mlp_mode[i] = KEEP_MLP_CLS; }
        attn_mode[0] = KEEP_ATTN_SPLIT_COMPQKV; mlp_mode[0] = KEEP_MLP_COMP;
        plan_custom = false;
    }
    keep_handle() { plan_default(); }
    // keep_classify: tiles whose top-2 cosine margin is below this are re-encoded in KEEP_PREC_STRICT before their label is taken.
    // Default = 2 x the north-star tolerance (both cosines of a pair can move by 1e-4 in opposite directions) + 25 %.
    float label_margin = 2.5e-4f;
    char* cls_buf = nullptr; size_t cls_bytes = 0;     // keep_classify scratch (features, similarity, flags, staged tiles): outside the arena, which the encodes carve
    int max_tiles = 256;
    int max_prompts = 64;
    int cls_tail = 1;            // last ViT block: proj / MLP on the CLS rows only (exact; 0 = evaluate every token)
    // Mean-input compensation of the weight-rounding error (keep_calibrate_bias).  A plain fp16 GEMM computes A_hi W_hi^T: the W_lo A_hi term it drops has
    // a part that is the SAME for every row -- W_lo a_mean, a_mean = the mean input row of that GEMM (GELU outputs are positive, LayerNorm outputs carry their
    // bias, attention outputs are averages) -- which no amount of averaging over tiles removes.  It is a constant vector per GEMM: folded into the bias the plain
    // launches use.  cal[i].sum[site]: column sums of the site's input over the calibration tiles; cal[i].bias[site]: bias + W_lo a_mean.
    struct SiteCal { float* sum[4] = {nullptr, nullptr, nullptr, nullptr}; float* bias[4] = {nullptr, nullptr, nullptr, nullptr}; double rows[4] = {0, 0, 0, 0}; };
    std::vector<SiteCal> cal;    // per ViT block; sites: 0 qkv, 1 proj, 2 fc1, 3 fc2
    bool capture = false;        // the running encode accumulates cal[i].sum
    bool bias_ready = false;     // cal[i].bias hold corrected biases for the loaded weights
    int cal_cls_tail = 1;        // cls_tail at the time of the calibration (switching it afterwards invalidates the last block's averages)
    int bias_correction = 1;     // plain launches use them (0: the checkpoint's own biases)
    void free_cal() {
        for (auto& c : cal) for (int k = 0; k < 4; ++k) { if (c.sum[k]) (void)hipFree(c.sum[k]); if (c.bias[k]) (void)hipFree(c.bias[k]); }
        cal.clear(); bias_ready = false;
    }
    int patch_split = 1;         // 0: the patch-embedding GEMM as one fp16 pass (experiments; measured in profiles/r05_patch_embed_plain.txt)
    int cls_qkv = 0;             // 1: last ViT block (with cls_tail): the q part of the qkv GEMM for the CLS rows only (exact).  Measured (round 5, tools/ab_options.py):
                                 // vit.qkv -0.09 ms per step on one stream, +0.02 ms of small launches, 6041 vs 6044 tiles/s end to end with two lanes: below the
                                 // 0.3 % it would have to return -- off by default
    // 2128 (default): the plain proj GEMMs of the image tower on the 256x128 / 4-wave / two-workgroups-per-CU kernel (GemmParams.impl_hint) -- proj is the one GEMM whose
    // tile is 40 % fp32 residual read-modify-write epilogue, and with two workgroups on a CU one's epilogue runs under the other's K loop: -8.5 % on the proj launches,
    // +0.57 % end to end in a six-round rotated A/B on the round-5 plan (profiles/r05_ab_two_workgroups_per_cu.txt; qkv / fc1 / fc2 on the same kernel lose 1.6-4.3 %:
    // 1.5 x the operand bytes per FLOP).  0: the persistent 256x256 kernel.  Bit-identical results either way (same K order per output).
    int proj_impl = 2128;
    int impl2128_mask = 0;       // experiments: the same kernel for the plain qkv (1) / fc1 (4) / fc2 (8) launches (2 = proj, same as proj_impl)
    // hipGraph replay of launch-bound calls (one prompt / one tile: ~100 dependent kernels of a few us each)
    struct GraphSlot { hipGraphExec_t exec; unsigned long long epoch; char* arena; };
    std::map<std::string, GraphSlot> graphs;
    int use_graphs = 1;
    unsigned long long opt_epoch = 0;   // bumped by keep_set_option / keep_finalize_weights: graphs captured under an older epoch are dropped
    hipStream_t cap_stream = nullptr;
    int dbg_calls = 0;
    int dbg_skip_ln = 0;         // diagnostics (takes effect from the 4th encode_image call, so the buffers hold real data): skip the ViT block LayerNorm launches (results wrong; bounds what fusing them away could gain)
    int lane_min_tiles = 16;     // a lane is only opened for at least this many tiles (32 tiles: 7.09 -> 6.50 ms as 2 x 16; 16 tiles as 2 x 8 loses)
    int lane_skew = 0;           // >0: lane l starts after lane l-1 finished stage `lane_skew` of block 0 (1 qkv .. 5 fc2)
    hipEvent_t ev_skew[4] = {nullptr, nullptr, nullptr, nullptr};
    int lane0_permille = 500;    // share of a 2-lane chunk given to lane 0 (experiments with workgroup-round packing)
    int n_streams = 2;           // concurrent sub-batches inside keep_encode_image (1 = everything on the caller's stream)
    hipStream_t aux[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    // The CLS-row chain (KEEP_MLP_CLS, with KEEP_ATTN_PROJ_CLS in front of it) of a lane on a stream of its own: 7-8 small DEPENDENT launches per block.  In
    // the lane's own stream each of them queues behind whatever persistent GEMM of the other lane holds the CUs, the lane advances one small kernel per big
    // kernel of its neighbour, and the two lanes end up taking turns -- with the chain in all 24 blocks the two-lane step was the single-stream sum (round 6).
    // Forked after the residual gather, joined before the next block: the chain runs under the lane's own LayerNorm-2 / fc1 / fc2.
    // MEASURED NEGATIVE, off: 6 334 against 7 086 tiles/s on one box (tools/ab_options.py --calibrate --arm base --arm cls_side_stream=0, three rotated rounds,
    // profiles/r06_ab_cls_side_stream.txt): the three cross-stream waits per block and lane (144 per step) cost more than the chain's exposure -- a
    // cross-queue dependency is a barrier packet the next persistent GEMM sits behind.  The premise was wrong too: the kernel times of a step add up to
    // 37.8 ms on one stream and the two-lane step takes 37.7 -- the lanes already pack the GPU back to back; what the chains cost is their own latency.
    int cls_side_stream = 0;
    int cls_chain_early = 1;     // 2: a chain that starts with the CLS-row proj is issued even before the plain proj; 1: the chain (own stream) is issued before the block's LayerNorm-2 and scattered behind its fc2; 0: all of it behind fc2, last reduce writes the rows back
    hipStream_t aux_cls[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_cls[4][3] = {};
    float* cls_splitk[4] = {nullptr, nullptr, nullptr, nullptr};     // the chain's own K-slice scratch (the lane's is in use by its main stream)

    // workspace arena
    char* arena = nullptr;
    size_t arena_bytes = 0;
    int* err_flag = nullptr;     // device int, sticky: bit 0 out-of-range token ids, bit 1 non-finite output features (fp16 range exceeded)

    // profiling
    int prof_mode = 0;           // 0 off, 1 the tags of prof_mask, 2 all
    unsigned long long prof_mask = 0;
    struct Rec { hipEvent_t a, b; int tag; };
    std::vector<Rec> recs;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    double prof_ms[T_COUNT] = {0};
    int64_t prof_n[T_COUNT] = {0};
    double prof_flops[T_COUNT] = {0};      // executed FLOPs (2*M*N*K) of the profiled GEMM launches

    int fail(int code, const char* fmt, ...) {
        char buf[1024];
        va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        err = buf;
        return code;
    }
    // Where the 11 bits of an fp16 operand are not enough (tools/precision_study.py: block 0 alone is 41 % of the cosine error
    // variance, blocks 0-1 52 %, and outside them the MLP GEMMs carry > 80 %):
    //   attention side (qkv, q/k/v storage, softmax probabilities, proj) of block i: split product or plain
    //   MLP (fc1, fc2) of block i: 0 plain | 1 split (three fp16 passes) | 2 compensated (fp16 pass + two MX-fp4 correction terms)
    int plan_attn(int i) const { return (precision == KEEP_PREC_COMP && i >= 0 && i < MAX_BLOCKS) ? attn_mode[i] : KEEP_ATTN_PLAIN; }
    // lanes too small for the 256x256 kernel take split products wherever a compensated one is asked for (they run on the small-M / K-sliced kernels)
    bool vit_attn_split(int i, int lane_tiles = 1 << 20) const {
        if (precision == KEEP_PREC_STRICT || i < strict_blocks) return true;
        const int a = plan_attn(i);
        return a == KEEP_ATTN_SPLIT || a == KEEP_ATTN_SPLIT_COMPQKV || ((a == KEEP_ATTN_COMPQKV || a == KEEP_ATTN_COMPQKV_PROJ_CLS) && !(lane_tiles >= comp_min_tiles && vit_has_q));
    }
    bool vit_qkv_comp(int i, int lane_tiles) const {
        if (precision != KEEP_PREC_COMP || i < strict_blocks || !(lane_tiles >= comp_min_tiles && vit_has_q)) return false;
        const int a = plan_attn(i);
        return a == KEEP_ATTN_SPLIT_COMPQKV || a == KEEP_ATTN_COMPQKV || a == KEEP_ATTN_COMPQKV_PROJ_CLS;
    }
    // fc1 / fc2 of block i: 0 plain | 1 split (three fp16 passes) | 2 compensated (both MX-fp4 terms) | 3 compensated, W_lo term only | 4 plain + CLS rows split
    // (lane_tiles == 0: the last block's CLS-rows-only tail, which is the "CLS rows as split products" half on its own)
    int vit_mlp_mode(int i, int lane_tiles) const {
        if (precision == KEEP_PREC_STRICT || i < strict_blocks) return KEEP_MLP_SPLIT;
        if (precision != KEEP_PREC_COMP || i < 0 || i >= MAX_BLOCKS) return KEEP_MLP_PLAIN;
        const int m = mlp_mode[i];
        if (m == KEEP_MLP_COMP || m == KEEP_MLP_COMP_W) return (lane_tiles >= comp_min_tiles && vit_has_q) ? m : KEEP_MLP_SPLIT;
        if (m == KEEP_MLP_CLS) return lane_tiles == 0 ? KEEP_MLP_SPLIT : KEEP_MLP_CLS;
        return m;
    }
    // the text tower is 1 % of a slide's work: in the compensated mode it simply runs split products throughout, at every length BertModel accepts
    // (T <= 512 = max_position_embeddings; above 256 keys the split attention runs as two key windows of <= 256, merged like an online softmax)
    bool txt_split(int l, int T) const { (void)T; return precision == KEEP_PREC_STRICT || l < strict_blocks || precision == KEEP_PREC_COMP; }
    bool any_split() const { return precision != KEEP_PREC_FP16 || strict_blocks > 0; }
    bool vit_has_q = false;      // every fc1 / fc2 weight has its fp4 side planes (dims % 128 == 0)
    bool any_comp() const {
        if (precision != KEEP_PREC_COMP || !vit_has_q) return false;
        for (int i = 0; i < MAX_BLOCKS && i < (vit_depth ? vit_depth : MAX_BLOCKS); ++i)
            if (mlp_mode[i] == KEEP_MLP_COMP || mlp_mode[i] == KEEP_MLP_COMP_W || attn_mode[i] == KEEP_ATTN_SPLIT_COMPQKV || attn_mode[i] == KEEP_ATTN_COMPQKV || attn_mode[i] == KEEP_ATTN_COMPQKV_PROJ_CLS) return true;
        return false;
    }

    bool prof_on(int tag) const { return prof_mode == 2 || (prof_mode == 1 && ((prof_mask >> tag) & 1ull)); }
    void prof_add_flops(int tag, double f) { if (prof_on(tag)) prof_flops[tag] += f; }
    void prof_begin(int tag, hipStream_t s) {
        if (!prof_on(tag)) return;
        Rec r; r.tag = tag;
        if (!pool.empty()) { r.a = pool.back().first; r.b = pool.back().second; pool.pop_back(); }
        else { hipEventCreate(&r.a); hipEventCreate(&r.b); }
        hipEventRecord(r.a, s);
        recs.push_back(r);
    }
    void prof_end(int tag, hipStream_t s) {
        if (!prof_on(tag)) return;
        hipEventRecord(recs.back().b, s);
    }
    void prof_collect() {
        for (auto& r : recs) {
            hipEventSynchronize(r.b);
            float ms = 0.f;
            hipEventElapsedTime(&ms, r.a, r.b);
            prof_ms[r.tag] += ms; prof_n[r.tag] += 1;
            pool.push_back({r.a, r.b});
        }
        recs.clear();
    }
};

namespace {

struct Scope {     // RAII profile bracket
    keep_handle* h; int tag; hipStream_t s;
    Scope(keep_handle* h_, int t, hipStream_t s_) : h(h_), tag(t), s(s_) { h->prof_begin(tag, s); }
    ~Scope() { h->prof_end(tag, s); }
};

bool starts_with(const std::string& s, const char* p) { return s.rfind(p, 0) == 0; }
bool ends_with(const std::string& s, const char* p) {
    const size_t n = strlen(p);
    return s.size() >= n && s.compare(s.size() - n, n, p) == 0;
}

// GEMM weights are stored as fp16 planes only; everything else keeps fp32.
bool is_gemm_weight(const std::string& k) {
    if (k == "visual.patch_embed.proj.weight") return true;
    if (starts_with(k, "visual.blocks.") && ends_with(k, ".weight") &&
        (k.find(".attn.qkv.") != std::string::npos || k.find(".attn.proj.") != std::string::npos ||
         k.find(".mlp.fc1.") != std::string::npos || k.find(".mlp.fc2.") != std::string::npos)) return true;
    if (starts_with(k, "text.encoder.layer.") && ends_with(k, ".weight") && k.find("LayerNorm") == std::string::npos) return true;
    return false;
}

int64_t numel_of(const std::vector<int64_t>& s) { int64_t n = 1; for (auto d : s) n *= d; return n; }

const WTensor* find(keep_handle* h, const std::string& k) {
    auto it = h->w.find(k);
    return it == h->w.end() ? nullptr : &it->second;
}

bool shape_is(const WTensor* t, std::initializer_list<int64_t> s) {
    if (!t || t->shape.size() != s.size()) return false;
    size_t i = 0;
    for (auto d : s) if (t->shape[i++] != d) return false;
    return true;
}

size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct Carver {
    char* base; size_t off = 0;
    explicit Carver(char* b) : base(b) {}
    template <typename T> T* take(size_t n) { T* p = reinterpret_cast<T*>(base + off); off += align_up(n * sizeof(T)); return p; }
};

struct VitWs { float* splitk; float* resid; f16 *xn_hi, *xn_lo, *qkv_hi, *qkv_lo, *att_hi, *att_lo, *mlp_hi, *mlp_lo, *pat_hi, *pat_lo; float *cls, *h1;
               unsigned char *xn_q, *xn_sc, *mlp_q, *mlp_sc;     // MX-fp4 side planes of the LayerNorm-2 output and of the MLP hidden (compensated mode)
               // compact CLS-row buffers for the last block
               float* c_resid; f16 *c_att_hi, *c_att_lo, *c_xn_hi, *c_xn_lo, *c_mlp_hi, *c_mlp_lo; };
struct TxtWs { float* splitk; float* resid; f16 *xn_hi, *xn_lo, *qkv_hi, *qkv_lo, *att_hi, *att_lo, *mlp_hi, *mlp_lo; float* attn_part; size_t attn_part_bytes; };

size_t vit_ws_bytes(const keep_handle* h, int64_t Bc, bool split) {
    const size_t M = (size_t)Bc * 197, Mp = (size_t)Bc * 196, D = h->vit_D, F = h->vit_F, k = split ? 2 : 1;
    const size_t comp = h->any_comp() ? align_up(keepk::q4_data_bytes(M, D)) + align_up(keepk::q4_scale_bytes(M, D)) +
                                        align_up(keepk::q4_data_bytes(M, F)) + align_up(keepk::q4_scale_bytes(M, F)) : 0;
    return comp + align_up(SKINNY_WS_BYTES) + align_up(M * D * 4) + k * (align_up(blk_elems(M, D) * 2) * 2 + align_up(M * 3 * D * 2) + align_up(blk_elems(M, F) * 2)) + 2 * align_up(blk_elems(Mp, 768) * 2) +
           align_up((size_t)Bc * D * 4) + align_up((size_t)Bc * h->proj_dim * 4) + 4096 +
           align_up((size_t)Bc * D * 4) + 2 * (2 * align_up(blk_elems(Bc, D) * 2) + align_up(blk_elems(Bc, F) * 2));
}
VitWs carve_vit(const keep_handle* h, char* arena, int64_t Bc, bool split) {
    const size_t M = (size_t)Bc * 197, Mp = (size_t)Bc * 196, D = h->vit_D, F = h->vit_F;
    Carver c(arena); VitWs w{};
    w.splitk = c.take<float>(SKINNY_WS_BYTES / 4);
    w.resid = c.take<float>(M * D);
    w.xn_hi = c.take<f16>(blk_elems(M, D));  w.xn_lo = split ? c.take<f16>(blk_elems(M, D)) : nullptr;
    w.qkv_hi = c.take<f16>(M * 3 * D);       w.qkv_lo = split ? c.take<f16>(M * 3 * D) : nullptr;
    w.att_hi = c.take<f16>(blk_elems(M, D)); w.att_lo = split ? c.take<f16>(blk_elems(M, D)) : nullptr;
    w.mlp_hi = c.take<f16>(blk_elems(M, F)); w.mlp_lo = split ? c.take<f16>(blk_elems(M, F)) : nullptr;
    w.pat_hi = c.take<f16>(blk_elems(Mp, 768)); w.pat_lo = c.take<f16>(blk_elems(Mp, 768));
    w.cls = c.take<float>((size_t)Bc * D);
    w.h1 = c.take<float>((size_t)Bc * h->proj_dim);
    w.c_resid = c.take<float>((size_t)Bc * D);
    w.c_att_hi = c.take<f16>(blk_elems(Bc, D)); w.c_att_lo = c.take<f16>(blk_elems(Bc, D));
    w.c_xn_hi = c.take<f16>(blk_elems(Bc, D));  w.c_xn_lo = c.take<f16>(blk_elems(Bc, D));
    w.c_mlp_hi = c.take<f16>(blk_elems(Bc, F)); w.c_mlp_lo = c.take<f16>(blk_elems(Bc, F));
    if (h->any_comp()) {
        w.xn_q = c.take<unsigned char>(keepk::q4_data_bytes(M, D));  w.xn_sc = c.take<unsigned char>(keepk::q4_scale_bytes(M, D));
        w.mlp_q = c.take<unsigned char>(keepk::q4_data_bytes(M, F)); w.mlp_sc = c.take<unsigned char>(keepk::q4_scale_bytes(M, F));
    }
    return w;
}
size_t txt_ws_bytes(const keep_handle* h, int64_t Pc, int64_t T, bool split) {
    const size_t M = (size_t)Pc * T, H = h->bert_H, F = h->bert_F, k = split ? 2 : 1;
    // split attention over more than 256 keys parks a partial state per (prompt, head, query) between its two key windows
    const size_t part = (split && T > 256) ? align_up((size_t)Pc * h->bert_heads * T * ATT_PART_FLOATS * sizeof(float)) : 0;
    return align_up(SKINNY_WS_BYTES) + align_up(M * H * 4) + k * (align_up(blk_elems(M, H) * 2) * 2 + align_up(M * 3 * H * 2) + align_up(blk_elems(M, F) * 2)) + part + 4096;
}
TxtWs carve_txt(const keep_handle* h, char* arena, int64_t Pc, int64_t T, bool split) {
    const size_t M = (size_t)Pc * T, H = h->bert_H, F = h->bert_F;
    Carver c(arena); TxtWs w{};
    w.splitk = c.take<float>(SKINNY_WS_BYTES / 4);
    w.resid = c.take<float>(M * H);
    w.xn_hi = c.take<f16>(blk_elems(M, H));  w.xn_lo = split ? c.take<f16>(blk_elems(M, H)) : nullptr;
    w.qkv_hi = c.take<f16>(M * 3 * H);       w.qkv_lo = split ? c.take<f16>(M * 3 * H) : nullptr;
    w.att_hi = c.take<f16>(blk_elems(M, H)); w.att_lo = split ? c.take<f16>(blk_elems(M, H)) : nullptr;
    w.mlp_hi = c.take<f16>(blk_elems(M, F)); w.mlp_lo = split ? c.take<f16>(blk_elems(M, F)) : nullptr;
    if (split && T > 256) {
        w.attn_part_bytes = (size_t)Pc * h->bert_heads * T * ATT_PART_FLOATS * sizeof(float);
        w.attn_part = c.take<float>(w.attn_part_bytes / sizeof(float));
    }
    return w;
}

int ensure_arena(keep_handle* h, size_t bytes) {
    if (bytes <= h->arena_bytes) return KEEP_OK;
    HIPCHK(h, hipDeviceSynchronize());
    if (h->arena) HIPCHK(h, hipFree(h->arena));
    h->arena = nullptr; h->arena_bytes = 0;
    HIPCHK(h, hipMalloc(&h->arena, bytes));
    h->arena_bytes = bytes;
    return KEEP_OK;
}

int check_launch(keep_handle* h, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return h->fail(KEEP_EHIP, "%s: %s", what, hipGetErrorString(e));
    return KEEP_OK;
}


void drop_graphs(keep_handle* h) {
    for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second.exec);
    h->graphs.clear();
}

// Replays `body` (which must only enqueue work on the stream it is given: no allocation, no synchronisation) as one
// graph launch on `s`; captures it on first use.
template <class F>
int graph_run(keep_handle* h, const std::string& key, hipStream_t s, F&& body) {
    auto it = h->graphs.find(key);
    if (it != h->graphs.end() && (it->second.epoch != h->opt_epoch || it->second.arena != h->arena)) {
        (void)hipGraphExecDestroy(it->second.exec);
        h->graphs.erase(it);
        it = h->graphs.end();
    }
    if (it == h->graphs.end()) {
        if (!h->cap_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
        HIPCHK(h, hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
        const int rc = body(h->cap_stream);
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(h->cap_stream, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        HIPCHK(h, e);
        hipGraphExec_t ex = nullptr;
        const hipError_t ei = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        HIPCHK(h, ei);
        it = h->graphs.emplace(key, keep_handle::GraphSlot{ex, h->opt_epoch, h->arena}).first;
    }
    HIPCHK(h, hipGraphLaunch(it->second.exec, s));
    return KEEP_OK;
}

int run_gemm(keep_handle* h, int tag, GemmParams p, int epi, hipStream_t s, float* splitk) {
    h->prof_add_flops(tag, 2.0 * p.M * (double)p.N * p.K);     // algorithmic FLOPs: extra passes of a split / compensated product are not counted
    p.splitk_ws = splitk; p.splitk_bytes = SKINNY_WS_BYTES;
    return launch_gemm_f16(p, epi, s);
}

// offer the LayerNorm that follows a residual GEMM to the GEMM itself (taken only on the small-M split-K path)
void offer_ln(GemmParams& p, const LnParams& ln) {
    p.ln_gamma = ln.gamma; p.ln_beta = ln.beta; p.ln_eps = ln.eps;
    p.ln_out_hi = ln.out_hi; p.ln_out_lo = ln.out_lo; p.ln_out_f32 = ln.out_f32;
}

GemmParams gemm_params(const keep_handle* h, const f16* a_hi, const f16* a_lo, const WTensor* w, int M, bool split, const float* bias) {
    GemmParams p{};
    p.tune = &h->tune;
    p.a_hi = a_hi; p.a_lo = a_lo; p.w_hi = w->hi; p.w_lo = w->lo;
    p.M = M; p.N = (int)w->shape[0]; p.K = (int)(w->numel / w->shape[0]);
    p.nseg = split ? 3 : 1;
    p.bias = bias;
    p.patches_per_img = 196;
    return p;
}

// ---------------------------------------------------------------------------------------------
// One sub-batch of tiles in flight on one stream ("lane").  keep_encode_image runs up to n_streams lanes
// concurrently and issues their kernels layer-interleaved, so one lane's memory-bound phases (LayerNorm,
// attention staging, GEMM epilogues, partial last rounds of workgroups) overlap the other lane's MFMA phases.
struct VitLane {
    const void* pixels; int pix_dtype; int Bc; float* out; hipStream_t s; VitWs ws; bool cls_compact = false;
    hipEvent_t skew_ev = nullptr; int skew_stage = 0;     // recorded after stage `skew_stage` of block 0 (lane_skew)
    bool xn_ready = false;                                // the previous block's fc2 already wrote this block's LayerNorm-1 output
    hipStream_t cs = nullptr; hipEvent_t ce[3] = {nullptr, nullptr, nullptr}; float* cs_splitk = nullptr;      // side stream of the CLS-row chain (nullable)
    bool c_resid_live = false;                            // ws.c_resid holds the CLS rows of ws.resid as they are NOW (left there by the previous block's CLS-row chain): no gather
};

int vit_begin(keep_handle* h, VitLane& L) {
    const int D = h->vit_D, Bc = L.Bc;
    hipStream_t s = L.s; VitWs& ws = L.ws; const void* pixels = L.pixels; const int pix_dtype = L.pix_dtype;
    // The patch embed is 0.25 % of the FLOPs but its rounding error feeds all 24 blocks: it runs as the hi/lo split product
    // (option "patch_split" = 0, experiments: one fp16 pass -- KEEP_PREC_FP16 and KEEP_PREC_COMP only).
    const bool sp0 = h->patch_split || h->precision == KEEP_PREC_STRICT || h->strict_blocks > 0;
    {
        Scope sc(h, T_VIT_IM2COL, s);
        launch_im2col(pixels, pix_dtype, Bc, ws.pat_hi, sp0 ? ws.pat_lo : nullptr,
                      find(h, "visual.cls_token")->f32, find(h, "visual.pos_embed")->f32, ws.resid, D, s);
    }
    {
        Scope sc(h, T_VIT_PATCH, s);
        GemmParams p = gemm_params(h, ws.pat_hi, ws.pat_lo, find(h, "visual.patch_embed.proj.weight"), Bc * 196, sp0,
                                   find(h, "visual.patch_embed.proj.bias")->f32);
        p.pos = find(h, "visual.pos_embed")->f32;
        p.resid = ws.resid;
        if (run_gemm(h, T_VIT_PATCH, p, EPI_PATCH, s, ws.splitk) < 0) return h->fail(KEEP_EUNSUPPORTED, "patch-embedding GEMM launch failed");
    }
    return KEEP_OK;
}

int vit_layer(keep_handle* h, VitLane& L, int i) {
    const int D = h->vit_D, Bc = L.Bc, M = Bc * 197;
    hipStream_t s = L.s; VitWs& ws = L.ws;
    auto mark = [&](int stage) { if (i == 0 && L.skew_ev && L.skew_stage == stage) (void)hipEventRecord(L.skew_ev, s); };
    const VitBlock& b = h->vblocks[i];
    // mean-input compensation: a PLAIN launch of site k uses the bias with W_lo a_mean folded in; while calibrating, every site's input is summed
    auto site_bias = [&](int site, const float* orig, bool plain) {
        return (plain && h->bias_ready && h->bias_correction && i < (int)h->cal.size() && h->cal[i].bias[site]) ? (const float*)h->cal[i].bias[site] : orig;
    };
    auto capture = [&](int site, const f16* a_hi, int rows, int K) {
        if (!h->capture || i >= (int)h->cal.size() || !h->cal[i].sum[site]) return;
        launch_blk_col_sum(a_hi, rows, K, h->cal[i].sum[site], s);
        h->cal[i].rows[site] += rows;
    };
    // Last block: everything after the attention is per-token and only the CLS token is pooled
    // (global_pool='token'), so its queries / proj / MLP are evaluated for the B CLS rows only.
    // Exact (same arithmetic on the rows that matter); the skipped FLOPs still count as algorithmic work.
    const bool cls_only = (i == h->vit_depth - 1) && h->cls_tail;
    const bool sp = h->vit_attn_split(i, Bc);                          // qkv / attention / proj as split products
    const int mlp = h->vit_mlp_mode(i, cls_only ? 0 : Bc);             // fc1 / fc2: 0 plain, 1 split, 2 compensated (both MX-fp4 terms), 3 compensated (W_lo term only)
    const bool mlp_lo = mlp == KEEP_MLP_SPLIT, mlp_q = mlp == KEEP_MLP_COMP || mlp == KEEP_MLP_COMP_W;
    const int mlp_comp = mlp == KEEP_MLP_COMP_W ? 1 : 2;               // GemmParams.comp of the block's fc1 / fc2
    const bool mlp_cls = mlp == KEEP_MLP_CLS;                          // every row plain, then the CLS rows again as split products (below)
    const bool mlp_plain = mlp == KEEP_MLP_PLAIN || mlp_cls;
    // KEEP_ATTN_PROJ_CLS: every row plain; the attention kernel also writes the CLS rows' output hi + lo (compact), and their proj runs again as a split product
    const bool proj_cls = !sp && !cls_only && (h->plan_attn(i) == KEEP_ATTN_PROJ_CLS || h->plan_attn(i) == KEEP_ATTN_COMPQKV_PROJ_CLS) && i >= h->strict_blocks;
#ifdef KEEP_DIAGNOSTICS
    const bool skip_ln = h->dbg_skip_ln == 1 && h->dbg_calls > 3;
#else
    const bool skip_ln = false;
#endif
    // qkv of a split-attention block in the compensated mode: fp16 pass + MX-fp4 correction terms instead of three fp16 passes (lanes
    // large enough for the 256x256 kernel; LayerNorm-1 then writes the fp4 planes of its output instead of the lo plane)
    const bool qkv_q = h->vit_qkv_comp(i, Bc) && b.qkv->q && ws.xn_q && !L.xn_ready;
    LnParams ln{};
    ln.tune = &h->tune;
    ln.x = ws.resid; ln.x_stride = D; ln.rows = M; ln.D = D; ln.eps = 1e-6f;
    ln.out_hi = ws.xn_hi; ln.out_lo = (sp && !qkv_q) ? ws.xn_lo : nullptr; ln.out_kt = D / 32;
    ln.out_q = qkv_q ? ws.xn_q : nullptr; ln.out_sc = qkv_q ? ws.xn_sc : nullptr;
    if (!L.xn_ready && !skip_ln) {
        Scope sc(h, T_VIT_LN, s);
        ln.gamma = b.n1w; ln.beta = b.n1b;
        if (launch_layernorm(ln, s)) return h->fail(KEEP_EUNSUPPORTED, "layernorm width %d", D);
    }
    L.xn_ready = false;
    // Last block, single-pass lanes: only the CLS row of every image is a query, so the q third of the qkv GEMM is computed for those Bc rows only
    // (exact: the skipped rows' q is never read; the FLOPs it saves are not counted as done).  K and V still need every token.
    const bool kv_only = cls_only && h->cls_qkv && !sp && !qkv_q && Bc >= 32 && !L.xn_ready && D % 256 == 0;
    {
        const int tag = (sp || qkv_q) ? T_VIT_QKV_X : T_VIT_QKV;
        Scope sc(h, tag, s);
        capture(0, ws.xn_hi, M, D);
        GemmParams p = gemm_params(h, ws.xn_hi, ws.xn_lo, b.qkv, M, sp && !qkv_q, site_bias(0, b.qkv_b, !sp && !qkv_q));
        if (!sp && !qkv_q && (h->impl2128_mask & 1)) p.impl_hint = 2128;
        if (kv_only) {          // weight rows D .. 3D-1 (n-tiles D/256 ..), written into columns D .. 3D-1 of the token-major qkv buffer
            p.N = 2 * D; p.w_hi += (int64_t)(D / 256) * (D / 32) * 8192; p.bias += D; p.out_ld = 3 * D; p.out_col0 = D;
        }
        p.out_hi = ws.qkv_hi; p.out_lo = sp ? ws.qkv_lo : nullptr;
        if (qkv_q) { p.comp = 2; p.a_q = ws.xn_q; p.a_sc = ws.xn_sc; p.w_q = b.qkv->q; p.w_sc = b.qkv->sc; }
        if (run_gemm(h, tag, p, EPI_F16, s, ws.splitk) < 0) return h->fail(KEEP_EUNSUPPORTED, "qkv GEMM launch failed");
    }
    if (kv_only) {              // q of the CLS rows: gather their LayerNorm-1 rows, [Bc, D] x W_q^T on the small-M kernel, into a compact buffer (free until the MLP)
        Scope sc(h, T_VIT_TAIL, s);
        launch_gather_rows_blk(ws.xn_hi, 197, ws.c_xn_hi, Bc, D, s);
        GemmParams p = gemm_params(h, ws.c_xn_hi, nullptr, b.qkv, Bc, false, site_bias(0, b.qkv_b, true));
        p.N = D; p.out_hi = ws.c_mlp_hi;
        if (run_gemm(h, T_VIT_TAIL, p, EPI_F16, s, ws.splitk) < 0) return h->fail(KEEP_EUNSUPPORTED, "CLS-query GEMM launch failed");
    }
    mark(1);
    {
        Scope sc(h, sp ? T_VIT_ATTN_X : T_VIT_ATTN, s);
        AttnParams a{};
        a.tune = &h->tune;
        a.qkv_hi = ws.qkv_hi; a.qkv_lo = ws.qkv_lo; a.out_hi = ws.att_hi; a.out_lo = sp ? ws.att_lo : nullptr;
        a.mask = nullptr; a.batch = Bc; a.ntok = 197; a.heads = h->vit_heads; a.split = sp; a.scale = 0.125f; a.out_kt = D / 32;
        a.q_rows = cls_only ? 1 : 0;
        if (kv_only) { a.q_hi = ws.c_mlp_hi; a.q_ld = D; }
        if (proj_cls) { a.cls_hi = ws.c_att_hi; a.cls_lo = ws.c_att_lo; }
#ifdef KEEP_DIAGNOSTICS
        if (!(h->dbg_skip_ln == 2 && h->dbg_calls > 3))      // dbg_skip_ln = 2: skip the attention launches instead (bounds what a faster attention could gain)
#endif
        if (launch_attention(a, s)) return h->fail(KEEP_EUNSUPPORTED, "attention launch failed");
    }
    mark(2);
    const int Mr = cls_only ? Bc : M;
    float* resid = cls_only ? ws.c_resid : ws.resid;
    const f16 *att_hi = ws.att_hi, *att_lo = ws.att_lo;
    f16 *xn_hi = ws.xn_hi, *xn_lo = ws.xn_lo, *mlp_hi = ws.mlp_hi, *mlp_lo_p = ws.mlp_lo;
    if (cls_only) {
        Scope sc(h, T_VIT_HEAD, s);
        launch_gather_rows_f32(ws.resid, (int64_t)197 * D, ws.c_resid, Bc, D, s);
        launch_gather_rows_blk(ws.att_hi, 197, ws.c_att_hi, Bc, D, s);
        if (sp) launch_gather_rows_blk(ws.att_lo, 197, ws.c_att_lo, Bc, D, s);
        att_hi = ws.c_att_hi; att_lo = ws.c_att_lo; xn_hi = ws.c_xn_hi; xn_lo = ws.c_xn_lo; mlp_hi = ws.c_mlp_hi; mlp_lo_p = ws.c_mlp_lo;
        L.cls_compact = true;
    }
    ln.x = resid; ln.rows = Mr; ln.out_hi = xn_hi; ln.out_lo = mlp_lo ? xn_lo : nullptr;
    ln.out_q = mlp_q ? ws.xn_q : nullptr; ln.out_sc = mlp_q ? ws.xn_sc : nullptr; ln.out_q_hi_only = mlp == KEEP_MLP_COMP_W;
    ln.gamma = b.n2w; ln.beta = b.n2b;
    if (proj_cls && !L.c_resid_live) {      // the CLS rows' residual as it enters proj (the plain proj below updates these rows too; the split result replaces that)
        Scope sc(h, T_VIT_TAIL, s);
        launch_gather_rows_f32(ws.resid, (int64_t)197 * D, ws.c_resid, Bc, D, s);
    }
    int did = 0;
    auto main_proj = [&]() -> int {
        const int tag = cls_only ? T_VIT_TAIL : sp ? T_VIT_PROJ_X : T_VIT_PROJ;
        Scope sc(h, tag, s);
        capture(1, att_hi, Mr, D);
        GemmParams p = gemm_params(h, att_hi, att_lo, b.proj, Mr, sp, site_bias(1, b.proj_b, !sp));
        p.ls = b.ls1; p.resid = resid; p.impl_hint = (h->impl2128_mask & 2) ? 2128 : h->proj_impl;
        if (!mlp_q) offer_ln(p, ln);                 // the fused LayerNorm of the small-M path does not write fp4 planes
        did = run_gemm(h, tag, p, EPI_RESID_LS, s, ws.splitk);
        if (did < 0) return h->fail(KEEP_EUNSUPPORTED, "proj GEMM launch failed");
        return KEEP_OK;
    };
    // cls_chain_early = 2: a chain that starts with the CLS-row proj needs nothing of the plain proj (its residual is the one in FRONT of it): issued before it
    const bool chain_first = proj_cls && mlp_cls && h->cls_chain_early == 2 && L.cs == nullptr;
    if (!chain_first) { const int rcp = main_proj(); if (rcp) return rcp; }
    mark(3);
    bool cls_ln_done = false;   // LayerNorm-2 of the compact CLS rows already written (hi + lo) by the CLS-row proj's epilogue
    // The CLS-row chain of this block (KEEP_ATTN_PROJ_CLS and / or KEEP_MLP_CLS) on the compact [Bc, D] rows.  Its stream: the lane's own -- then the whole chain is
    // issued BEHIND the plain fc2, and its last reduce writes the rows straight back into the token stream -- or the lane's side stream ("cls_side_stream", off:
    // measured negative), forked here and joined by a scatter behind the plain fc2.
    const bool side = mlp_cls && L.cs != nullptr;
    hipStream_t cs = side ? L.cs : s;
    float* c_splitk = side ? L.cs_splitk : ws.splitk;
    if (mlp_cls && !proj_cls) { // the CLS rows' residual as it enters the MLP, i.e. BEHIND this block's proj (a live compact copy is the residual in front of it): always gathered
        Scope sc(h, T_VIT_TAIL, s);
        launch_gather_rows_f32(ws.resid, (int64_t)197 * D, ws.c_resid, Bc, D, s);
    }
    auto cls_proj = [&]() -> int {
        // [Bc, D] x W_proj^T as a split product on the small-M kernels: the CLS rows' attention output from the fp32 accumulators (hi + lo) against W hi + lo,
        // + LayerScale + the residual.  With KEEP_MLP_CLS in the same block the chain continues on the compact rows (its LayerNorm-2 is fused into this GEMM's reduce)
        Scope sc(h, T_VIT_TAIL, cs);
        GemmParams r = gemm_params(h, ws.c_att_hi, ws.c_att_lo, b.proj, Bc, true, b.proj_b);
        r.ls = b.ls1; r.resid = ws.c_resid;
        LnParams cl{};
        if (mlp_cls) {
            cl.tune = &h->tune;
            cl.x = ws.c_resid; cl.x_stride = D; cl.rows = Bc; cl.D = D; cl.eps = 1e-6f; cl.gamma = b.n2w; cl.beta = b.n2b;
            cl.out_hi = ws.c_xn_hi; cl.out_lo = ws.c_xn_lo; cl.out_kt = D / 32;
            offer_ln(r, cl);
        }
        const int rc = run_gemm(h, T_VIT_TAIL, r, EPI_RESID_LS, cs, c_splitk);
        if (rc < 0) return h->fail(KEEP_EUNSUPPORTED, "CLS-row proj GEMM launch failed");
        cls_ln_done = mlp_cls && (rc & GEMM_DID_LN);
        return KEEP_OK;
    };
    auto cls_mlp = [&](bool write_back) -> int {
        // LayerNorm-2 -> fc1 + GELU -> fc2 + LayerScale + residual as split products on the compact rows; `write_back`: the last reduce also stores the rows into the
        // token stream (they replace what the plain fc2 wrote there: the caller orders this behind it).  0.5 % of the rows; the feature is pooled from them.
        Scope sc(h, T_VIT_TAIL, cs);
        LnParams cl{};
        cl.tune = &h->tune;
        cl.x = ws.c_resid; cl.x_stride = D; cl.rows = Bc; cl.D = D; cl.eps = 1e-6f; cl.gamma = b.n2w; cl.beta = b.n2b;
        cl.out_hi = ws.c_xn_hi; cl.out_lo = ws.c_xn_lo; cl.out_kt = D / 32;
        if (!cls_ln_done && launch_layernorm(cl, cs)) return h->fail(KEEP_EUNSUPPORTED, "layernorm width %d", D);
        GemmParams p = gemm_params(h, ws.c_xn_hi, ws.c_xn_lo, b.fc1, Bc, true, b.fc1_b);
        p.out_hi = ws.c_mlp_hi; p.out_lo = ws.c_mlp_lo; p.out_kt = h->vit_F / 32;
        if (run_gemm(h, T_VIT_TAIL, p, EPI_GELU_F16, cs, c_splitk) < 0) return h->fail(KEEP_EUNSUPPORTED, "CLS-row fc1 GEMM launch failed");
        GemmParams r = gemm_params(h, ws.c_mlp_hi, ws.c_mlp_lo, b.fc2, Bc, true, b.fc2_b);
        r.ls = b.ls2; r.resid = ws.c_resid;
        if (write_back) { r.resid_copy = ws.resid; r.resid_copy_ld = (int64_t)197 * D; }
        const int rc = run_gemm(h, T_VIT_TAIL, r, EPI_RESID_LS, cs, c_splitk);
        if (rc < 0) return h->fail(KEEP_EUNSUPPORTED, "CLS-row fc2 GEMM launch failed");
        if (write_back && (rc & GEMM_NO_RESID_COPY)) launch_scatter_rows_f32(ws.c_resid, ws.resid, (int64_t)197 * D, Bc, D, cs);      // (kernel-selection experiments only)
        return KEEP_OK;
    };
    const bool early = mlp_cls && !side && h->cls_chain_early;      // in the lane's own stream, but issued HERE (before LayerNorm-2; = 2: even before the plain proj) and scattered behind the plain fc2
    if (side || early) {        // fork: everything the chain reads (the gathered residual, the CLS rows' attention output) is queued on the lane's stream before this point
        if (side && (hipEventRecord(L.ce[0], s) != hipSuccess || hipStreamWaitEvent(cs, L.ce[0], 0) != hipSuccess)) return h->fail(KEEP_EHIP, "CLS-row chain: fork failed");
        int rc2 = proj_cls ? cls_proj() : KEEP_OK;
        if (!rc2) rc2 = cls_mlp(false);
        if (rc2) return rc2;
        if (chain_first && (rc2 = main_proj())) return rc2;
    } else if (proj_cls && !mlp_cls) {      // no CLS-row MLP behind it: the rows go back before LayerNorm-2 reads them
        const int rc2 = cls_proj();
        if (rc2) return rc2;
        Scope sc(h, T_VIT_TAIL, s);
        launch_scatter_rows_f32(ws.c_resid, ws.resid, (int64_t)197 * D, Bc, D, s);
    }
    if (!(did & GEMM_DID_LN) && !skip_ln) {
        Scope sc(h, T_VIT_LN, s);
        if (launch_layernorm(ln, s)) return h->fail(KEEP_EUNSUPPORTED, "layernorm width %d", D);
    }
    {
        const int tag = cls_only ? T_VIT_TAIL : !mlp_plain ? T_VIT_FC1_X : T_VIT_FC1;
        Scope sc(h, tag, s);
        capture(2, xn_hi, Mr, D);
        GemmParams p = gemm_params(h, xn_hi, xn_lo, b.fc1, Mr, mlp_lo, site_bias(2, b.fc1_b, mlp_plain));
        p.out_hi = mlp_hi; p.out_lo = mlp_lo ? mlp_lo_p : nullptr; p.out_kt = h->vit_F / 32;
        if (mlp_plain && (h->impl2128_mask & 4)) p.impl_hint = 2128;
        if (mlp_q) {
            p.comp = mlp_comp; p.a_q = ws.xn_q; p.a_sc = ws.xn_sc; p.w_q = b.fc1->q; p.w_sc = b.fc1->sc;
            p.out_q = ws.mlp_q; p.out_sc = ws.mlp_sc;
        }
        if (run_gemm(h, tag, p, EPI_GELU_F16, s, ws.splitk) < 0) return h->fail(KEEP_EUNSUPPORTED, "fc1 GEMM launch failed");
    }
    mark(4);
    {
        const int tag = cls_only ? T_VIT_TAIL : !mlp_plain ? T_VIT_FC2_X : T_VIT_FC2;
        Scope sc(h, tag, s);
        capture(3, mlp_hi, Mr, h->vit_F);
        GemmParams p = gemm_params(h, mlp_hi, mlp_lo_p, b.fc2, Mr, mlp_lo, site_bias(3, b.fc2_b, mlp_plain));
        p.ls = b.ls2; p.resid = resid;
        if (mlp_plain && (h->impl2128_mask & 8)) p.impl_hint = 2128;
        if (mlp_q) { p.comp = mlp_comp; p.a_q = ws.mlp_q; p.a_sc = ws.mlp_sc; p.w_q = b.fc2->q; p.w_sc = b.fc2->sc; }
        if (i + 1 < h->vit_depth && !cls_only && !mlp_cls) {        // next block's LayerNorm-1 reads exactly the rows written here (not when CLS rows are still to be replaced)
            const VitBlock& nb = h->vblocks[i + 1];
            ln.x = ws.resid; ln.rows = M; ln.out_hi = ws.xn_hi; ln.out_lo = h->vit_attn_split(i + 1, Bc) ? ws.xn_lo : nullptr;
            ln.out_q = nullptr; ln.out_sc = nullptr; ln.out_q_hi_only = 0;
            ln.gamma = nb.n1w; ln.beta = nb.n1b;
            offer_ln(p, ln);
        }
        const int rc = run_gemm(h, tag, p, EPI_RESID_LS, s, ws.splitk);
        if (rc < 0) return h->fail(KEEP_EUNSUPPORTED, "fc2 GEMM launch failed");
        L.xn_ready = (rc & GEMM_DID_LN) != 0;
    }
    if (mlp_cls && side) {      // the join: the chain's rows replace what the plain fc2 wrote (ordered after it), and the lane goes on behind the scatter
        Scope sc(h, T_VIT_TAIL, cs);
        if (hipEventRecord(L.ce[1], s) != hipSuccess || hipStreamWaitEvent(cs, L.ce[1], 0) != hipSuccess) return h->fail(KEEP_EHIP, "CLS-row chain: join failed");
        launch_scatter_rows_f32(ws.c_resid, ws.resid, (int64_t)197 * D, Bc, D, cs);
        if (hipEventRecord(L.ce[2], cs) != hipSuccess || hipStreamWaitEvent(s, L.ce[2], 0) != hipSuccess) return h->fail(KEEP_EHIP, "CLS-row chain: join failed");
    } else if (early) {
        Scope sc(h, T_VIT_TAIL, s);
        launch_scatter_rows_f32(ws.c_resid, ws.resid, (int64_t)197 * D, Bc, D, s);
    } else if (mlp_cls) {       // in the lane's own stream: the whole chain behind the plain fc2, the rows written back by its last reduce
        int rc2 = proj_cls ? cls_proj() : KEEP_OK;
        if (!rc2) rc2 = cls_mlp(true);
        if (rc2) return rc2;
    }
    L.c_resid_live = mlp_cls;   // (any other block's proj / fc2 moved the CLS rows of the token stream on without the compact copy)
    mark(5);
    return KEEP_OK;
}

int vit_end(keep_handle* h, VitLane& L) {
    const int D = h->vit_D, Bc = L.Bc;
    hipStream_t s = L.s; VitWs& ws = L.ws; float* out = L.out;
    {
        // final LayerNorm is per-token, global_pool='token' reads row 0 only -> normalise CLS rows only
        Scope sc(h, T_VIT_HEAD, s);
        LnParams ln{};
        ln.tune = &h->tune;
        ln.x = L.cls_compact ? ws.c_resid : ws.resid; ln.x_stride = L.cls_compact ? (int64_t)D : (int64_t)197 * D;
        ln.rows = Bc; ln.D = D; ln.eps = 1e-6f;
        ln.gamma = find(h, "visual.norm.weight")->f32; ln.beta = find(h, "visual.norm.bias")->f32;
        ln.out_f32 = ws.cls; ln.out_f32_stride = D;
        launch_layernorm(ln, s);
        const WTensor* w0 = find(h, "visual_head.0.weight");
        const WTensor* w2 = find(h, "visual_head.2.weight");
        SgemmParams g{};
        g.tune = &h->tune;
        g.a = ws.cls; g.lda = D; g.b = w0->f32; g.ldb = D; g.out = ws.h1; g.ldo = h->proj_dim;
        g.bias = find(h, "visual_head.0.bias")->f32; g.M = Bc; g.N = h->proj_dim; g.K = D; g.scale = 1.f; g.act = ACT_GELU;
        if (launch_sgemm_f32(g, s)) return h->fail(KEEP_EUNSUPPORTED, "visual_head.0 shape");
        g.a = ws.h1; g.lda = h->proj_dim; g.b = w2->f32; g.ldb = h->proj_dim; g.out = out; g.ldo = h->proj_dim;
        g.bias = find(h, "visual_head.2.bias")->f32; g.K = h->proj_dim; g.act = ACT_NONE;
        if (launch_sgemm_f32(g, s)) return h->fail(KEEP_EUNSUPPORTED, "visual_head.2 shape");
        launch_l2norm_rows(out, Bc, h->proj_dim, 1e-12f, s, h->err_flag);
    }
    return check_launch(h, "encode_image");
}

int txt_chunk(keep_handle* h, const int64_t* ids, const int64_t* types, const int64_t* mask, int Pc, int T,
              float* out, hipStream_t s) {
    const bool any_split = h->any_split();
    const int H = h->bert_H, M = Pc * T;
    TxtWs ws = carve_txt(h, h->arena, Pc, T, any_split);
    {
        Scope sc(h, T_TXT_EMBED, s);
        launch_bert_embed_ln(ids, types, find(h, "text.embeddings.word_embeddings.weight")->f32,
                             find(h, "text.embeddings.position_embeddings.weight")->f32,
                             find(h, "text.embeddings.token_type_embeddings.weight")->f32,
                             find(h, "text.embeddings.LayerNorm.weight")->f32,
                             find(h, "text.embeddings.LayerNorm.bias")->f32, 1e-12f, Pc, T, H, h->bert_vocab,
                             h->bert_types, ws.resid, ws.xn_hi, any_split ? ws.xn_lo : nullptr, h->err_flag, s);
    }
    for (int l = 0; l < h->bert_layers; ++l) {
        const BertLayer& b = h->blayers[l];
        const bool sp = h->txt_split(l, T);
        const bool sp_next = (l + 1 < h->bert_layers) && h->txt_split(l + 1, T);
        {
            Scope sc(h, T_TXT_QKV, s);
            GemmParams p = gemm_params(h, ws.xn_hi, ws.xn_lo, &b.qkv, M, sp, b.qkv_b);
            p.out_hi = ws.qkv_hi; p.out_lo = sp ? ws.qkv_lo : nullptr;
            if (run_gemm(h, T_TXT_QKV, p, EPI_F16, s, ws.splitk) < 0) return h->fail(KEEP_EUNSUPPORTED, "text qkv GEMM launch failed");
        }
        {
            Scope sc(h, T_TXT_ATTN, s);
            AttnParams a{};
            a.tune = &h->tune;
            a.qkv_hi = ws.qkv_hi; a.qkv_lo = ws.qkv_lo; a.out_hi = ws.att_hi; a.out_lo = sp ? ws.att_lo : nullptr;
            a.mask = mask; a.batch = Pc; a.ntok = T; a.heads = h->bert_heads; a.split = sp; a.scale = 0.125f; a.out_kt = H / 32;
            a.part_ws = ws.attn_part; a.part_bytes = ws.attn_part_bytes;
            if (launch_attention(a, s)) return h->fail(KEEP_EUNSUPPORTED, "sequence length %d unsupported (max 512)", T);
        }
        LnParams ln{};
        ln.tune = &h->tune;
        ln.x = ws.resid; ln.x_stride = H; ln.rows = M; ln.D = H; ln.eps = 1e-12f;
        ln.out_f32 = ws.resid; ln.out_f32_stride = H; ln.out_hi = ws.xn_hi; ln.out_kt = H / 32;
        ln.gamma = b.ln1w; ln.beta = b.ln1b; ln.out_lo = sp ? ws.xn_lo : nullptr;
        int did;
        {
            Scope sc(h, T_TXT_OUT, s);
            GemmParams p = gemm_params(h, ws.att_hi, ws.att_lo, b.o, M, sp, b.o_b);
            p.resid = ws.resid; p.out_f32 = ws.resid;
            offer_ln(p, ln);
            did = run_gemm(h, T_TXT_OUT, p, EPI_RESID_F32, s, ws.splitk);
            if (did < 0) return h->fail(KEEP_EUNSUPPORTED, "text attention-output GEMM launch failed");
        }
        if (!(did & GEMM_DID_LN)) {
            Scope sc(h, T_TXT_LN, s);
            if (launch_layernorm(ln, s)) return h->fail(KEEP_EUNSUPPORTED, "layernorm width %d", H);
        }
        {
            Scope sc(h, T_TXT_FFN1, s);
            GemmParams p = gemm_params(h, ws.xn_hi, ws.xn_lo, b.i, M, sp, b.i_b);
            p.out_hi = ws.mlp_hi; p.out_lo = sp ? ws.mlp_lo : nullptr; p.out_kt = h->bert_F / 32;
            if (run_gemm(h, T_TXT_FFN1, p, EPI_GELU_F16, s, ws.splitk) < 0) return h->fail(KEEP_EUNSUPPORTED, "text FFN GEMM launch failed");
        }
        ln.gamma = b.ln2w; ln.beta = b.ln2b; ln.out_lo = sp_next ? ws.xn_lo : nullptr;
        {
            Scope sc(h, T_TXT_FFN2, s);
            GemmParams p = gemm_params(h, ws.mlp_hi, ws.mlp_lo, b.d, M, sp, b.d_b);
            p.resid = ws.resid; p.out_f32 = ws.resid;
            offer_ln(p, ln);
            did = run_gemm(h, T_TXT_FFN2, p, EPI_RESID_F32, s, ws.splitk);
            if (did < 0) return h->fail(KEEP_EUNSUPPORTED, "text FFN GEMM launch failed");
        }
        if (!(did & GEMM_DID_LN)) {
            Scope sc(h, T_TXT_LN, s);
            launch_layernorm(ln, s);
        }
    }
    {
        Scope sc(h, T_TXT_POOL, s);
        SgemmParams g{};
        g.tune = &h->tune;
        g.a = ws.resid; g.lda = (int64_t)T * H;            // row p*T: the [CLS] token of prompt p
        g.b = find(h, "text.pooler.dense.weight")->f32; g.ldb = H; g.out = out; g.ldo = H;
        g.bias = find(h, "text.pooler.dense.bias")->f32; g.M = Pc; g.N = H; g.K = H; g.scale = 1.f; g.act = ACT_TANH;
        if (launch_sgemm_f32(g, s)) return h->fail(KEEP_EUNSUPPORTED, "pooler shape");
        launch_l2norm_rows(out, Pc, H, 1e-12f, s, h->err_flag);
    }
    return check_launch(h, "encode_text");
}

// store one state_dict entry
int store_tensor(keep_handle* h, const std::string& key, const float* dev, const std::vector<int64_t>& shape) {
    WTensor t; t.shape = shape; t.numel = numel_of(shape);
    if (t.numel <= 0) return h->fail(KEEP_EINVAL, "%s: empty tensor", key.c_str());
    if (is_gemm_weight(key)) {
        // fp16 hi/lo planes in blk layout; rows (out features) must fill whole 256-row tiles
        const int64_t n = t.shape[0], k = t.numel / t.shape[0];
        if (n % 256 || k % 32) return h->fail(KEEP_EUNSUPPORTED, "%s: [%lld,%lld] is not tileable (rows %% 256, cols %% 32)", key.c_str(), (long long)n, (long long)k);
        // The GEMM operand planes are fp16: 11 significant bits between 6.1e-5 and 65504, fewer below (subnormals), none above.  A weight whose
        // entries sit above that window cannot be represented at all: refused.  (Below it: see the warning further down.)
        {
            float host[2] = {0.f, 0.f};
            HIPCHK(h, hipMemsetAsync(h->err_flag + 2, 0, 2 * sizeof(float), nullptr));
            launch_weight_stats(dev, t.numel, reinterpret_cast<float*>(h->err_flag + 2), nullptr);
            HIPCHK(h, hipMemcpy(host, h->err_flag + 2, sizeof host, hipMemcpyDeviceToHost));
            const double rms = sqrt((double)host[1] / (double)t.numel);
            if (!(host[0] <= 6.0e4f)) return h->fail(KEEP_EUNSUPPORTED, "%s: max |w| = %g does not fit the fp16 operand planes (65504) or is not finite", key.c_str(), (double)host[0]);
            // A weight far below fp16's normal range (a projection whose magnitude lives in its LayerScale, a pruned or dead layer) still loads, as it
            // does in the reference.  proj / fc2 of the image tower are pre-scaled by a power of two into the window -- exact: their epilogue is
            // ls * (acc + bias), and finalize_vit hands it ls / 2^k and bias * 2^k -- any other weight keeps its entries (they fall into fp16
            // subnormals and lose RELATIVE precision; what such a layer adds to the stream is as small as the layer) and the caller is told.
            if (rms > 0.0 && rms < 2.5e-4) {
                const bool foldable = starts_with(key, "visual.blocks.") && (key.find(".attn.proj.weight") != std::string::npos || key.find(".mlp.fc2.weight") != std::string::npos);
                char buf[512];
                if (foldable && t.numel < (1ll << 31)) {
                    float k2 = exp2f(roundf(log2f(0.02f / (float)rms)));
                    while (host[0] * k2 > 3.0e4f) k2 *= 0.5f;
                    t.prescale = k2;
                    snprintf(buf, sizeof buf, "%s: rms %g is below fp16's normal range; stored as 2^%d * W with LayerScale / bias adjusted (exact)", key.c_str(), rms, (int)log2f(k2));
                } else {
                    snprintf(buf, sizeof buf, "%s: rms %g is below what the fp16 operand planes resolve with 11 bits (entries fall into fp16 subnormals): "
                                              "this layer's products carry fewer significant bits than the error budget assumes", key.c_str(), rms);
                }
                h->load_warnings += (h->load_warnings.empty() ? "" : "\n") + std::string(buf);
            }
        }
        float* scaled = nullptr;
        if (t.prescale != 1.f) {
            HIPCHK(h, hipMalloc(&scaled, t.numel * sizeof(float)));
            launch_scale_vec(dev, (int)t.numel, t.prescale, scaled, nullptr);
            dev = scaled;
        }
        HIPCHK(h, hipMalloc(&t.hi, t.numel * sizeof(f16)));
        HIPCHK(h, hipMalloc(&t.lo, t.numel * sizeof(f16)));
        // the MLP weights of the image tower also get the MX-fp4 side planes of the compensated product (quant4.h)
        if ((key.find(".mlp.fc") != std::string::npos || key.find(".attn.qkv.") != std::string::npos) && starts_with(key, "visual.") && k % 128 == 0 && k >= 256) {
            HIPCHK(h, hipMalloc(&t.q, keepk::q4_data_bytes(n, k)));
            HIPCHK(h, hipMalloc(&t.sc, keepk::q4_scale_bytes(n, k)));
            launch_quant_blockify(dev, t.hi, t.lo, t.q, t.sc, (int)n, (int)k, nullptr);
        } else {
            launch_split_blockify(dev, t.hi, t.lo, (int)n, (int)k, nullptr);
        }
        HIPCHK(h, hipStreamSynchronize(nullptr));
        if (scaled) (void)hipFree(scaled);
    } else {
        const size_t bytes = (size_t)(t.numel > 4 ? t.numel : 4) * sizeof(float);
        HIPCHK(h, hipMalloc(&t.f32, bytes));
        HIPCHK(h, hipMemcpy(t.f32, dev, t.numel * sizeof(float), hipMemcpyDeviceToDevice));
    }
    auto it = h->w.find(key);
    if (it != h->w.end()) {
        if (it->second.f32) hipFree(it->second.f32);
        if (it->second.hi) hipFree(it->second.hi);
        if (it->second.lo) hipFree(it->second.lo);
        if (it->second.q) hipFree(it->second.q);
        if (it->second.sc) hipFree(it->second.sc);
    }
    h->w[key] = t;
    h->finalized = false;
    return KEEP_OK;
}

const float* need_vec(keep_handle* h, const std::string& key, int64_t n, std::string& missing) {
    const WTensor* t = find(h, key);
    if (!t || !t->f32 || t->numel != n) { missing += (missing.empty() ? "" : ", ") + key; return nullptr; }
    return t->f32;
}
const WTensor* need_mat(keep_handle* h, const std::string& key, int64_t n, int64_t k, std::string& missing) {
    const WTensor* t = find(h, key);
    if (!t || !t->hi || t->shape.empty() || t->shape[0] != n || t->numel != n * k) {
        missing += (missing.empty() ? "" : ", ") + key; return nullptr;
    }
    return t;
}

int finalize_vit(keep_handle* h) {
    h->vblocks.clear(); h->vit_depth = 0;
    h->free_cal();               // corrected biases belong to the weights they were calibrated on
    for (float* v : h->owned_vecs) (void)hipFree(v);
    h->owned_vecs.clear();
    const WTensor* pe = find(h, "visual.patch_embed.proj.weight");
    if (!pe) {
        for (auto& kv : h->w) if (starts_with(kv.first, "visual")) return h->fail(KEEP_EKEY, "missing key visual.patch_embed.proj.weight");
        return KEEP_OK;     // image tower not loaded
    }
    if (pe->shape.size() != 4 || pe->shape[1] != 3 || pe->shape[2] != 16 || pe->shape[3] != 16)
        return h->fail(KEEP_EUNSUPPORTED, "patch_embed.proj.weight must be [D,3,16,16]");
    const int64_t D = pe->shape[0];
    if (D % 256 || D > 1024) return h->fail(KEEP_EUNSUPPORTED, "embed dim %lld unsupported", (long long)D);
    int depth = 0;
    while (find(h, "visual.blocks." + std::to_string(depth) + ".attn.qkv.weight")) ++depth;
    if (!depth) return h->fail(KEEP_EKEY, "missing key visual.blocks.0.attn.qkv.weight");
    const WTensor* fc1 = find(h, "visual.blocks.0.mlp.fc1.weight");
    if (!fc1) return h->fail(KEEP_EKEY, "missing key visual.blocks.0.mlp.fc1.weight");
    const int64_t F = fc1->shape[0];
    const WTensor* h0 = find(h, "visual_head.0.weight");
    if (!h0 || h0->shape.size() != 2 || h0->shape[1] != D) return h->fail(KEEP_EKEY, "missing or mis-shaped key visual_head.0.weight");
    const int64_t PJ = h0->shape[0];
    std::string miss;
    need_vec(h, "visual.cls_token", D, miss);
    need_vec(h, "visual.pos_embed", 197 * D, miss);
    need_vec(h, "visual.patch_embed.proj.bias", D, miss);
    need_vec(h, "visual.norm.weight", D, miss);
    need_vec(h, "visual.norm.bias", D, miss);
    need_vec(h, "visual_head.0.bias", PJ, miss);
    need_vec(h, "visual_head.2.weight", PJ * PJ, miss);
    need_vec(h, "visual_head.2.bias", PJ, miss);
    for (int i = 0; i < depth; ++i) {
        const std::string p = "visual.blocks." + std::to_string(i) + ".";
        VitBlock b{};
        b.n1w = need_vec(h, p + "norm1.weight", D, miss); b.n1b = need_vec(h, p + "norm1.bias", D, miss);
        b.n2w = need_vec(h, p + "norm2.weight", D, miss); b.n2b = need_vec(h, p + "norm2.bias", D, miss);
        b.qkv = need_mat(h, p + "attn.qkv.weight", 3 * D, D, miss); b.qkv_b = need_vec(h, p + "attn.qkv.bias", 3 * D, miss);
        b.proj = need_mat(h, p + "attn.proj.weight", D, D, miss);   b.proj_b = need_vec(h, p + "attn.proj.bias", D, miss);
        b.fc1 = need_mat(h, p + "mlp.fc1.weight", F, D, miss);      b.fc1_b = need_vec(h, p + "mlp.fc1.bias", F, miss);
        b.fc2 = need_mat(h, p + "mlp.fc2.weight", D, F, miss);      b.fc2_b = need_vec(h, p + "mlp.fc2.bias", D, miss);
        b.ls1 = need_vec(h, p + "ls1.gamma", D, miss);              b.ls2 = need_vec(h, p + "ls2.gamma", D, miss);
        h->vblocks.push_back(b);
    }
    if (!miss.empty()) return h->fail(KEEP_EKEY, "missing or mis-shaped key(s): %s", miss.c_str());
    if (F % 256 || D % 256 || PJ % 16) return h->fail(KEEP_EUNSUPPORTED, "ViT dims not tileable");
    // pre-scaled proj / fc2 planes (store_tensor): ls * (acc + b) with acc = 2^k * (a . w)  ->  (ls / 2^k) * (acc + 2^k * b), exact in fp32
    auto rescaled = [&](const float* v, float f) -> const float* {
        float* o = nullptr;
        if (hipMalloc(&o, D * sizeof(float)) != hipSuccess) return nullptr;
        launch_scale_vec(v, (int)D, f, o, nullptr);
        h->owned_vecs.push_back(o);
        return o;
    };
    for (auto& b : h->vblocks) {
        if (b.proj->prescale != 1.f) { b.ls1 = rescaled(b.ls1, 1.f / b.proj->prescale); b.proj_b = rescaled(b.proj_b, b.proj->prescale); }
        if (b.fc2->prescale != 1.f) { b.ls2 = rescaled(b.ls2, 1.f / b.fc2->prescale); b.fc2_b = rescaled(b.fc2_b, b.fc2->prescale); }
        if (!b.ls1 || !b.proj_b || !b.ls2 || !b.fc2_b) return h->fail(KEEP_EHIP, "hipMalloc failed for a rescaled LayerScale / bias vector");
    }
    HIPCHK(h, hipStreamSynchronize(nullptr));
    h->vit_has_q = true;
    for (auto& b : h->vblocks) if (!b.fc1->q || !b.fc2->q) h->vit_has_q = false;
    if (depth > keep_handle::MAX_BLOCKS) return h->fail(KEEP_EUNSUPPORTED, "image tower of %d blocks (the per-block precision plan holds %d)", depth, keep_handle::MAX_BLOCKS);
    h->vit_depth = depth; h->vit_D = (int)D; h->vit_heads = (int)(D / 64); h->vit_F = (int)F; h->proj_dim = (int)PJ;
    return KEEP_OK;
}

int finalize_bert(keep_handle* h) {
    for (auto& l : h->blayers) { if (l.qkv.hi) hipFree(l.qkv.hi); if (l.qkv.lo) hipFree(l.qkv.lo); if (l.qkv_b) hipFree(l.qkv_b); }
    h->blayers.clear(); h->bert_layers = 0;
    const WTensor* we = find(h, "text.embeddings.word_embeddings.weight");
    if (!we) {
        for (auto& kv : h->w) if (starts_with(kv.first, "text.")) return h->fail(KEEP_EKEY, "missing key text.embeddings.word_embeddings.weight");
        return KEEP_OK;
    }
    if (we->shape.size() != 2) return h->fail(KEEP_EINVAL, "word_embeddings must be 2-D");
    const int64_t V = we->shape[0], H = we->shape[1];
    if (H != 768 && H != 1024) return h->fail(KEEP_EUNSUPPORTED, "hidden size %lld unsupported (768 or 1024)", (long long)H);
    int L = 0;
    while (find(h, "text.encoder.layer." + std::to_string(L) + ".attention.self.query.weight")) ++L;
    if (!L) return h->fail(KEEP_EKEY, "missing key text.encoder.layer.0.attention.self.query.weight");
    const WTensor* iw = find(h, "text.encoder.layer.0.intermediate.dense.weight");
    if (!iw) return h->fail(KEEP_EKEY, "missing key text.encoder.layer.0.intermediate.dense.weight");
    const int64_t F = iw->shape[0];
    const WTensor* pos = find(h, "text.embeddings.position_embeddings.weight");
    const WTensor* typ = find(h, "text.embeddings.token_type_embeddings.weight");
    if (!pos || !typ || pos->shape.size() != 2 || typ->shape.size() != 2 || pos->shape[1] != H || typ->shape[1] != H)
        return h->fail(KEEP_EKEY, "missing or mis-shaped position/token_type embeddings");
    std::string miss;
    need_vec(h, "text.embeddings.LayerNorm.weight", H, miss);
    need_vec(h, "text.embeddings.LayerNorm.bias", H, miss);
    need_vec(h, "text.pooler.dense.weight", H * H, miss);
    need_vec(h, "text.pooler.dense.bias", H, miss);
    h->blayers.resize(L);
    for (int l = 0; l < L; ++l) {
        const std::string p = "text.encoder.layer." + std::to_string(l) + ".";
        BertLayer& b = h->blayers[l];
        const WTensor* q = need_mat(h, p + "attention.self.query.weight", H, H, miss);
        const WTensor* k = need_mat(h, p + "attention.self.key.weight", H, H, miss);
        const WTensor* v = need_mat(h, p + "attention.self.value.weight", H, H, miss);
        const float* qb = need_vec(h, p + "attention.self.query.bias", H, miss);
        const float* kb = need_vec(h, p + "attention.self.key.bias", H, miss);
        const float* vb = need_vec(h, p + "attention.self.value.bias", H, miss);
        b.o = need_mat(h, p + "attention.output.dense.weight", H, H, miss);
        b.o_b = need_vec(h, p + "attention.output.dense.bias", H, miss);
        b.ln1w = need_vec(h, p + "attention.output.LayerNorm.weight", H, miss);
        b.ln1b = need_vec(h, p + "attention.output.LayerNorm.bias", H, miss);
        b.i = need_mat(h, p + "intermediate.dense.weight", F, H, miss);
        b.i_b = need_vec(h, p + "intermediate.dense.bias", F, miss);
        b.d = need_mat(h, p + "output.dense.weight", H, F, miss);
        b.d_b = need_vec(h, p + "output.dense.bias", H, miss);
        b.ln2w = need_vec(h, p + "output.LayerNorm.weight", H, miss);
        b.ln2b = need_vec(h, p + "output.LayerNorm.bias", H, miss);
        if (!miss.empty()) continue;
        // fuse q|k|v into one [3H,H] weight so the layer needs a single projection GEMM
        b.qkv.shape = {3 * H, H}; b.qkv.numel = 3 * H * H;
        HIPCHK(h, hipMalloc(&b.qkv.hi, b.qkv.numel * sizeof(f16)));
        HIPCHK(h, hipMalloc(&b.qkv.lo, b.qkv.numel * sizeof(f16)));
        HIPCHK(h, hipMalloc(&b.qkv_b, 3 * H * sizeof(float)));
        const WTensor* parts[3] = {q, k, v};
        const float* bparts[3] = {qb, kb, vb};
        for (int j = 0; j < 3; ++j) {
            HIPCHK(h, hipMemcpy(b.qkv.hi + (size_t)j * H * H, parts[j]->hi, H * H * sizeof(f16), hipMemcpyDeviceToDevice));
            HIPCHK(h, hipMemcpy(b.qkv.lo + (size_t)j * H * H, parts[j]->lo, H * H * sizeof(f16), hipMemcpyDeviceToDevice));
            HIPCHK(h, hipMemcpy(b.qkv_b + (size_t)j * H, bparts[j], H * sizeof(float), hipMemcpyDeviceToDevice));
        }
    }
    if (!miss.empty()) { h->blayers.clear(); return h->fail(KEEP_EKEY, "missing or mis-shaped key(s): %s", miss.c_str()); }
    if (F % 256 || H % 256) return h->fail(KEEP_EUNSUPPORTED, "BERT dims not tileable");
    h->bert_layers = L; h->bert_H = (int)H; h->bert_heads = (int)(H / 64); h->bert_F = (int)F;
    h->bert_vocab = (int)V; h->bert_maxpos = (int)pos->shape[0]; h->bert_types = (int)typ->shape[0];
    return KEEP_OK;
}

bool known_key(const std::string& k) {
    static const char* exact[] = {"logit_scale", "visual.cls_token", "visual.pos_embed", "visual.patch_embed.proj.weight",
        "visual.patch_embed.proj.bias", "visual.norm.weight", "visual.norm.bias", "visual_head.0.weight", "visual_head.0.bias",
        "visual_head.2.weight", "visual_head.2.bias", "text.embeddings.word_embeddings.weight",
        "text.embeddings.position_embeddings.weight", "text.embeddings.token_type_embeddings.weight",
        "text.embeddings.LayerNorm.weight", "text.embeddings.LayerNorm.bias", "text.pooler.dense.weight", "text.pooler.dense.bias"};
    for (auto e : exact) if (k == e) return true;
    static const char* vsuf[] = {"norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
        "ls1.gamma", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias", "ls2.gamma"};
    static const char* tsuf[] = {"attention.self.query.weight", "attention.self.query.bias", "attention.self.key.weight",
        "attention.self.key.bias", "attention.self.value.weight", "attention.self.value.bias", "attention.output.dense.weight",
        "attention.output.dense.bias", "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias",
        "intermediate.dense.weight", "intermediate.dense.bias", "output.dense.weight", "output.dense.bias",
        "output.LayerNorm.weight", "output.LayerNorm.bias"};
    auto layered = [&](const char* prefix, const char* const* suf, size_t n) {
        if (!starts_with(k, prefix)) return false;
        size_t i = strlen(prefix), j = i;
        while (j < k.size() && k[j] >= '0' && k[j] <= '9') ++j;
        if (j == i || j >= k.size() || k[j] != '.') return false;
        const std::string rest = k.substr(j + 1);
        for (size_t q = 0; q < n; ++q) if (rest == suf[q]) return true;
        return false;
    };
    return layered("visual.blocks.", vsuf, sizeof vsuf / sizeof *vsuf) || layered("text.encoder.layer.", tsuf, sizeof tsuf / sizeof *tsuf);
}

int tag_by_name(const char* name) {
    for (int i = 0; i < T_COUNT; ++i) if (!strcmp(name, kTagNames[i])) return i;
    return -1;
}

// temp device buffers for the op entry points
struct Tmp {
    std::vector<void*> ptrs;
    ~Tmp() { for (auto p : ptrs) hipFree(p); }
    template <typename T> T* get(size_t n) { void* p = nullptr; if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr; ptrs.push_back(p); return (T*)p; }
};

// one wave: shader-clock cycles (s_memtime) against the constant 100 MHz counter (s_memrealtime) over ~`spin_us` microseconds
__global__ void clock_probe_kernel(long long* out, int spin_ticks) {
    if (threadIdx.x != 0) return;
    const long long r0 = (long long)__builtin_amdgcn_s_memrealtime(), c0 = (long long)__builtin_readcyclecounter();
    long long r1 = r0;
    while (r1 - r0 < spin_ticks) { __builtin_amdgcn_s_sleep(32); r1 = (long long)__builtin_amdgcn_s_memrealtime(); }
    const long long c1 = (long long)__builtin_readcyclecounter();
    out[0] = c1 - c0; out[1] = r1 - r0;
}

__global__ void f16_planes_to_f32_kernel(const f16* hi, const f16* lo, float* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)hi[i] + (lo ? (float)lo[i] : 0.f);
}
void planes_to_f32(const f16* hi, const f16* lo, float* out, int64_t n, hipStream_t s) {
    int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(f16_planes_to_f32_kernel, dim3(blocks), dim3(256), 0, s, hi, lo, out, n);
}

// the image tower on B tiles (arguments checked, device selected by the caller)
int encode_image_run(keep_handle* h, const void* pixels, int pix_dtype, int64_t B, float* out, hipStream_t s) {
    ++h->dbg_calls;
    if (h->use_graphs && !h->prof_mode && B * 197 <= SKINNY_MAX_M && B <= h->max_tiles) {
        const size_t pxb = pix_dtype == KEEP_PIX_F32 ? 4 : (pix_dtype == KEEP_PIX_U8_HWC ? 1 : 2);
        const bool sp = h->any_split();
        const size_t ws_bytes = align_up(vit_ws_bytes(h, B, sp)), ib = (size_t)B * 3 * 224 * 224 * pxb, ob = (size_t)B * h->proj_dim * sizeof(float);
        int rc = ensure_arena(h, ws_bytes + align_up(ib) + align_up(ob));
        if (rc) return rc;
        char* st_pix = h->arena + ws_bytes;
        float* st_out = (float*)(h->arena + ws_bytes + align_up(ib));
        HIPCHK(h, hipMemcpyAsync(st_pix, pixels, ib, hipMemcpyDeviceToDevice, s));
        char key[96];
        snprintf(key, sizeof key, "img|%lld|%d|%d", (long long)B, pix_dtype, h->precision);     // (keep_classify switches the precision per call, without an option epoch)
        rc = graph_run(h, key, s, [&](hipStream_t cs) {
            VitLane L{};
            L.Bc = (int)B; L.pixels = st_pix; L.pix_dtype = pix_dtype; L.out = st_out; L.s = cs;
            L.ws = carve_vit(h, h->arena, L.Bc, sp);
            int r = vit_begin(h, L);
            for (int i = 0; !r && i < h->vit_depth; ++i) r = vit_layer(h, L, i);
            return r ? r : vit_end(h, L);
        });
        if (rc) return rc;
        HIPCHK(h, hipMemcpyAsync(out, st_out, ob, hipMemcpyDeviceToDevice, s));
        return KEEP_OK;
    }
    const size_t px = pix_dtype == KEEP_PIX_F32 ? 4 : (pix_dtype == KEEP_PIX_U8_HWC ? 1 : 2);    // bytes per value; 3*224*224 values per tile in every layout
    // lanes: split the batch over n_streams concurrent sub-batches once there is enough work for each
    int lanes = h->n_streams;
    while (lanes > 1 && B < (int64_t)lanes * h->lane_min_tiles) --lanes;
    int64_t per = (B + lanes - 1) / lanes;
    if (per > h->max_tiles) per = h->max_tiles;
    const bool split = h->any_split();
    const int64_t chunk_max = per * lanes;
    const bool uneven = lanes == 2 && h->lane0_permille != 500;
    const int64_t lane_cap = uneven ? (chunk_max * (h->lane0_permille > 500 ? h->lane0_permille : 1000 - h->lane0_permille) + 999) / 1000 : per;
    const size_t lane_bytes = align_up(vit_ws_bytes(h, lane_cap, split));
    int rc = ensure_arena(h, lane_bytes * lanes);
    if (rc) return rc;
    if (lanes > 1) {
        if (!h->ev_fork) HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        for (int l = 0; l < lanes; ++l) {
            if (!h->aux[l]) HIPCHK(h, hipStreamCreateWithFlags(&h->aux[l], hipStreamNonBlocking));
            if (!h->ev_join[l]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_join[l], hipEventDisableTiming));
        }
        HIPCHK(h, hipEventRecord(h->ev_fork, s));
        for (int l = 0; l < lanes; ++l) HIPCHK(h, hipStreamWaitEvent(h->aux[l], h->ev_fork, 0));
    }
    for (int64_t b0 = 0; b0 < B; b0 += per * lanes) {
        VitLane L[4];
        int nl = 0;
        const int64_t chunk = (B - b0) < chunk_max ? (B - b0) : chunk_max;
        const int64_t n0 = uneven ? (chunk * h->lane0_permille + 500) / 1000 : per;
        for (int l = 0; l < lanes; ++l) {
            const int64_t lo = b0 + (uneven ? (l ? n0 : 0) : l * per);
            const int64_t cap = uneven ? (l ? chunk - n0 : n0) : per;
            if (lo >= B || cap <= 0) break;
            VitLane& x = L[nl++];
            x.Bc = (int)((B - lo) < cap ? (B - lo) : cap);
            x.pixels = (const char*)pixels + (size_t)lo * 3 * 224 * 224 * px;
            x.pix_dtype = pix_dtype;
            x.out = out + lo * h->proj_dim;
            x.s = lanes > 1 ? h->aux[l] : s;
            x.ws = carve_vit(h, h->arena + (size_t)l * lane_bytes, x.Bc, split);
            if (lanes > 1 && h->cls_side_stream && !h->capture && x.Bc >= h->lane_min_tiles) {
                if (!h->aux_cls[l]) HIPCHK(h, hipStreamCreateWithFlags(&h->aux_cls[l], hipStreamNonBlocking));
                for (int e = 0; e < 3; ++e) if (!h->ev_cls[l][e]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_cls[l][e], hipEventDisableTiming));
                if (!h->cls_splitk[l]) HIPCHK(h, hipMalloc(&h->cls_splitk[l], SKINNY_WS_BYTES));
                x.cs = h->aux_cls[l]; x.cs_splitk = h->cls_splitk[l];
                for (int e = 0; e < 3; ++e) x.ce[e] = h->ev_cls[l][e];
            }
        }
        const bool skew = nl > 1 && h->lane_skew > 0;
        for (int l = 0; l < nl; ++l) {
            if (skew) {
                if (!h->ev_skew[l]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_skew[l], hipEventDisableTiming));
                L[l].skew_ev = h->ev_skew[l]; L[l].skew_stage = h->lane_skew;
                if (l > 0) HIPCHK(h, hipStreamWaitEvent(L[l].s, h->ev_skew[l - 1], 0));
            }
            if ((rc = vit_begin(h, L[l]))) return rc;
            if (skew && (rc = vit_layer(h, L[l], 0))) return rc;
        }
        for (int i = skew ? 1 : 0; i < h->vit_depth; ++i)
            for (int l = 0; l < nl; ++l) if ((rc = vit_layer(h, L[l], i))) return rc;
        for (int l = 0; l < nl; ++l) if ((rc = vit_end(h, L[l]))) return rc;
    }
    if (lanes > 1)
        for (int l = 0; l < lanes; ++l) {
            HIPCHK(h, hipEventRecord(h->ev_join[l], h->aux[l]));
            HIPCHK(h, hipStreamWaitEvent(s, h->ev_join[l], 0));
        }
    return KEEP_OK;
}


int similarity_run(keep_handle* h, const float* img, const float* txt, int64_t N, int64_t P, int64_t D, float scale, int mode,
                   void* out, int32_t* argmax_out, hipStream_t s) {
    Scope sc(h, T_SIM, s);
    if (h->tune.sgemv_m > 0 && launch_sim_small(img, txt, (int)N, (int)P, (int)D, scale, mode, out, argmax_out, s) == 0)
        return check_launch(h, "similarity");
    if (h->tune.sgemv_m > 0 && launch_sim_mid(img, txt, (int)N, (int)P, (int)D, scale, mode, out, argmax_out, s) == 0)
        return check_launch(h, "similarity");
    float* logits = (float*)out;
    const bool need_tmp = (mode == KEEP_SIM_ARGMAX && !out) || mode == KEEP_SIM_SOFTMAX_F16 || mode == KEEP_SIM_TOP2SCORE;
    const int nb = (int)((N + 255) / 256);
    if (need_tmp) {
        // scratch lives at the tail end of the arena so that a preceding encode on the same stream
        // (which uses the front) is not disturbed; stream order serialises reuse.
        const size_t bytes = align_up((size_t)N * P * 4) + align_up((size_t)nb * 4) + 256;
        int rc = ensure_arena(h, bytes);
        if (rc) return rc;
        logits = (float*)h->arena;
    }
    SgemmParams g{};
    g.tune = &h->tune;
    g.a = img; g.lda = D; g.b = txt; g.ldb = D; g.out = logits; g.ldo = P; g.bias = nullptr;
    g.M = (int)N; g.N = (int)P; g.K = (int)D; g.act = ACT_NONE;
    g.scale = (mode == KEEP_SIM_RAW || mode == KEEP_SIM_ARGMAX) ? scale : 1.0f;
    if (launch_sgemm_f32(g, s)) return h->fail(KEEP_EUNSUPPORTED, "similarity shape");
    if (mode == KEEP_SIM_ARGMAX) launch_row_argmax(logits, (int)N, (int)P, argmax_out, s);
    else if (mode == KEEP_SIM_SOFTMAX) launch_row_softmax(logits, (int)N, (int)P, scale, logits, s);
    else if (mode == KEEP_SIM_SOFTMAX_F16) launch_row_softmax_f16(logits, (int)N, (int)P, scale, (f16*)out, s);
    else if (mode == KEEP_SIM_TOP2SCORE) {
        float* partial = (float*)(h->arena + align_up((size_t)N * P * 4));
        launch_top2_score(logits, (int)N, (int)P, partial, (float*)out, s);
    }
    return check_launch(h, "similarity");
}


}  // namespace

// =============================================================================================
extern "C" {

const char* keep_version(void) { return "keep_hip 0.1 (gfx950)"; }

int keep_create(int device_id, keep_handle** out) {
    if (!out) return KEEP_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return KEEP_EHIP;
    DevGuard guard(device_id);
    if (!guard.ok) return KEEP_EHIP;
    keep_handle* h = new keep_handle();
    h->device = device_id;
    if (hipMalloc(&h->err_flag, 4 * sizeof(int)) != hipSuccess) { delete h; return KEEP_ENOMEM; }     // [0] sticky error bits, [2..3] load-time weight statistics
    hipMemset(h->err_flag, 0, 4 * sizeof(int));
    *out = h;
    return KEEP_OK;
}

int keep_destroy(keep_handle* h) {
    if (!h) return KEEP_OK;
    DevGuard guard(h->device);
    hipDeviceSynchronize();
    h->prof_collect();
    for (auto& e : h->pool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto& kv : h->w) { if (kv.second.f32) hipFree(kv.second.f32); if (kv.second.hi) hipFree(kv.second.hi); if (kv.second.lo) hipFree(kv.second.lo);
                            if (kv.second.q) hipFree(kv.second.q); if (kv.second.sc) hipFree(kv.second.sc); }
    if (h->tune.dbg) hipFree(h->tune.dbg);
    for (float* v : h->owned_vecs) hipFree(v);
    h->free_cal();
    for (auto& l : h->blayers) { if (l.qkv.hi) hipFree(l.qkv.hi); if (l.qkv.lo) hipFree(l.qkv.lo); if (l.qkv_b) hipFree(l.qkv_b); }
    drop_graphs(h);
    if (h->cap_stream) hipStreamDestroy(h->cap_stream);
    if (h->arena) hipFree(h->arena);
    if (h->cls_buf) hipFree(h->cls_buf);
    if (h->err_flag) hipFree(h->err_flag);
    for (int l = 0; l < 4; ++l) { if (h->aux[l]) hipStreamDestroy(h->aux[l]); if (h->ev_join[l]) hipEventDestroy(h->ev_join[l]); }
    for (int l = 0; l < 4; ++l) {
        if (h->aux_cls[l]) hipStreamDestroy(h->aux_cls[l]);
        for (int e = 0; e < 3; ++e) if (h->ev_cls[l][e]) hipEventDestroy(h->ev_cls[l][e]);
        if (h->cls_splitk[l]) (void)hipFree(h->cls_splitk[l]);
    }
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    delete h;
    return KEEP_OK;
}

const char* keep_last_error(keep_handle* h) { return h ? h->err.c_str() : "null handle"; }

const char* keep_load_warnings(keep_handle* h) {
    if (!h) return "";
    static thread_local std::string out;
    out.swap(h->load_warnings);
    h->load_warnings.clear();
    return out.c_str();
}

int keep_load_tensor(keep_handle* h, const char* key, const float* data, int ndim, const int64_t* shape, int on_device) {
    if (!h || !key || !data || ndim < 0 || ndim > 8) return KEEP_EINVAL;
    KEEP_ON_DEVICE(h);
    const std::string k(key);
    if (k == "text.embeddings.position_ids" || k == "text.embeddings.token_type_ids") return KEEP_OK;   // buffers of older checkpoints
    if (!known_key(k)) return h->fail(KEEP_EKEY, "unexpected key %s", key);
    std::vector<int64_t> shp(shape, shape + ndim);
    if (ndim == 0) shp = {1};
    const int64_t n = numel_of(shp);
    if (n <= 0) return h->fail(KEEP_EINVAL, "%s: bad shape", key);
    if (on_device) {
        // the repack below runs on the null stream; whatever produced `data` (e.g. a dtype conversion on the caller's
        // stream) must have finished first, and this entry point takes no stream: load time, so simply drain the device
        HIPCHK(h, hipDeviceSynchronize());
        return store_tensor(h, k, data, shp);
    }
    float* tmp = nullptr;
    HIPCHK(h, hipMalloc(&tmp, n * sizeof(float)));
    hipError_t e = hipMemcpy(tmp, data, n * sizeof(float), hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? store_tensor(h, k, tmp, shp) : h->fail(KEEP_EHIP, "H2D copy of %s failed", key);
    hipFree(tmp);
    return rc;
}

int keep_finalize_weights(keep_handle* h) {
    if (!h) return KEEP_EINVAL;
    KEEP_ON_DEVICE(h);
    ++h->opt_epoch;
    int rc = finalize_vit(h);
    if (rc) return rc;
    rc = finalize_bert(h);
    if (rc) return rc;
    if (!h->vit_depth && !h->bert_layers) return h->fail(KEEP_EKEY, "no tower loaded");
    HIPCHK(h, hipDeviceSynchronize());
    h->finalized = true;
    return KEEP_OK;
}

int keep_vit_depth(keep_handle* h) { return h && h->finalized ? h->vit_depth : 0; }
int keep_bert_layers(keep_handle* h) { return h && h->finalized ? h->bert_layers : 0; }

int keep_set_option(keep_handle* h, const char* name, double value) {
    if (!h || !name) return KEEP_EINVAL;
    const std::string n(name);
    const int v = (int)value;
    ++h->opt_epoch;               // captured graphs bake kernel selection and precision in: drop them lazily
    KeepTune& t = h->tune;
    if (n == "graphs") { h->use_graphs = v ? 1 : 0; return KEEP_OK; }
    if (n == "label_margin") { if (!(value >= 0.0) || value > 2.0) return h->fail(KEEP_EINVAL, "label_margin must be in [0, 2]"); h->label_margin = (float)value; return KEEP_OK; }
    if (n == "precision") { if (v != KEEP_PREC_FP16 && v != KEEP_PREC_STRICT && v != KEEP_PREC_COMP) return h->fail(KEEP_EINVAL, "precision %d", v); h->precision = v; }
    else if (n == "strict_blocks") { if (v < 0) return h->fail(KEEP_EINVAL, "strict_blocks < 0"); h->strict_blocks = v; }
    // the four prefix shorthands rewrite the whole per-block plan (a plan set block by block through keep_set_block_precision is replaced)
    else if (n == "comp_full_blocks") { if (v < 0) return h->fail(KEEP_EINVAL, "comp_full_blocks < 0"); h->comp_full_blocks = v; h->plan_from_prefix(); }
    else if (n == "comp_mlp_blocks") { if (v < 0) return h->fail(KEEP_EINVAL, "comp_mlp_blocks < 0"); h->comp_mlp_blocks = v; h->plan_from_prefix(); }
    else if (n == "comp_qkv") { h->comp_qkv = v ? 1 : 0; h->plan_from_prefix(); }
    else if (n == "comp_qkv_from") { if (v < 0) return h->fail(KEEP_EINVAL, "comp_qkv_from < 0"); h->comp_qkv_from = v; h->plan_from_prefix(); }
    else if (n == "comp_min_tiles") { if (v < 24) return h->fail(KEEP_EINVAL, "comp_min_tiles must be >= 24 (the compensated product needs the 256x256 kernel)"); h->comp_min_tiles = v; }
    else if (n == "fused_screening") { if (v < 0 || v > 2) return h->fail(KEEP_EINVAL, "fused_screening must be 0..2"); h->fused_screening = v; }
    else if (n == "max_tiles") { if (v < 1) return h->fail(KEEP_EINVAL, "max_tiles < 1"); h->max_tiles = v; }
    else if (n == "max_prompts") { if (v < 1) return h->fail(KEEP_EINVAL, "max_prompts < 1"); h->max_prompts = v; }
    else if (n == "cls_tail") { h->cls_tail = v ? 1 : 0; if (h->bias_ready && h->cal_cls_tail != h->cls_tail) h->bias_ready = false; }   // (the mean-input biases of the last block were averaged under the other setting: recalibrate)
    else if (n == "cls_qkv") { h->cls_qkv = v ? 1 : 0; }
    else if (n == "cls_side_stream") { h->cls_side_stream = v ? 1 : 0; }
    else if (n == "cls_chain_early") { if (v < 0 || v > 2) return h->fail(KEEP_EINVAL, "cls_chain_early must be 0..2"); h->cls_chain_early = v; }
    else if (n == "patch_split") { h->patch_split = v ? 1 : 0; }
    else if (n == "bias_correction") { h->bias_correction = v ? 1 : 0; }
    else if (n == "impl2128_mask") { if (v < 0 || v > 15) return h->fail(KEEP_EINVAL, "impl2128_mask must be 0..15"); h->impl2128_mask = v; }
    else if (n == "proj_impl") { if (v != 0 && v != 2128) return h->fail(KEEP_EINVAL, "proj_impl must be 0 or 2128"); h->proj_impl = v; }
    else if (n == "streams") { if (v < 1 || v > 4) return h->fail(KEEP_EINVAL, "streams must be 1..4"); h->n_streams = v; }
    else if (n == "gemm_persistent") { if (v < 0 || v > 1024) return h->fail(KEEP_EINVAL, "gemm_persistent must be 0..1024"); t.gemm_persistent = v; }
    else if (n == "gemm_splitk_tiles") { if (v < 0 || v > 256) return h->fail(KEEP_EINVAL, "gemm_splitk_tiles must be 0..256"); t.gemm_splitk_tiles = v; }
    else if (n == "sgemv_m") { if (v < 0 || v > 16) return h->fail(KEEP_EINVAL, "sgemv_m must be 0..16"); t.sgemv_m = v; }
    else if (n == "skinny_wide") { t.skinny_wide = v ? 1 : 0; }
    else if (n == "gemm_skinny_m") { if (v < 0 || v > SKINNY_MAX_M) return h->fail(KEEP_EINVAL, "gemm_skinny_m must be 0..%d", SKINNY_MAX_M); t.gemm_skinny_m = v; }
    else if (n == "lane_min_tiles") { if (v < 6) return h->fail(KEEP_EINVAL, "lane_min_tiles must be >= 6"); h->lane_min_tiles = v; }
    else if (n == "lane_skew") { if (v < 0 || v > 5) return h->fail(KEEP_EINVAL, "lane_skew must be 0..5"); h->lane_skew = v; }
    else if (n == "lane0_permille") { if (v < 100 || v > 900) return h->fail(KEEP_EINVAL, "lane0_permille must be 100..900"); h->lane0_permille = v; }
    else if (n == "ln_impl") { if (v < 0 || v > 2) return h->fail(KEEP_EINVAL, "ln_impl must be 0, 1 or 2"); t.ln_impl = v; }
    else if (n == "attn_waves") { if (v != 4 && v != 8 && v != 16) return h->fail(KEEP_EINVAL, "attn_waves must be 4, 8 or 16 (16: persistent double-buffered kernel for the image tower)"); t.attn_waves = v; }
    else if (n == "gemm_impl") {
        bool ok = v == 0 || v == 128 || v == 256;
        if (!ok) return h->fail(KEEP_EINVAL, "gemm_impl %d (0, 128, 256)", v);
        t.gemm_impl = v;
    }
#ifdef KEEP_DIAGNOSTICS
    // result-changing / timing diagnostics exist only in -DKEEP_DIAGNOSTICS builds (tools/gemm_timeline.py, tools/attn_timeline.py)
    else if (n == "dbg_skip_ln") h->dbg_skip_ln = v;
    else if (n == "gemm_ablate") { t.gemm_ablate = v; }
    else if (n == "gemm_dbg") {
        if (v && !t.dbg) { HIPCHK(h, hipMalloc(&t.dbg, (size_t)65536 * 4 * sizeof(long long))); HIPCHK(h, hipMemset(t.dbg, 0, (size_t)65536 * 4 * sizeof(long long))); }
        if (!v && t.dbg) { hipFree(t.dbg); t.dbg = nullptr; }
    }
#endif
    else return h->fail(KEEP_EINVAL, "unknown option %s", name);
    return KEEP_OK;
}
double keep_get_option(keep_handle* h, const char* name) {
    if (!h || !name) return -1;
    const std::string n(name);
    const KeepTune& t = h->tune;
    if (n == "precision") return h->precision;
    if (n == "label_margin") return h->label_margin;
    if (n == "strict_blocks") return h->strict_blocks;
    if (n == "comp_full_blocks") return h->comp_full_blocks;
    if (n == "comp_mlp_blocks") return h->comp_mlp_blocks;
    if (n == "comp_min_tiles") return h->comp_min_tiles;
    if (n == "comp_qkv") return h->comp_qkv;
    if (n == "comp_qkv_from") return h->comp_qkv_from;
    if (n == "plan_custom") return h->plan_custom ? 1 : 0;
    if (n == "max_tiles") return h->max_tiles;
    if (n == "max_prompts") return h->max_prompts;
    if (n == "gemm_impl") return t.gemm_impl;
    if (n == "streams") return h->n_streams;
    if (n == "graphs") return h->use_graphs;
    if (n == "gemm_skinny_m") return t.gemm_skinny_m;
    if (n == "skinny_wide") return t.skinny_wide;
    if (n == "sgemv_m") return t.sgemv_m;
    if (n == "gemm_splitk_tiles") return t.gemm_splitk_tiles;
    if (n == "gemm_persistent") return t.gemm_persistent;
    if (n == "ln_impl") return t.ln_impl;
    if (n == "attn_waves") return t.attn_waves;
    if (n == "lane_skew") return h->lane_skew;
    if (n == "lane0_permille") return h->lane0_permille;
    if (n == "cls_tail") return h->cls_tail;
    if (n == "proj_impl") return h->proj_impl;
    if (n == "impl2128_mask") return h->impl2128_mask;
    if (n == "cls_qkv") return h->cls_qkv;
    if (n == "cls_side_stream") return h->cls_side_stream;
    if (n == "cls_chain_early") return h->cls_chain_early;
    if (n == "patch_split") return h->patch_split;
    if (n == "bias_correction") return h->bias_correction;
    if (n == "bias_ready") return h->bias_ready ? 1 : 0;
    return -1;
}

int keep_set_block_precision(keep_handle* h, int block, int attn_mode, int mlp_mode) {
    if (!h) return KEEP_EINVAL;
    if (block < 0 || block >= keep_handle::MAX_BLOCKS) return h->fail(KEEP_EINVAL, "block %d outside 0..%d", block, keep_handle::MAX_BLOCKS - 1);
    if (attn_mode > KEEP_ATTN_COMPQKV_PROJ_CLS || mlp_mode > KEEP_MLP_CLS) return h->fail(KEEP_EINVAL, "attn_mode %d (0..5) / mlp_mode %d (0..4); negative = leave", attn_mode, mlp_mode);
    ++h->opt_epoch;               // captured graphs bake the plan in
    if (attn_mode >= 0) h->attn_mode[block] = (unsigned char)attn_mode;
    if (mlp_mode >= 0) h->mlp_mode[block] = (unsigned char)mlp_mode;
    h->plan_custom = true;
    return KEEP_OK;
}
int keep_get_block_precision(keep_handle* h, int block, int* attn_mode, int* mlp_mode) {
    if (!h) return KEEP_EINVAL;
    if (block < 0 || block >= keep_handle::MAX_BLOCKS) return h->fail(KEEP_EINVAL, "block %d outside 0..%d", block, keep_handle::MAX_BLOCKS - 1);
    if (attn_mode) *attn_mode = h->attn_mode[block];
    if (mlp_mode) *mlp_mode = h->mlp_mode[block];
    return KEEP_OK;
}

int keep_reserve(keep_handle* h, int64_t tiles, int64_t prompts, int64_t seq) {
    if (!h || !h->finalized) return h ? h->fail(KEEP_ESTATE, "weights not finalised") : KEEP_EINVAL;
    KEEP_ON_DEVICE(h);
    size_t need = 0;
    if (tiles > 0 && h->vit_depth) {
        int lanes = h->n_streams;
        while (lanes > 1 && tiles < (int64_t)lanes * h->lane_min_tiles) --lanes;
        int64_t per = (tiles + lanes - 1) / lanes;
        if (per > h->max_tiles) per = h->max_tiles;
        need = align_up(vit_ws_bytes(h, per, h->any_split())) * lanes;
        if (tiles * 197 <= SKINNY_MAX_M)          // graph-replayed call: + staged pixels (fp32 at most) and outputs
            need += align_up((size_t)tiles * 3 * 224 * 224 * 4) + align_up((size_t)tiles * h->proj_dim * 4);
    }
    if (prompts > 0 && seq > 0 && h->bert_layers) {
        const int64_t pc = prompts < h->max_prompts ? prompts : h->max_prompts;
        size_t t = align_up(txt_ws_bytes(h, pc, seq, h->any_split()));
        if (prompts * seq <= TXT_GRAPH_ROWS)      // graph-replayed call: + staged ids / types / mask and outputs
            t += 3 * align_up((size_t)prompts * seq * 8) + align_up((size_t)prompts * h->bert_H * 4);
        need = t > need ? t : need;
    }
    return ensure_arena(h, need);
}
int64_t keep_workspace_bytes(keep_handle* h) { return h ? (int64_t)h->arena_bytes : 0; }

int keep_encode_image(keep_handle* h, const void* pixels, int pix_dtype, int64_t B, float* out, void* stream) {
    if (!h) return KEEP_EINVAL;
    if (!h->finalized || !h->vit_depth) return h->fail(KEEP_ESTATE, "image tower not loaded / finalised");
    if (!pixels || !out || B < 0) return h->fail(KEEP_EINVAL, "null pointer or negative batch");
    if (pix_dtype < KEEP_PIX_F32 || pix_dtype > KEEP_PIX_U8_HWC) return h->fail(KEEP_EINVAL, "pixel dtype %d", pix_dtype);
    if (B == 0) return KEEP_OK;
    KEEP_ON_DEVICE(h);
    return encode_image_run(h, pixels, pix_dtype, B, out, (hipStream_t)stream);
}

int keep_encode_text(keep_handle* h, const int64_t* ids, const int64_t* types, const int64_t* mask, int64_t P, int64_t T,
                     float* out, void* stream) {
    if (!h) return KEEP_EINVAL;
    if (!h->finalized || !h->bert_layers) return h->fail(KEEP_ESTATE, "text tower not loaded / finalised");
    if (!ids || !out || P < 0 || T < 1) return h->fail(KEEP_EINVAL, "null pointer or bad shape");
    if (T > h->bert_maxpos) return h->fail(KEEP_EINVAL, "sequence length %lld exceeds max_position_embeddings %d", (long long)T, h->bert_maxpos);
    if (T > 512) return h->fail(KEEP_EUNSUPPORTED, "sequence length %lld unsupported (the attention kernels cover 512 tokens, BertModel's max_position_embeddings)", (long long)T);
    if (P == 0) return KEEP_OK;
    KEEP_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    const int64_t pc_max = P < h->max_prompts ? P : h->max_prompts;
    const size_t ws_bytes = align_up(txt_ws_bytes(h, pc_max, T, h->any_split()));
    if (h->use_graphs && !h->prof_mode && P * T <= TXT_GRAPH_ROWS && P <= h->max_prompts) {
        // launch-bound size: stage the caller's tensors into fixed buffers and replay the whole tower as one graph
        const size_t nb = (size_t)P * T * sizeof(int64_t), ob = (size_t)P * h->bert_H * sizeof(float);
        int rc = ensure_arena(h, ws_bytes + 3 * align_up(nb) + align_up(ob));
        if (rc) return rc;
        int64_t* st_ids = (int64_t*)(h->arena + ws_bytes);
        int64_t* st_types = (int64_t*)(h->arena + ws_bytes + align_up(nb));
        int64_t* st_mask = (int64_t*)(h->arena + ws_bytes + 2 * align_up(nb));
        float* st_out = (float*)(h->arena + ws_bytes + 3 * align_up(nb));
        HIPCHK(h, hipMemcpyAsync(st_ids, ids, nb, hipMemcpyDeviceToDevice, s));
        if (types) HIPCHK(h, hipMemcpyAsync(st_types, types, nb, hipMemcpyDeviceToDevice, s));
        if (mask) HIPCHK(h, hipMemcpyAsync(st_mask, mask, nb, hipMemcpyDeviceToDevice, s));
        char key[96];
        snprintf(key, sizeof key, "txt|%lld|%lld|%d|%d", (long long)P, (long long)T, types ? 1 : 0, mask ? 1 : 0);
        // (the token-range flag is sticky: set by the embedding kernel, cleared only by keep_token_error once the host has seen it --
        // so nothing has to be reset per call, and no memset node is captured: one replayed with a stale value on ROCm 7.2)
        rc = graph_run(h, key, s, [&](hipStream_t cs) {
            return txt_chunk(h, st_ids, types ? st_types : nullptr, mask ? st_mask : nullptr, (int)P, (int)T, st_out, cs);
        });
        if (rc) return rc;
        HIPCHK(h, hipMemcpyAsync(out, st_out, ob, hipMemcpyDeviceToDevice, s));
        return KEEP_OK;
    }
    int rc = ensure_arena(h, ws_bytes);
    if (rc) return rc;
    for (int64_t p0 = 0; p0 < P; p0 += pc_max) {
        const int pc = (int)((P - p0) < pc_max ? (P - p0) : pc_max);
        rc = txt_chunk(h, ids + p0 * T, types ? types + p0 * T : nullptr, mask ? mask + p0 * T : nullptr, pc, (int)T,
                       out + p0 * h->bert_H, s);
        if (rc) return rc;
    }
    return KEEP_OK;
}

/* keep_calibrate_bias (include/keep_hip.h): mean-input compensation of the weight-rounding error, measured on the caller's tiles. */
int keep_calibrate_bias(keep_handle* h, const void* pixels, int pix_dtype, int64_t B, void* stream) {
    if (!h) return KEEP_EINVAL;
    if (!h->finalized || !h->vit_depth) return h->fail(KEEP_ESTATE, "image tower not loaded / finalised");
    if (pix_dtype < KEEP_PIX_F32 || pix_dtype > KEEP_PIX_U8_HWC) return h->fail(KEEP_EINVAL, "pixel dtype %d", pix_dtype);
    if (B == 0) { h->bias_ready = false; ++h->opt_epoch; return KEEP_OK; }           // zero tiles: forget the calibration
    if (!pixels || B < 8) return h->fail(KEEP_EINVAL, "bias calibration needs at least 8 tiles");
    KEEP_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    ++h->opt_epoch;
    const int D = h->vit_D, F = h->vit_F;
    const int widths[4] = {D, D, D, F}, outs[4] = {3 * D, D, F, D};
    h->bias_ready = false;
    if ((int)h->cal.size() != h->vit_depth) {
        h->free_cal();
        h->cal.resize(h->vit_depth);
        bool ok = true;
        for (auto& c : h->cal)
            for (int k = 0; k < 4 && ok; ++k)
                ok = hipMalloc(&c.sum[k], widths[k] * sizeof(float)) == hipSuccess && hipMalloc(&c.bias[k], outs[k] * sizeof(float)) == hipSuccess;
        if (!ok) {                                   // never leave a half-allocated table behind: the next call would skip the allocation
            (void)hipGetLastError();
            h->free_cal();
            return h->fail(KEEP_ENOMEM, "bias calibration: out of device memory");
        }
    }
    for (auto& c : h->cal)
        for (int k = 0; k < 4; ++k) {
            if (hipMemsetAsync(c.sum[k], 0, widths[k] * sizeof(float), s) != hipSuccess) { (void)hipGetLastError(); h->free_cal(); return h->fail(KEEP_EHIP, "bias calibration: memset failed"); }
            c.rows[k] = 0;
        }
    h->cal_cls_tail = h->cls_tail;                   // the last block's sites 1-3 average the CLS rows only when cls_tail is on: the biases belong to that setting
    float* scratch = nullptr;
    HIPCHK(h, hipMalloc(&scratch, (size_t)B * h->proj_dim * sizeof(float)));
    // one lane, no graph replay: every site is visited once per sub-batch, on ONE stream, so the sums accumulate in a fixed order
    const int streams_was = h->n_streams, graphs_was = h->use_graphs, prec_was = h->precision;
    h->n_streams = 1; h->use_graphs = 0; h->precision = KEEP_PREC_STRICT; h->capture = true;      // split products: the cleanest activations to average
    int rc = encode_image_run(h, pixels, pix_dtype, B, scratch, s);
    h->n_streams = streams_was; h->use_graphs = graphs_was; h->precision = prec_was; h->capture = false;
    if (!rc) {
        for (int i = 0; i < h->vit_depth && !rc; ++i) {
            const VitBlock& b = h->vblocks[i];
            const WTensor* w[4] = {b.qkv, b.proj, b.fc1, b.fc2};
            const float* bias[4] = {b.qkv_b, b.proj_b, b.fc1_b, b.fc2_b};
            for (int k = 0; k < 4; ++k) {
                if (!w[k]->lo || !(h->cal[i].rows[k] > 0)) { rc = h->fail(KEEP_ESTATE, "block %d site %d was not visited by the calibration encode", i, k); break; }
                launch_bias_mean_corr(w[k]->lo, h->cal[i].sum[k], (float)(1.0 / h->cal[i].rows[k]), bias[k], h->cal[i].bias[k], outs[k], widths[k], s);
            }
        }
    }
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(scratch);
    if (rc) return rc;
    HIPCHK(h, e);
    h->bias_ready = true;
    return check_launch(h, "calibrate_bias");
}

int keep_resize_crop_u8(keep_handle* h, const unsigned char* src, int64_t B, int64_t H, int64_t W, const int32_t* xbounds, const int32_t* xweights,
                        int xksize, int64_t out_w, const int32_t* ybounds, const int32_t* yweights, int yksize, int64_t out_h,
                        int64_t crop_left, int64_t crop_top, int64_t size, unsigned char* out, void* stream) {
    if (!h) return KEEP_EINVAL;
    if (!src || !xbounds || !xweights || !ybounds || !yweights || !out || B < 0 || H < 1 || W < 1 || xksize < 1 || yksize < 1)
        return h->fail(KEEP_EINVAL, "bad resize arguments");
    if (size < 1 || crop_left < 0 || crop_top < 0 || crop_left + size > out_w || crop_top + size > out_h)
        return h->fail(KEEP_EINVAL, "crop window [%lld+%lld, %lld+%lld] outside the resized image %lldx%lld", (long long)crop_left, (long long)size,
                       (long long)crop_top, (long long)size, (long long)out_w, (long long)out_h);
    if (B == 0) return KEEP_OK;
    KEEP_ON_DEVICE(h);
    // the horizontally resized rows live in the arena; whatever runs next on this stream (e.g. the encode of the cropped tiles) is
    // ordered behind the two kernels, so reusing the arena there is safe
    const size_t tmp_bytes = align_up((size_t)B * H * size * 3);
    int rc = ensure_arena(h, tmp_bytes);
    if (rc) return rc;
    launch_resize_crop_u8(src, (int)B, (int)H, (int)W, xbounds, xweights, xksize, (int)crop_left, (int)size, ybounds, yweights, yksize,
                          (int)crop_top, (int)size, (unsigned char*)h->arena, out, (hipStream_t)stream);
    return check_launch(h, "resize_crop_u8");
}

int keep_token_error(keep_handle* h, void* stream) {
    if (!h) return KEEP_EINVAL;
    KEEP_ON_DEVICE(h);
    int flag = 0;
    HIPCHK(h, hipMemcpyAsync(&flag, h->err_flag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    if (flag) HIPCHK(h, hipMemsetAsync(h->err_flag, 0, sizeof(int), (hipStream_t)stream));     // seen by the host: re-arm
    return flag & 3;
}

int keep_token_error_async(keep_handle* h, int32_t* host_flag, void* stream) {
    if (!h || !host_flag) return KEEP_EINVAL;
    KEEP_ON_DEVICE(h);
    HIPCHK(h, hipMemcpyAsync(host_flag, h->err_flag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    return KEEP_OK;
}

int keep_similarity(keep_handle* h, const float* img, const float* txt, int64_t N, int64_t P, int64_t D, float scale, int mode,
                    void* out, int32_t* argmax_out, void* stream) {
    if (!h) return KEEP_EINVAL;
    if (!img || !txt || N < 0 || P < 1 || D < 16 || D % 16) return h->fail(KEEP_EINVAL, "bad similarity arguments");
    if (mode < KEEP_SIM_RAW || mode > KEEP_SIM_TOP2SCORE) return h->fail(KEEP_EINVAL, "similarity mode %d", mode);
    if (mode == KEEP_SIM_ARGMAX && !argmax_out) return h->fail(KEEP_EINVAL, "argmax_out is null");
    if (mode != KEEP_SIM_ARGMAX && !out) return h->fail(KEEP_EINVAL, "out is null");
    if (N == 0) return KEEP_OK;
    KEEP_ON_DEVICE(h);
    return similarity_run(h, img, txt, N, P, D, scale, mode, out, argmax_out, (hipStream_t)stream);
}

/* keep_classify (include/keep_hip.h): labels with the accuracy of the split-product arithmetic at (nearly) the cost of the default one. */
int keep_classify(keep_handle* h, const void* pixels, int pix_dtype, int64_t B, const float* txt, int64_t P, float scale, float margin,
                  float* feats_out, float* sim_out, int32_t* labels_out, int64_t* n_rechecked, void* stream) {
    if (!h) return KEEP_EINVAL;
    if (n_rechecked) *n_rechecked = 0;
    if (!h->finalized || !h->vit_depth) return h->fail(KEEP_ESTATE, "image tower not loaded / finalised");
    if (!pixels || !txt || !labels_out || B < 0 || P < 1) return h->fail(KEEP_EINVAL, "null pointer or bad shape");
    if (pix_dtype < KEEP_PIX_F32 || pix_dtype > KEEP_PIX_U8_HWC) return h->fail(KEEP_EINVAL, "pixel dtype %d", pix_dtype);
    if (!(scale > 0.f)) return h->fail(KEEP_EINVAL, "classify needs scale > 0 (labels are the argmax of scale * cos)");
    if (((uintptr_t)pixels & 15) != 0) return h->fail(KEEP_EINVAL, "pixels must be 16-byte aligned");
    if (B == 0) return KEEP_OK;
    if (B > (1 << 30)) return h->fail(KEEP_EINVAL, "batch too large");
    KEEP_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    const int D = h->proj_dim;
    if (margin < 0.f) margin = h->label_margin;
    const size_t px = pix_dtype == KEEP_PIX_F32 ? 4 : (pix_dtype == KEEP_PIX_U8_HWC ? 1 : 2);
    const size_t tile_bytes = (size_t)3 * 224 * 224 * px;
    const int64_t stage_tiles = B < 256 ? B : 256;                      // flagged tiles are re-encoded in sub-batches of at most 256
    size_t total = 0;
    auto reserve = [&](size_t bytes) { const size_t at = total; total += align_up(bytes); return at; };
    const size_t o_feats = reserve((size_t)B * D * 4), o_sim = reserve((size_t)B * P * 4), o_flags = reserve((size_t)B * 4),
                 o_list = reserve((size_t)B * 4), o_count = reserve(256), o_stage = reserve((size_t)stage_tiles * tile_bytes),
                 o_f2 = reserve((size_t)stage_tiles * D * 4);
    if (total > h->cls_bytes) {
        HIPCHK(h, hipStreamSynchronize(s));
        if (h->cls_buf) HIPCHK(h, hipFree(h->cls_buf));
        h->cls_buf = nullptr; h->cls_bytes = 0;
        HIPCHK(h, hipMalloc(&h->cls_buf, total));
        h->cls_bytes = total;
    }
    char* b = h->cls_buf;
    float* feats = feats_out ? feats_out : (float*)(b + o_feats);
    float* sim = sim_out ? sim_out : (float*)(b + o_sim);
    int* list = (int*)(b + o_list);
    int rc = encode_image_run(h, pixels, pix_dtype, B, feats, s);
    if (rc) return rc;
    rc = similarity_run(h, feats, txt, B, P, D, scale, KEEP_SIM_ARGMAX, sim, labels_out, s);
    if (rc) return rc;
    if (margin == 0.f || P == 1 || h->precision == KEEP_PREC_STRICT) return check_launch(h, "classify");
    launch_top2_margin_flags(sim, (int)B, (int)P, margin * scale, (int*)(b + o_flags), list, (int*)(b + o_count), s);
    int count = 0;
    HIPCHK(h, hipMemcpyAsync(&count, b + o_count, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));                                 // the one host round trip of the call: how many tiles to look at again
    if (n_rechecked) *n_rechecked = count;
    if (count == 0) return check_launch(h, "classify");
    const int saved = h->precision;
    h->precision = KEEP_PREC_STRICT;                                    // per call, no option epoch: graph keys carry the precision
    // equal sub-batches (272 flagged tiles: 136 + 136, not 256 + 16 -- a 16-tile encode is latency-bound and costs a third of a 256-tile one)
    const int64_t parts = (count + stage_tiles - 1) / stage_tiles, per = (count + parts - 1) / parts;
    for (int64_t c0 = 0; c0 < count && !rc; c0 += per) {
        const int n = (int)((count - c0) < per ? (count - c0) : per);
        launch_gather_tiles(pixels, (int64_t)tile_bytes, list + c0, n, b + o_stage, s);
        rc = encode_image_run(h, b + o_stage, pix_dtype, n, (float*)(b + o_f2), s);
        if (!rc) launch_scatter_rows((const float*)(b + o_f2), list + c0, n, D, feats, s);
    }
    h->precision = saved;
    if (rc) return rc;
    rc = similarity_run(h, feats, txt, B, P, D, scale, KEEP_SIM_ARGMAX, sim, labels_out, s);
    if (rc) return rc;
    return check_launch(h, "classify");
}

int keep_prompt_scores(keep_handle* h, const float* feats, const float* bank, int64_t N, int64_t K, int64_t C, int64_t D,
                       float* scores_out, void* stream) {
    if (!h) return KEEP_EINVAL;
    if (!feats || !bank || !scores_out || N < 1 || K < 1 || C < 2 || D < 16 || D % 16) return h->fail(KEEP_EINVAL, "bad prompt_scores arguments");
    KEEP_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    Scope sc(h, T_SIM, s);
    const int64_t KC = K * C;
    if ((C == 2 || C == 4) && D % 128 == 0 && D >= 256 && h->fused_screening) {
        // Fused path: ONE compensated GEMM [N,D] x [D,K*C] whose epilogue takes the per-(tile, classifier)
        // top-2 score in the accumulator registers and sums it over the tile's rows; no logit reaches HBM.
        const int64_t KCp = (KC + 255) / 256 * 256, nslots = (N + 255) / 256 * 2, kpad = KCp / C;
        const bool comp = h->fused_screening == 1;                      // 1: fp16 + MX-fp4 corrections; 2: three fp16 passes
        size_t total = 0;
        auto reserve = [&](size_t bytes) { const size_t at = total; total += align_up(bytes); return at; };     // offsets first: the arena may move
        const size_t o_ahi = reserve(blk_elems(N, D) * sizeof(f16));
        const size_t o_alo = reserve(comp ? 0 : blk_elems(N, D) * sizeof(f16));
        const size_t o_aq = reserve(keepk::q4_data_bytes(N, D));
        const size_t o_asc = reserve(keepk::q4_scale_bytes(N, D));
        const size_t o_whi = reserve(blk_elems(KCp, D) * sizeof(f16));
        const size_t o_wlo = reserve(blk_elems(KCp, D) * sizeof(f16));
        const size_t o_wq = reserve(keepk::q4_data_bytes(KCp, D));
        const size_t o_wsc = reserve(keepk::q4_scale_bytes(KCp, D));
        const size_t o_part = reserve((size_t)nslots * kpad * sizeof(float));
        int rc = ensure_arena(h, total);
        if (rc) return rc;
        char* a = h->arena;
        launch_quant_blockify(feats, (f16*)(a + o_ahi), comp ? nullptr : (f16*)(a + o_alo), (unsigned char*)(a + o_aq), (unsigned char*)(a + o_asc), (int)N, (int)D, s);
        launch_quant_blockify(bank, (f16*)(a + o_whi), (f16*)(a + o_wlo), (unsigned char*)(a + o_wq), (unsigned char*)(a + o_wsc), (int)KC, (int)D, s);
        // (rows KC..KCp of the bank planes are zero-filled by the blockify kernel: their scores land beyond K and are never read)
        GemmParams p{};
        p.tune = &h->tune;
        p.a_hi = (f16*)(a + o_ahi); p.a_lo = (f16*)(a + o_alo); p.w_hi = (f16*)(a + o_whi); p.w_lo = (f16*)(a + o_wlo);
        p.M = (int)N; p.N = (int)KCp; p.K = (int)D; p.nseg = comp ? 1 : 3; p.patches_per_img = 196;
        if (comp) { p.comp = 2; p.a_q = (unsigned char*)(a + o_aq); p.a_sc = (unsigned char*)(a + o_asc); p.w_q = (unsigned char*)(a + o_wq); p.w_sc = (unsigned char*)(a + o_wsc); }
        p.top2_c = (int)C; p.top2_partial = (float*)(a + o_part); p.top2_kpad = (int)kpad;
        h->prof_add_flops(T_SIM, 2.0 * N * (double)KC * D);
        if (launch_gemm_f16(p, EPI_TOP2, s) < 0) return h->fail(KEEP_EUNSUPPORTED, "fused prompt screening launch failed");
        launch_top2_slots_reduce((float*)(a + o_part), (int)nslots, (int)kpad, (int)K, 1.0f / (float)N, scores_out, s);
        return check_launch(h, "prompt_scores");
    }
    int64_t chunk = ((int64_t)256 << 20) / (KC * 4);          // <= 256 MiB of logits alive at a time
    if (chunk < 256) chunk = 256;
    if (chunk > N) chunk = N;
    const int max_rb = 64;
    const size_t b_logits = align_up((size_t)chunk * KC * 4), b_part = align_up((size_t)max_rb * K * 4), b_sums = align_up((size_t)K * 4);
    int rc = ensure_arena(h, b_logits + b_part + b_sums);
    if (rc) return rc;
    float* logits = (float*)h->arena;
    float* partial = (float*)(h->arena + b_logits);
    float* sums = (float*)(h->arena + b_logits + b_part);
    HIPCHK(h, hipMemsetAsync(sums, 0, (size_t)K * 4, s));
    for (int64_t r0 = 0; r0 < N; r0 += chunk) {
        const int64_t n = (N - r0) < chunk ? (N - r0) : chunk;
        SgemmParams g{};
        g.tune = &h->tune;
        g.a = feats + r0 * D; g.lda = D; g.b = bank; g.ldb = D; g.out = logits; g.ldo = KC; g.bias = nullptr;
        g.M = (int)n; g.N = (int)KC; g.K = (int)D; g.scale = 1.0f; g.act = ACT_NONE;
        if (launch_sgemm_f32(g, s)) return h->fail(KEEP_EUNSUPPORTED, "prompt_scores shape");
        launch_group_top2(logits, (int)n, (int)K, (int)C, partial, max_rb, sums, s);
    }
    launch_scale_vec(sums, (int)K, 1.0f / (float)N, scores_out, s);
    return check_launch(h, "prompt_scores");
}

int keep_group_argmax(keep_handle* h, const float* feats, const float* bank, int64_t N, int64_t K, int64_t C, int64_t D,
                      int32_t* labels_out, void* stream) {
    if (!h) return KEEP_EINVAL;
    if (!feats || !bank || !labels_out || N < 1 || K < 1 || C < 1 || D < 16 || D % 16) return h->fail(KEEP_EINVAL, "bad group_argmax arguments");
    KEEP_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    Scope sc(h, T_SIM, s);
    const int64_t KC = K * C;
    int64_t chunk = ((int64_t)256 << 20) / (KC * 4);
    if (chunk < 256) chunk = 256;
    if (chunk > N) chunk = N;
    int rc = ensure_arena(h, align_up((size_t)chunk * KC * 4));
    if (rc) return rc;
    float* logits = (float*)h->arena;
    for (int64_t r0 = 0; r0 < N; r0 += chunk) {
        const int64_t n = (N - r0) < chunk ? (N - r0) : chunk;
        SgemmParams g{};
        g.tune = &h->tune;
        g.a = feats + r0 * D; g.lda = D; g.b = bank; g.ldb = D; g.out = logits; g.ldo = KC; g.bias = nullptr;
        g.M = (int)n; g.N = (int)KC; g.K = (int)D; g.scale = 1.0f; g.act = ACT_NONE;
        if (launch_sgemm_f32(g, s)) return h->fail(KEEP_EUNSUPPORTED, "group_argmax shape");
        // [n][K][C] is contiguous: one argmax per (tile, round) row of C scores, first maximum wins (numpy.argmax)
        launch_row_argmax(logits, (int)(n * K), (int)C, labels_out + r0 * K, s);
    }
    return check_launch(h, "group_argmax");
}

int keep_retrieval_rank(keep_handle* h, const float* txt, const float* img, int64_t P, int64_t N, int64_t D,
                        const int32_t* target, int32_t* rank_out, void* stream) {
    if (!h) return KEEP_EINVAL;
    if (!txt || !img || !rank_out || P < 1 || N < 1 || D < 16 || D % 16 || (!target && P > N)) return h->fail(KEEP_EINVAL, "bad retrieval_rank arguments");
    KEEP_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    Scope sc(h, T_SIM, s);
    int64_t chunk = ((int64_t)256 << 20) / (N * 4);
    if (chunk < 64) chunk = 64;
    if (chunk > P) chunk = P;
    int rc = ensure_arena(h, align_up((size_t)chunk * N * 4));
    if (rc) return rc;
    float* sim = (float*)h->arena;
    for (int64_t r0 = 0; r0 < P; r0 += chunk) {
        const int64_t n = (P - r0) < chunk ? (P - r0) : chunk;
        SgemmParams g{};
        g.tune = &h->tune;
        g.a = txt + r0 * D; g.lda = D; g.b = img; g.ldb = D; g.out = sim; g.ldo = N; g.bias = nullptr;
        g.M = (int)n; g.N = (int)N; g.K = (int)D; g.scale = 1.0f; g.act = ACT_NONE;
        if (launch_sgemm_f32(g, s)) return h->fail(KEEP_EUNSUPPORTED, "retrieval_rank shape");
        launch_diag_rank(sim, (int)n, (int)N, target, (int)r0, rank_out, s);
    }
    return check_launch(h, "retrieval_rank");
}

int keep_refine(keep_handle* h, const float* probs, const int64_t* coords, int64_t N, int64_t C, int64_t patch, int overlap,
                float* out_mean, int32_t* is_first, void* stream) {
    if (!h) return KEEP_EINVAL;
    if (!probs || !coords || !out_mean || !is_first || N < 1 || C < 1 || N > (1 << 29)) return h->fail(KEEP_EINVAL, "bad refine arguments");
    KEEP_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    unsigned size = 1024;
    while ((int64_t)size < 2 * N) size <<= 1;
    const size_t b_keys = align_up((size_t)size * 8), b_first = align_up((size_t)size * 4);
    int rc = ensure_arena(h, b_keys + b_first);
    if (rc) return rc;
    launch_refine(probs, (const long long*)coords, (int)N, (int)C, (long long)patch, overlap,
                  (unsigned long long*)h->arena, (int*)(h->arena + b_keys), size, out_mean, (int*)is_first, s);
    return check_launch(h, "refine");
}

int keep_profile_enable(keep_handle* h, const char* tag) {
    if (!h) return KEEP_EINVAL;
    if (!tag) { h->prof_mode = 2; return KEEP_OK; }
    if (!*tag) { h->prof_mode = 0; return KEEP_OK; }
    // one tag, or several separated by commas ("vit.proj,vit.fc2")
    unsigned long long mask = 0;
    std::string names(tag);
    size_t a = 0;
    while (a <= names.size()) {
        const size_t b = names.find(',', a);
        const std::string one = names.substr(a, b == std::string::npos ? std::string::npos : b - a);
        const int t = tag_by_name(one.c_str());
        if (t < 0) return h->fail(KEEP_EINVAL, "unknown profile tag %s", one.c_str());
        mask |= 1ull << t;
        if (b == std::string::npos) break;
        a = b + 1;
    }
    h->prof_mode = 1; h->prof_mask = mask;
    return KEEP_OK;
}
int keep_profile_read(keep_handle* h, const char* tag, double* total_ms, int64_t* launches, double* flops) {
    if (!h || !tag) return KEEP_EINVAL;
    const int t = tag_by_name(tag);
    if (t < 0) return h->fail(KEEP_EINVAL, "unknown profile tag %s", tag);
    DevGuard guard(h->device);
    h->prof_collect();
    if (total_ms) *total_ms = h->prof_ms[t];
    if (launches) *launches = h->prof_n[t];
    if (flops) *flops = h->prof_flops[t];
    return KEEP_OK;
}
int keep_profile_reset(keep_handle* h) {
    if (!h) return KEEP_EINVAL;
    DevGuard guard(h->device);
    h->prof_collect();
    for (int i = 0; i < T_COUNT; ++i) { h->prof_ms[i] = 0; h->prof_n[i] = 0; h->prof_flops[i] = 0; }
    return KEEP_OK;
}

// ---------------------------------------------------------------- single-operator entry points
int keep_op_linear(keep_handle* h, const float* a, const float* w, const float* bias, const float* ls, const float* resid,
                   int64_t M, int64_t N, int64_t K, int epi, int split, float* out, void* stream) {
    if (!h || !a || !w || !bias || !out) return h ? h->fail(KEEP_EINVAL, "null pointer") : KEEP_EINVAL;
    if (M < 1 || N % 128 || N < 128 || K < 64 || K % 32) return h->fail(KEEP_EUNSUPPORTED, "linear needs N%%128==0 and K%%32==0");
    if (epi != EPI_F16 && epi != EPI_GELU_F16 && epi != EPI_RESID_LS && epi != EPI_RESID_F32) return h->fail(KEEP_EINVAL, "epilogue %d", epi);
    if ((epi == EPI_RESID_LS && (!ls || !resid)) || (epi == EPI_RESID_F32 && !resid)) return h->fail(KEEP_EINVAL, "missing ls/resid");
    KEEP_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    Tmp t;
    const size_t ae = blk_elems(M, K), we = blk_elems(N, K), oe = blk_elems(M, N);
    f16* a_hi = t.get<f16>(ae); f16* a_lo = t.get<f16>(ae);
    f16* w_hi = t.get<f16>(we); f16* w_lo = t.get<f16>(we);
    f16* o_hi = t.get<f16>(oe); f16* o_lo = t.get<f16>(oe);
    if (!a_hi || !a_lo || !w_hi || !w_lo || !o_hi || !o_lo) return h->fail(KEEP_ENOMEM, "temp alloc");
    const bool comp = split == 2 || split == 3;          // 3: the W_lo term only (K >= 512)
    unsigned char *a_q = nullptr, *a_sc = nullptr, *w_q = nullptr, *w_sc = nullptr;
    if (comp) {
        if (N % 256 || K % 128 || K < (split == 3 ? 512 : 256) || epi == EPI_RESID_F32) return h->fail(KEEP_EUNSUPPORTED, "compensated linear needs N%%256==0, K%%128==0, K>=256 (512 for the one-term form) and epilogue 0/1/2");
        a_q = t.get<unsigned char>(keepk::q4_data_bytes(M, K)); a_sc = t.get<unsigned char>(keepk::q4_scale_bytes(M, K));
        w_q = t.get<unsigned char>(keepk::q4_data_bytes(N, K)); w_sc = t.get<unsigned char>(keepk::q4_scale_bytes(N, K));
        if (!a_q || !a_sc || !w_q || !w_sc) return h->fail(KEEP_ENOMEM, "temp alloc");
        launch_quant_blockify(a, a_hi, a_lo, a_q, a_sc, (int)M, (int)K, s); launch_quant_blockify(w, w_hi, w_lo, w_q, w_sc, (int)N, (int)K, s);
    }
    else { launch_split_blockify(a, a_hi, a_lo, (int)M, (int)K, s); launch_split_blockify(w, w_hi, w_lo, (int)N, (int)K, s); }
    GemmParams p{};
    p.tune = &h->tune;
    p.a_hi = a_hi; p.a_lo = a_lo; p.w_hi = w_hi; p.w_lo = w_lo; p.M = (int)M; p.N = (int)N; p.K = (int)K;
    p.nseg = (split == 1) ? 3 : 1; p.bias = bias; p.ls = ls; p.patches_per_img = 196;
    if (comp) { p.comp = split == 3 ? 1 : 2; p.a_q = a_q; p.a_sc = a_sc; p.w_q = w_q; p.w_sc = w_sc; }
    p.splitk_ws = t.get<float>(SKINNY_WS_BYTES / 4); p.splitk_bytes = SKINNY_WS_BYTES;      // auto mode may take a split-K path (small or mid-size M), as the towers do
    if (!p.splitk_ws) return h->fail(KEEP_ENOMEM, "temp alloc");
    int launch_rc = 0;
    auto launch = [&](const GemmParams& q) { launch_rc = launch_gemm_f16(q, epi, s); };
    if (epi == EPI_F16 || epi == EPI_GELU_F16) {
        p.out_hi = o_hi; p.out_lo = (split == 1 || split == 2) ? o_lo : nullptr;
        // as in the towers: the GELU output feeds another GEMM (blk layout), the plain one feeds attention (row-major)
        p.out_kt = (epi == EPI_GELU_F16) ? (int)(N / 32) : 0;
        launch(p);
        if (p.out_kt) launch_unblockify_f32(o_hi, p.out_lo, out, (int)M, (int)N, s);
        else planes_to_f32(o_hi, p.out_lo, out, M * N, s);
    } else if (epi == EPI_RESID_LS) {
        HIPCHK(h, hipMemcpyAsync(out, resid, M * N * sizeof(float), hipMemcpyDeviceToDevice, s));
        p.resid = out;
        launch(p);
    } else {
        p.resid = const_cast<float*>(resid); p.out_f32 = out;
        launch(p);
    }
    HIPCHK(h, hipStreamSynchronize(s));
    if (launch_rc < 0) return h->fail(KEEP_EUNSUPPORTED, "op_linear: no kernel for this shape / mode");
    return check_launch(h, "op_linear");
}

int keep_op_mlp(keep_handle* h, const float* x, const float* ln_w, const float* ln_b, const float* fc1_w, const float* fc1_b,
                const float* fc2_w, const float* fc2_b, const float* ls, int64_t M, int64_t D, int64_t F, int mode, float* out, void* stream) {
    if (!h || !x || !ln_w || !ln_b || !fc1_w || !fc1_b || !fc2_w || !fc2_b || !ls || !out) return h ? h->fail(KEEP_EINVAL, "null pointer") : KEEP_EINVAL;
    if (M < 1 || (D != 768 && D != 1024) || F % 256 || F < 256 || mode < 0 || mode > 3) return h->fail(KEEP_EUNSUPPORTED, "op_mlp: D in {768, 1024}, F %% 256 == 0, mode 0..3");
    KEEP_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    Tmp t;
    const bool lo = mode == 1, q = mode == 2 || mode == 3;
    const int cmode = mode == 3 ? 1 : 2;                 // GemmParams.comp: the W_lo term only | both terms
    f16 *w1h = t.get<f16>(F * D), *w1l = t.get<f16>(F * D), *w2h = t.get<f16>(D * F), *w2l = t.get<f16>(D * F);
    f16 *xh = t.get<f16>(blk_elems(M, D)), *xl = t.get<f16>(blk_elems(M, D)), *mh = t.get<f16>(blk_elems(M, F)), *ml = t.get<f16>(blk_elems(M, F));
    unsigned char *w1q = t.get<unsigned char>(keepk::q4_data_bytes(F, D)), *w1s = t.get<unsigned char>(keepk::q4_scale_bytes(F, D));
    unsigned char *w2q = t.get<unsigned char>(keepk::q4_data_bytes(D, F)), *w2s = t.get<unsigned char>(keepk::q4_scale_bytes(D, F));
    unsigned char *xq = t.get<unsigned char>(keepk::q4_data_bytes(M, D)), *xs = t.get<unsigned char>(keepk::q4_scale_bytes(M, D));
    unsigned char *mq = t.get<unsigned char>(keepk::q4_data_bytes(M, F)), *ms = t.get<unsigned char>(keepk::q4_scale_bytes(M, F));
    float* ws = t.get<float>(SKINNY_WS_BYTES / 4);
    if (!w1h || !w1l || !w2h || !w2l || !xh || !xl || !mh || !ml || !w1q || !w1s || !w2q || !w2s || !xq || !xs || !mq || !ms || !ws) return h->fail(KEEP_ENOMEM, "temp alloc");
    launch_quant_blockify(fc1_w, w1h, w1l, w1q, w1s, (int)F, (int)D, s);
    launch_quant_blockify(fc2_w, w2h, w2l, w2q, w2s, (int)D, (int)F, s);
    HIPCHK(h, hipMemcpyAsync(out, x, M * D * sizeof(float), hipMemcpyDeviceToDevice, s));
    LnParams ln{};
    ln.tune = &h->tune;
    ln.x = x; ln.x_stride = D; ln.rows = (int)M; ln.D = (int)D; ln.eps = 1e-6f; ln.gamma = ln_w; ln.beta = ln_b;
    ln.out_hi = xh; ln.out_lo = lo ? xl : nullptr; ln.out_kt = (int)(D / 32); ln.out_q = q ? xq : nullptr; ln.out_sc = q ? xs : nullptr; ln.out_q_hi_only = mode == 3;
    if (launch_layernorm(ln, s)) return h->fail(KEEP_EUNSUPPORTED, "op_mlp: layernorm");
    GemmParams p{};
    p.tune = &h->tune; p.patches_per_img = 196; p.splitk_ws = ws; p.splitk_bytes = SKINNY_WS_BYTES;
    p.a_hi = xh; p.a_lo = xl; p.w_hi = w1h; p.w_lo = w1l; p.M = (int)M; p.N = (int)F; p.K = (int)D; p.nseg = lo ? 3 : 1; p.bias = fc1_b;
    p.out_hi = mh; p.out_lo = lo ? ml : nullptr; p.out_kt = (int)(F / 32);
    if (q) { p.comp = cmode; p.a_q = xq; p.a_sc = xs; p.w_q = w1q; p.w_sc = w1s; p.out_q = mq; p.out_sc = ms; }
    if (launch_gemm_f16(p, EPI_GELU_F16, s) < 0) return h->fail(KEEP_EUNSUPPORTED, "op_mlp: fc1");
    GemmParams r{};
    r.tune = &h->tune; r.patches_per_img = 196; r.splitk_ws = ws; r.splitk_bytes = SKINNY_WS_BYTES;
    r.a_hi = mh; r.a_lo = ml; r.w_hi = w2h; r.w_lo = w2l; r.M = (int)M; r.N = (int)D; r.K = (int)F; r.nseg = lo ? 3 : 1; r.bias = fc2_b;
    r.ls = ls; r.resid = out;
    if (q) { r.comp = cmode; r.a_q = mq; r.a_sc = ms; r.w_q = w2q; r.w_sc = w2s; }
    if (launch_gemm_f16(r, EPI_RESID_LS, s) < 0) return h->fail(KEEP_EUNSUPPORTED, "op_mlp: fc2");
    HIPCHK(h, hipStreamSynchronize(s));
    return check_launch(h, "op_mlp");
}

int keep_op_attention(keep_handle* h, const float* qkv, const int64_t* mask, int64_t B, int64_t T, int heads, int split,
                      float* out, void* stream) {
    if (!h || !qkv || !out || B < 1 || T < 1 || heads < 1) return h ? h->fail(KEEP_EINVAL, "bad attention arguments") : KEEP_EINVAL;
    KEEP_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    const int64_t M = B * T, D = (int64_t)heads * 64;
    Tmp t;
    f16* q_hi = t.get<f16>(M * 3 * D); f16* q_lo = t.get<f16>(M * 3 * D);
    // the towers write the attention output in blk layout when the width allows it (D % 32 == 0 always holds)
    const size_t oe = blk_elems(M, D);
    f16* o_hi = t.get<f16>(oe); f16* o_lo = t.get<f16>(oe);
    if (!q_hi || !q_lo || !o_hi || !o_lo) return h->fail(KEEP_ENOMEM, "temp alloc");
    launch_split_f16(qkv, q_hi, q_lo, M * 3 * D, s);
    AttnParams a{};
    a.tune = &h->tune;
    if (split && T > 256) {
        a.part_bytes = (size_t)B * heads * T * ATT_PART_FLOATS * sizeof(float);
        a.part_ws = t.get<float>(a.part_bytes / sizeof(float));
        if (!a.part_ws) return h->fail(KEEP_ENOMEM, "temp alloc");
    }
    a.qkv_hi = q_hi; a.qkv_lo = q_lo; a.out_hi = o_hi; a.out_lo = split ? o_lo : nullptr; a.mask = mask;
    a.batch = (int)B; a.ntok = (int)T; a.heads = heads; a.split = split; a.scale = 0.125f; a.out_kt = (int)(D / 32);
    if (launch_attention(a, s)) return h->fail(KEEP_EUNSUPPORTED, "sequence length %lld unsupported", (long long)T);
    launch_unblockify_f32(o_hi, split ? o_lo : nullptr, out, (int)M, (int)D, s);
    HIPCHK(h, hipStreamSynchronize(s));
    return check_launch(h, "op_attention");
}

// Matrix-pipe ceiling probe (keep_mfma_probe): no memory traffic inside the loop; every wave holds 2 A and 4 B fragments of the caller's data in registers and
// issues 8 independent v_mfma_f32_32x32x16_f16 per iteration (all (i, j) pairs: the pipe's inputs change with every instruction); one 8-wave workgroup per CU,
// two waves per SIMD, as the GEMM kernels run.
__global__ __launch_bounds__(512, 2) void mfma_probe_kernel(const f16x8* __restrict__ src, float* __restrict__ sink, int iters) {
    f16x8 a[2], b[4];
    const int t = blockIdx.x * 512 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = src[(size_t)t * 6 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = src[(size_t)t * 6 + 2 + i];
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    }
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) v += acc[i][r];
    sink[t] = v;
}

int keep_op_layernorm(keep_handle* h, const float* x, const float* add, const float* gamma, const float* beta, int64_t rows,
                      int64_t D, float eps, float* out, void* stream) {
    if (!h || !x || !gamma || !beta || !out || rows < 1) return h ? h->fail(KEEP_EINVAL, "bad layernorm arguments") : KEEP_EINVAL;
    KEEP_ON_DEVICE(h);
    LnParams p{};
    p.tune = &h->tune;
    p.x = x; p.x_stride = D; p.add = add; p.gamma = gamma; p.beta = beta; p.rows = (int)rows; p.D = (int)D; p.eps = eps;
    p.out_f32 = out; p.out_f32_stride = D;
    if (launch_layernorm(p, (hipStream_t)stream)) return h->fail(KEEP_EUNSUPPORTED, "layernorm width %lld", (long long)D);
    return check_launch(h, "op_layernorm");
}

int keep_op_sgemm(keep_handle* h, const float* a, const float* b, const float* bias, int64_t M, int64_t N, int64_t K, float scale,
                  int act, float* out, void* stream) {
    if (!h || !a || !b || !out) return h ? h->fail(KEEP_EINVAL, "null pointer") : KEEP_EINVAL;
    KEEP_ON_DEVICE(h);
    SgemmParams g{};
    g.tune = &h->tune;
    g.a = a; g.lda = K; g.b = b; g.ldb = K; g.out = out; g.ldo = N; g.bias = bias; g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.scale = scale; g.act = act;
    if (launch_sgemm_f32(g, (hipStream_t)stream)) return h->fail(KEEP_EUNSUPPORTED, "sgemm needs K%%16==0");
    return check_launch(h, "op_sgemm");
}

int keep_debug_read(keep_handle* h, void* host_dst, int64_t bytes) {
    if (!h || !host_dst || !h->tune.dbg || bytes > (int64_t)65536 * 4 * 8) return KEEP_EINVAL;      // diagnostics builds only
    KEEP_ON_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(host_dst, h->tune.dbg, bytes, hipMemcpyDeviceToHost));
    return KEEP_OK;
}

int keep_mfma_probe(keep_handle* h, const void* operands_f16, float* sink, int iters, double* flops_out, void* stream) {
    if (!h || !operands_f16 || !sink || iters < 1 || iters > (1 << 24)) return h ? h->fail(KEEP_EINVAL, "bad mfma_probe arguments") : KEEP_EINVAL;
    KEEP_ON_DEVICE(h);
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || cus < 1) cus = 256;
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(cus), dim3(512), 0, (hipStream_t)stream, (const f16x8*)operands_f16, sink, iters);
    if (flops_out) *flops_out = (double)cus * 8.0 * iters * 8.0 * 2.0 * 32 * 32 * 16;
    return check_launch(h, "mfma_probe");
}

int keep_clock_probe(keep_handle* h, int spin_us, long long* device_out2, void* stream) {
    if (!h || !device_out2 || spin_us < 1 || spin_us > 100000) return h ? h->fail(KEEP_EINVAL, "bad clock_probe arguments") : KEEP_EINVAL;
    KEEP_ON_DEVICE(h);
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, device_out2, spin_us * 100);
    return check_launch(h, "clock_probe");
}

int keep_op_l2norm(keep_handle* h, float* x, int64_t rows, int64_t D, void* stream) {
    if (!h || !x || rows < 1 || D < 1) return h ? h->fail(KEEP_EINVAL, "bad l2norm arguments") : KEEP_EINVAL;
    KEEP_ON_DEVICE(h);
    launch_l2norm_rows(x, (int)rows, (int)D, 1e-12f, (hipStream_t)stream);
    return check_launch(h, "op_l2norm");
}

}  // extern "C"
