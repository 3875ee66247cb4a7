"""encode_image / encode_text / similarity on the MI355X against the CPU oracle and the golden
vectors (HF BertModel / Dinov2-as-ViT-L outputs committed under tests/golden/).

Tolerances (BASELINE.json north_star): cosine similarities within 1e-4 of the fp32 reference,
argmax labels identical.
  * 'comp' -- the DEFAULT precision and the one bench.py reports -- is held to COS_TOL = 1e-4 everywhere: fp16 MFMA pass
    plus first-order correction terms where the error budget needs them (DESIGN.md "Precision").
  * 'strict' (hi/lo split operands, 3 MFMA passes) is held to a few 1e-6.
  * 'fp16' is the opt-in single-pass mode: every GEMM operand rounded to 11 bits gives sigma(dcos) ~ 3.3e-5 on these
    synthetic weights, i.e. a worst case of 1.0-1.5e-4 -- OUTSIDE the tolerance, which is why it is not the default.  It is
    only checked against its own documented budget FP16_TOL = 2.5e-4 (regression guard), never presented as compliant.
"""
import os

import numpy as np
import pytest
import torch

from keep_amd import KEEPModel
from keep_amd.config import KEEPShape, small_shape
from keep_amd.synth import synth_prompts, synth_state_dict, synth_tiles, towers_of
from oracle import keep_oracle as O

pytestmark = pytest.mark.gpu
COS_TOL = 1e-4
FP16_TOL = 2.5e-4
MODES = ["strict", "comp", "fp16"]


def tol(precision, strict_tol=2e-6):
    return {"strict": strict_tol, "comp": COS_TOL, "fp16": FP16_TOL}[precision]


_EXTRA_OPTS = {}        # engine options every model of a test gets (options belong to a handle, not to the process)


def make_model(sd, precision):
    m = KEEPModel(precision=precision, towers=towers_of(sd))
    for k, v in _EXTRA_OPTS.items():
        m.set_option(k, v)
    m.load_state_dict(sd, strict=True)
    return m.to("cuda:0").eval()


@pytest.fixture(scope="module")
def small():
    shape = small_shape(2, 2)
    sd = synth_state_dict(shape, seed=5)
    return sd


@pytest.fixture(scope="module")
def text_bank():
    g = torch.Generator().manual_seed(99)
    return torch.nn.functional.normalize(torch.randn(64, 768, generator=g), dim=-1)


@pytest.mark.parametrize("precision", MODES)
def test_encode_image_depth2_vs_oracle(small, text_bank, precision):
    x = synth_tiles(5, seed=3)
    with torch.no_grad():
        ref = O.encode_image(small, x)
    m = make_model(small, precision)
    out = m.encode_image(x)                      # CPU in -> CPU out, like the reference
    assert out.device.type == "cpu" and out.dtype == torch.float32 and out.shape == (5, 768)
    assert torch.allclose(out.norm(dim=-1), torch.ones(5), atol=1e-5)
    dcos = (out @ text_bank.t() - ref @ text_bank.t()).abs().max().item()
    print(f"[vit d2 {precision}] max|dfeat|={(out - ref).abs().max():.3e} max|dcos|={dcos:.3e}")
    assert dcos < tol(precision)
    assert torch.equal((out @ text_bank.t()).argmax(1), (ref @ text_bank.t()).argmax(1))
    # bf16 / fp16 pixel inputs (BASELINE config 2 feeds bf16 tiles)
    for dt in (torch.bfloat16, torch.float16):
        xd = x.to(dt)
        with torch.no_grad():
            ref_d = O.encode_image(small, xd.float())
        out_d = m.encode_image(xd.cuda()).cpu()
        assert (out_d @ text_bank.t() - ref_d @ text_bank.t()).abs().max() < tol(precision)


@pytest.mark.parametrize("precision", MODES)
def test_encode_text_2layers_vs_oracle(small, text_bank, precision):
    toks = synth_prompts(6, 256, seed=4)
    toks["attention_mask"][0, :] = 1
    with torch.no_grad():
        ref = O.encode_text(small, toks)
    m = make_model(small, precision)
    out = m.encode_text(toks)
    assert out.shape == (6, 768) and out.device.type == "cpu"
    dcos = (out @ text_bank.t() - ref @ text_bank.t()).abs().max().item()
    print(f"[bert l2 {precision}] max|dfeat|={(out - ref).abs().max():.3e} max|dcos|={dcos:.3e}")
    assert dcos < tol(precision)
    # HF defaults: no token_type_ids / attention_mask given
    out2 = m.encode_text({"input_ids": toks["input_ids"][:2]})
    with torch.no_grad():
        ref2 = O.encode_text(small, {"input_ids": toks["input_ids"][:2]})
    assert (out2 - ref2).abs().max() < (5e-6 if precision == "strict" else 2e-3)
    # shorter sequences
    for T in (64, 40):
        t = {k: v[:3, :T].contiguous() for k, v in toks.items()}
        with torch.no_grad():
            r = O.encode_text(small, t)
        o = m.encode_text(t)
        assert (o @ text_bank.t() - r @ text_bank.t()).abs().max() < tol(precision)


def test_padding_trim_is_invisible(small, text_bank):
    """encode_text runs at the longest valid length (2.7 of 45.9 GFLOP at length 16); the result
    must equal the padded computation, and a batch holding a row with NO valid token (HF: uniform attention over all
    T keys) must not be trimmed."""
    m = make_model(small, "strict")
    toks = synth_prompts(5, 256, seed=21)                  # valid lengths 8..32
    with torch.no_grad():
        ref = O.encode_text(small, toks)
    out = m.encode_text(toks)
    assert m.last_text_length == 32 or m.last_text_length == 16
    m.trim_padding = False
    full = m.encode_text(toks)
    assert m.last_text_length == 256
    assert (out - full).abs().max() < 1e-6 and (out - ref).abs().max() < 5e-6
    m.trim_padding = True
    toks["attention_mask"][2, :] = 0                        # fully masked row
    with torch.no_grad():
        ref0 = O.encode_text(small, toks)
    out0 = m.encode_text(toks)
    assert m.last_text_length == 256 and (out0 - ref0).abs().max() < 5e-6
    toks["attention_mask"][2, 200] = 1                      # a hole-y mask: only trailing all-padding columns go
    with torch.no_grad():
        ref1 = O.encode_text(small, toks)
    out1 = m.encode_text(toks)
    assert m.last_text_length == 208 and (out1 - ref1).abs().max() < 5e-6


def test_long_and_single_inputs(small, text_bank):
    """T = 512 (max_position_embeddings; 32-tile attention kernel), one tile, one prompt."""
    m = make_model(small, "fp16")
    toks = synth_prompts(2, 512, seed=17, max_len=400)
    toks["attention_mask"][1, :] = 1
    with torch.no_grad():
        ref = O.encode_text(small, toks)
    out = m.encode_text(toks)
    assert (out @ text_bank.t() - ref @ text_bank.t()).abs().max() < FP16_TOL
    one = {k: v[:1, :300].contiguous() for k, v in toks.items()}
    with torch.no_grad():
        ref1 = O.encode_text(small, one)
    assert (m.encode_text(one) @ text_bank.t() - ref1 @ text_bank.t()).abs().max() < FP16_TOL
    x = synth_tiles(1, seed=18)
    with torch.no_grad():
        refi = O.encode_image(small, x)
    assert (m.encode_image(x) @ text_bank.t() - refi @ text_bank.t()).abs().max() < FP16_TOL
    # 256 < T <= 512 in the split-product modes: the split attention runs over two key windows of <= 256 keys and merges them (online softmax),
    # so the default mode holds the 1e-4 tolerance and strict its 5e-6 at every length BertModel accepts (keep_inference.py:60-62)
    ms = make_model(small, "strict")
    assert (ms.encode_text(toks) @ text_bank.t() - ref @ text_bank.t()).abs().max() < 5e-6
    assert (ms.encode_text(one) @ text_bank.t() - ref1 @ text_bank.t()).abs().max() < 5e-6
    mc = make_model(small, "comp")
    d512 = (mc.encode_text(toks) @ text_bank.t() - ref @ text_bank.t()).abs().max().item()
    d300 = (mc.encode_text(one) @ text_bank.t() - ref1 @ text_bank.t()).abs().max().item()
    print(f"[long text] comp: T=512 max|dcos| {d512:.3e}, T=300 (one prompt) {d300:.3e}")
    assert d512 < COS_TOL and d300 < COS_TOL
    # ragged batch above 256 tokens: key-padding masks that end inside the first window, inside the second, and a fully padded second window
    rag = synth_prompts(4, 384, seed=19, max_len=380)
    rag["attention_mask"][0, :] = 1
    rag["attention_mask"][1, 200:] = 0
    rag["attention_mask"][2, 257:] = 0
    with torch.no_grad():
        refr = O.encode_text(small, rag)
    assert (ms.encode_text(rag) @ text_bank.t() - refr @ text_bank.t()).abs().max() < 5e-6
    assert (mc.encode_text(rag) @ text_bank.t() - refr @ text_bank.t()).abs().max() < COS_TOL
    too_long = synth_prompts(1, 513, seed=20)
    too_long["attention_mask"][:] = 1                        # (otherwise the padding trim shortens the call)
    with pytest.raises(ValueError):
        ms.encode_text(too_long)


def test_forward_and_errors(small):
    m = make_model(small, "fp16")
    x, toks = synth_tiles(2, seed=1), synth_prompts(3, 256, seed=2)
    out = m.forward(x, toks)
    assert set(out) == {"vision_features", "text_features"}
    assert out["vision_features"].shape == (2, 768) and out["text_features"].shape == (3, 768)
    assert m.eval() is m and m.to("cuda:0") is m
    assert abs(float(m.logit_scale) - np.log(1 / 0.04)) < 1e-6
    with pytest.raises(ValueError):
        m.encode_image(torch.zeros(1, 3, 256, 256))
    with pytest.raises(ValueError):
        m.encode_image(torch.zeros(3, 224, 224))
    bad = {k: v.clone() for k, v in toks.items()}
    bad["input_ids"][0, 3] = 40000
    with pytest.raises(IndexError):
        m.encode_text(bad)
    with pytest.raises(ValueError):
        m.encode_text({"input_ids": torch.zeros(1, 600, dtype=torch.int64)})
    assert m.encode_image(torch.zeros(0, 3, 224, 224)).shape == (0, 768)


def test_strict_state_dict_semantics(small):
    m = KEEPModel().to("cuda:0")
    missing = {k: v for k, v in small.items() if k != "visual.blocks.1.mlp.fc2.bias"}
    with pytest.raises(RuntimeError, match="visual.blocks.1.mlp.fc2.bias"):
        m.load_state_dict(missing)
    extra = dict(small)
    extra["visual.bogus"] = torch.zeros(3)
    with pytest.raises(RuntimeError, match="visual.bogus"):
        KEEPModel().to("cuda:0").load_state_dict(extra)
    ok = dict(small)
    ok["text.embeddings.position_ids"] = torch.arange(512)[None].float()      # buffer of older checkpoints
    KEEPModel().to("cuda:0").load_state_dict(ok)
    # strict=True means BOTH towers, as in the reference (keep_inference.py:83): a checkpoint without any text.* key is an
    # error at load time, not a surprise at the first encode_text -- unless the engine was built for one tower on purpose
    image_only = {k: v for k, v in small.items() if not k.startswith("text.")}
    with pytest.raises(RuntimeError, match="text tower"):
        KEEPModel().to("cuda:0").load_state_dict(image_only)
    with pytest.raises(RuntimeError, match="image tower"):
        KEEPModel().to("cuda:0").load_state_dict({k: v for k, v in small.items() if k.startswith("text.") or k == "logit_scale"})
    KEEPModel(towers=("image",)).to("cuda:0").load_state_dict(image_only)
    KEEPModel().to("cuda:0").load_state_dict(image_only, strict=False)


@pytest.fixture()
def no_splitk():
    """The small-M split-K GEMM (gemm_f16_skinny.hip) sums K in a different order than the 256x256 kernel, so results
    of calls that cross its row threshold agree to rounding, not bit for bit.  The bit-exactness properties below are
    about batching / lanes / the CLS tail, so every model they build pins its GEMM path to one kernel."""
    _EXTRA_OPTS.update({"gemm_skinny_m": 0,        # no register-direct split-K kernel
                        "gemm_splitk_tiles": 0,    # no K-sliced 256x256 path for mid-size calls
                        "sgemv_m": 0})             # no few-row fp32 kernel for the head / pooler
    yield
    _EXTRA_OPTS.clear()


def test_splitk_path_matches_big_kernel(small, text_bank):
    """Same inputs through both GEMM paths: equal to rounding, each path bit-reproducible."""
    x = synth_tiles(1, seed=71).cuda()                       # M = 197 rows: register-direct split-K kernel by default
    x6 = synth_tiles(6, seed=73).cuda()
    toks = {k: v.cuda() for k, v in synth_prompts(2, 64, seed=72).items()}
    for precision, ptol in (("strict", 2e-6), ("comp", 2e-4), ("fp16", 2e-4)):
        m = make_model(small, precision)
        a_img, a_txt = m.encode_image(x), m.encode_text(toks)
        assert torch.equal(m.encode_image(x), a_img) and torch.equal(m.encode_text(toks), a_txt)
        a_mid = m.encode_image(x6)                           # M = 1182 rows: 256x256 tiles cut into K slices
        m.set_option("gemm_skinny_m", 0); m.set_option("sgemv_m", 0); m.set_option("gemm_splitk_tiles", 0)
        try:
            b_img, b_txt = m.encode_image(x), m.encode_text(toks)
            b_mid = m.encode_image(x6)
        finally:
            m.set_option("gemm_skinny_m", 320); m.set_option("sgemv_m", 16); m.set_option("gemm_splitk_tiles", 64)
        d_mid = (a_mid - b_mid).abs().max().item()
        print(f"[mid-size split-K vs 256x256 {precision}] max|dfeat|={d_mid:.3e}")
        assert d_mid < ptol and torch.equal(m.encode_image(x6), a_mid)
        d = max((a_img - b_img).abs().max().item(), (a_txt - b_txt).abs().max().item())
        print(f"[splitk vs 256x256 {precision}] max|dfeat|={d:.3e}")
        assert d < ptol
        with torch.no_grad():
            ref = O.encode_image(small, x.cpu())
        assert ((a_img.cpu() - ref) @ text_bank.t()).abs().max() < tol(precision)


def test_options_belong_to_their_handle(small):
    """Kernel-selection options are per handle (there is no process-wide kernel state): pinning model A to the 256x256 kernel
    must not change what model B computes, and vice versa."""
    x = synth_tiles(1, seed=171).cuda()                      # 197 rows: the default engine takes the register-direct split-K kernel
    b = make_model(small, "fp16")
    ref_b = b.encode_image(x)
    a = make_model(small, "fp16")
    for k in ("gemm_skinny_m", "gemm_splitk_tiles", "sgemv_m"):
        a.set_option(k, 0)
    out_a = a.encode_image(x)
    assert torch.equal(b.encode_image(x), ref_b)             # B is untouched by A's options
    a2 = make_model(small, "fp16")
    for k in ("gemm_skinny_m", "gemm_splitk_tiles", "sgemv_m"):
        a2.set_option(k, 0)
    assert torch.equal(a2.encode_image(x), out_a)            # and A's path is a property of A's options alone
    assert not torch.equal(out_a, ref_b)                     # (the two paths sum K in different orders: equal only to rounding)
    assert (out_a - ref_b).abs().max() < 2e-4


def test_graph_replay_is_bit_identical(small):
    """Launch-bound calls (<= 1024 rows) are captured once and replayed as a hipGraph: same kernels, same bits; a changed
    option or reloaded weights must invalidate the captured graph."""
    m = make_model(small, "fp16")
    x1, x2 = synth_tiles(1, seed=81).cuda(), synth_tiles(2, seed=82).cuda().to(torch.bfloat16)
    toks = {k: v.cuda() for k, v in synth_prompts(3, 256, seed=83).items()}
    m.set_option("graphs", 0)
    ref = [m.encode_image(x1), m.encode_image(x2), m.encode_text(toks), m.encode_text({"input_ids": toks["input_ids"][:1]})]
    m.set_option("graphs", 1)
    for _ in range(3):                                       # capture, then replays
        got = [m.encode_image(x1), m.encode_image(x2), m.encode_text(toks), m.encode_text({"input_ids": toks["input_ids"][:1]})]
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
    # many replays of interleaved graphs: no spurious token-range error, same bits every time
    first = {}
    for it in range(90):
        n = 1 + it % 3
        t = {k: v[:n].contiguous() for k, v in toks.items()}
        o = m.encode_text(t)
        assert torch.equal(o, first.setdefault(n, o.clone()))
    # different data through the same captured graph
    y1 = synth_tiles(1, seed=84).cuda()
    g1 = m.encode_image(y1)
    m.set_option("graphs", 0)
    assert torch.equal(m.encode_image(y1), g1)
    m.set_option("graphs", 1)
    # an option that changes the kernels invalidates the capture
    m.set_precision("strict")
    s1 = m.encode_image(x1)
    with torch.no_grad():
        o1 = O.encode_image(small, x1.cpu())
    assert (s1.cpu() - o1).abs().max() < 5e-6
    # out-of-range token ids are still reported through the replayed graph
    bad = {k: v.clone() for k, v in toks.items()}
    bad["input_ids"][0, 3] = 10 ** 6
    with pytest.raises(IndexError):
        m.encode_text(bad)                        # device inputs: reported lazily (no host synchronisation per call) ...
        m.check_errors()                          # ... at the next engine call or here
    m.encode_text(toks)
    # lazy reporting: the call itself returns, the NEXT engine call raises; immediate mode raises in the call
    m.encode_text(bad)
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        m.encode_image(x1)
    m.encode_image(x1)                            # reported once
    # the WSI pattern (thousands of one-prompt calls, then similarity): an error in the middle is reported by one of the later calls --
    # the device flag is sticky and every call's check is queued, so nothing is overwritten or dropped
    with pytest.raises(IndexError):
        m.encode_text(bad)
        feats = [m.encode_text({k: v[:1].contiguous() for k, v in toks.items()}) for _ in range(6)]
        torch.cuda.synchronize()
        m.similarity(feats[0], feats[1])
    m.similarity(m.encode_text(toks), m.encode_text(toks))      # acknowledged: clean again
    torch.cuda.synchronize()
    m.check_errors()
    m.check_token_ids = True
    with pytest.raises(IndexError):
        m.encode_text(bad)
    m.check_token_ids = False
    m.encode_text(bad); m.check_errors()          # never checked


def test_every_path_boundary_agrees_with_the_plain_path(small):
    """Batch sizes on both sides of every dispatch threshold (register-direct split-K <= 320 rows, graph replay <= 1024 rows,
    K-sliced 256x256 below 64 tiles, second lane from 16 tiles per lane, 256-tile sub-batches): the default engine must
    agree with the plain one (256x256 kernel only, one stream, no graphs) to rounding."""
    sizes = (1, 2, 5, 6, 8, 13, 16, 17, 31, 32, 33, 64, 65, 100, 257)
    x = synth_tiles(max(sizes), seed=77).cuda().to(torch.bfloat16)
    toks = {k: v.cuda() for k, v in synth_prompts(70, 64, seed=78).items()}
    for precision, ptol in (("strict", 3e-6), ("comp", 3e-4), ("fp16", 3e-4)):
        m = make_model(small, precision)
        got = {b: m.encode_image(x[:b]) for b in sizes}
        got_t = {pn: m.encode_text({k: v[:pn] for k, v in toks.items()}) for pn in (1, 5, 16, 17, 64, 70)}
        for k, v in (("gemm_skinny_m", 0), ("gemm_splitk_tiles", 0), ("sgemv_m", 0)):
            m.set_option(k, v)
        m.set_option("graphs", 0); m.set_option("streams", 1)
        try:
            plain = m.encode_image(x)
            plain_t = m.encode_text(toks)
        finally:
            for k, v in (("gemm_skinny_m", 320), ("gemm_splitk_tiles", 64), ("sgemv_m", 16)):
                m.set_option(k, v)
        worst = max((got[b] - plain[:b]).abs().max().item() for b in sizes)
        worst_t = max((got_t[pn] - plain_t[:pn]).abs().max().item() for pn in got_t)
        print(f"[path boundaries {precision}] image max|dfeat|={worst:.3e} text {worst_t:.3e}")
        assert worst < ptol and worst_t < ptol


def test_batch_chunking_is_invisible(small, no_splitk):
    m = make_model(small, "fp16")
    x = synth_tiles(7, seed=8).cuda()
    full = m.encode_image(x)
    m.set_option("max_tiles", 3)
    assert torch.equal(m.encode_image(x), full)
    toks = {k: v.cuda() for k, v in synth_prompts(5, 64, seed=9).items()}
    t_full = m.encode_text(toks)
    m.set_option("max_prompts", 2)
    assert torch.equal(m.encode_text(toks), t_full)


def test_multi_stream_lanes_match_single_stream(small, no_splitk):
    """encode_image splits batches >= 64 over internal streams (layer-interleaved lanes); results must not change."""
    m = make_model(small, "fp16")
    x = synth_tiles(70, seed=21).cuda()
    m.set_option("streams", 1)
    one = m.encode_image(x)
    for lanes in (2, 3):
        m.set_option("streams", lanes)
        assert torch.equal(m.encode_image(x), one)
        assert torch.equal(m.encode_image(x[:65]), one[:65])         # uneven split
    with torch.no_grad():
        ref = O.encode_image(small, x[60:70].cpu())
    assert (one[60:70].cpu() - ref).abs().max() < 2e-3
    # results stay ordered on the caller's stream: consume them immediately on that stream
    y = m.encode_image(x) * 2.0
    assert torch.equal(y, one * 2.0)


def test_non_default_stream_and_weight_reload(small):
    m = make_model(small, "fp16")
    x = synth_tiles(70, seed=5).cuda()
    ref = m.encode_image(x)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        out = m.encode_image(x)
        doubled = out * 2                      # consumer on the same (non-default) stream
    st.synchronize()
    assert torch.equal(out, ref) and torch.equal(doubled, ref * 2)
    # loading another state_dict replaces the weights in place
    other = synth_state_dict(small_shape(2, 2), seed=6)
    m.load_state_dict(other)
    with torch.no_grad():
        want = O.encode_image(other, x[:3].cpu())
    got = m.encode_image(x[:3]).cpu()
    assert (got - want).norm(dim=-1).max() < 3e-3 and (got - ref[:3].cpu()).abs().max() > 1e-3


def test_cls_only_tail_is_exact(small, no_splitk):
    """Last block on the CLS rows only (engine option cls_tail, default on) vs every token."""
    for precision in ("fp16", "strict"):
        m = make_model(small, precision)
        x = synth_tiles(6, seed=13).cuda()
        fast = m.encode_image(x)
        m.set_option("cls_tail", 0)
        full = m.encode_image(x)
        assert (fast - full).abs().max() < 2e-6


def test_last_block_query_for_the_cls_rows_only(small):
    """Lanes of >= 32 tiles: the last block's qkv GEMM computes K | V for every token and Q for the CLS rows alone (a [B, D] x W_q^T product on the
    small-M kernel, engine option cls_qkv) -- every other token's query is never read.  Same features as with all 3 D columns computed,
    to the fp16 rounding of the CLS query (the two kernels sum K in different orders); both inside the mode's tolerance of the oracle."""
    x = synth_tiles(70, seed=17)                                   # two lanes of 35 tiles
    with torch.no_grad():
        ref = O.encode_image(small, x)
    for precision in ("fp16", "comp"):
        m = make_model(small, precision)
        m.set_option("cls_qkv", 1)                                 # (off by default: measured level end to end)
        m.profile_enable("vit.tail")
        m.profile_reset()
        fast = m.encode_image(x.cuda())
        torch.cuda.synchronize()
        n_tail = m.profile_read("vit.tail")[1]
        m.set_option("cls_qkv", 0)
        m.profile_reset()
        full = m.encode_image(x.cuda())
        torch.cuda.synchronize()
        assert n_tail > m.profile_read("vit.tail")[1]             # the CLS-query gather + GEMM did run (timed under the tail tag)
        m.profile_disable()
        d = (fast - full).norm(dim=1).max().item()
        print(f"[cls_qkv {precision}] |fast - full| {d:.3e}; vs oracle fast {(fast.cpu() - ref).norm(dim=1).max():.3e} full {(full.cpu() - ref).norm(dim=1).max():.3e}")
        assert d < 2e-4 and (fast.cpu() - ref).norm(dim=1).max() < 1.2 * max((full.cpu() - ref).norm(dim=1).max().item(), 1e-5)


# ------------------------------------------------------------------ full depth vs golden (HF outputs)
@pytest.mark.parametrize("precision", MODES)
def test_full_depth_image_tower_vs_golden(golden_dir, text_bank, precision):
    g = np.load(os.path.join(golden_dir, "vit_d24.npz"))
    sd = synth_state_dict(KEEPShape(), seed=int(g["weight_seed"]), text=False)
    x = synth_tiles(int(g["batch"]), seed=int(g["tile_seed"]))
    m = make_model(sd, precision)
    out = m.encode_image(x)
    ref = torch.from_numpy(g["features"])
    dcos = (out @ text_bank.t() - ref @ text_bank.t()).abs().max().item()
    print(f"[vit d24 {precision}] max|dfeat|={(out - ref).abs().max():.3e} |df|={(out - ref).norm(dim=-1).max():.3e} max|dcos|={dcos:.3e}")
    assert dcos < tol(precision, 5e-6)
    assert torch.equal((out @ text_bank.t()).argmax(1), (ref @ text_bank.t()).argmax(1))


@pytest.mark.parametrize("precision", ["strict", "comp"])
def test_bench_weights_vs_both_image_tower_pins(golden_dir, text_bank, precision):
    """The weights bench.py runs on (seed 0), 8 tiles, against BOTH independent fp32 implementations the oracle is pinned to
    (tools/make_golden.py vit24_bench): transformers' Dinov2Model configured as ViT-L/16, and the module tree of timm's
    vit_large_patch16_224 on the ATen ops timm dispatches (F.conv2d / F.layer_norm / F.scaled_dot_product_attention / F.gelu / F.linear),
    loaded strictly from the release key layout.  The fixture also records how far the oracle and the two pins are from each other (< 3e-7)."""
    g = np.load(os.path.join(golden_dir, "vit_d24_bench.npz"))
    assert float(g["oracle_dfeat"]) < 1e-6 and float(g["oracle_dfeat_aten_timm"]) < 1e-6 and float(g["pins_dfeat"]) < 1e-6
    sd = synth_state_dict(KEEPShape(), seed=int(g["weight_seed"]), text=False)
    x = synth_tiles(int(g["batch"]), seed=int(g["tile_seed"]))
    m = make_model(sd, precision)
    out = m.encode_image(x)
    for name in ("features", "features_aten_timm") + (("features_timm",) if "features_timm" in g.files else ()):     # the third once tools/pin_against_timm.py has run
        ref = torch.from_numpy(g[name])
        dcos = (out @ text_bank.t() - ref @ text_bank.t()).abs().max().item()
        print(f"[bench weights {precision} vs {name}] max|dfeat|={(out - ref).abs().max():.3e} max|dcos|={dcos:.3e}")
        assert dcos < tol(precision, 5e-6)
        assert torch.equal((out @ text_bank.t()).argmax(1), (ref @ text_bank.t()).argmax(1))


@pytest.mark.parametrize("precision", MODES)
def test_full_depth_text_tower_vs_golden(golden_dir, text_bank, precision):
    g = np.load(os.path.join(golden_dir, "bert_l12.npz"))
    sd = synth_state_dict(KEEPShape(), seed=int(g["weight_seed"]), vision=False)
    toks = {k: torch.from_numpy(g[k].astype(np.int64)) for k in ("input_ids", "token_type_ids", "attention_mask")}
    m = make_model(sd, precision)
    out = m.encode_text(toks)
    ref = torch.from_numpy(g["features"])
    dcos = (out @ text_bank.t() - ref @ text_bank.t()).abs().max().item()
    print(f"[bert l12 {precision}] max|dfeat|={(out - ref).abs().max():.3e} max|dcos|={dcos:.3e}")
    assert dcos < tol(precision, 5e-6)


@pytest.mark.parametrize("precision", ["comp", "fp16"])
def test_bench_sized_batch_is_position_independent(no_splitk, precision):
    """BASELINE config 2 size (256 tiles, full depth, two internal lanes) in the mode bench.py reports ('comp': fp16 pass + MX-fp4
    correction planes, whose block scales are per row) and in the plain fp16 mode: the oracle cannot run this in seconds, so use a
    size-independent property -- a tile's embedding must not depend on where it sits in the batch, which lane it lands in, or who its
    neighbours are.  The 8 distinct tiles are oracle-checked at small batch by the tests above."""
    sd = synth_state_dict(KEEPShape(), seed=41, text=False)
    m = make_model(sd, precision)
    base = synth_tiles(8, seed=42).to(torch.bfloat16).cuda()
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(43))
    idx = (perm % 8).cuda()
    big = m.encode_image(base[idx])
    for k in range(8):                                       # every copy of tile k, wherever it sits, has the same bits
        rows = big[idx == k]
        assert torch.equal(rows, rows[:1].expand_as(rows)), k
    other = m.encode_image(base[torch.flip(idx, dims=[0])])
    assert torch.equal(torch.flip(other, dims=[0]), big)     # lanes swapped: same bits
    small = m.encode_image(base)                             # 8 tiles take the small-M kernels ('comp': split products instead of fp4 planes)
    assert (small[idx] - big).abs().max() < (3e-4 if precision == "fp16" else 1.5e-4)
    with torch.no_grad():
        ref = O.encode_image(sd, base[:2].float().cpu())
    assert (big[(idx == 0).nonzero()[0, 0]].cpu() - ref[0]).norm() < 3e-3


def test_calls_of_512_and_4096_tiles_are_position_independent(no_splitk):
    """Calls of >= 512 tiles run as chunks of two 256-tile lanes (what bench.py's config-3 / config-4 / slide legs and
    `keep_amd.distributed.encode_tiles_sharded` now pass per call: +1.7 % over 256-tile calls, profiles/r05_step_boundary_bubble.txt).  Same property as above at
    those sizes: every copy of a tile has the same bits wherever it sits, in whichever chunk and lane; against the 256-tile call (lanes of 128: the CLS-row
    chain sums its K slices in another order) the features agree to fp32 rounding."""
    sd = synth_state_dict(KEEPShape(), seed=41, text=False)
    m = make_model(sd, "comp")
    base = synth_tiles(8, seed=42).to(torch.bfloat16).cuda()
    ref256 = m.encode_image(base[(torch.arange(256) % 8).cuda()])[:8]
    for n in (512, 4096):
        idx = (torch.randperm(n, generator=torch.Generator().manual_seed(n)) % 8).cuda()
        big = m.encode_image(base[idx])
        for k in range(8):
            rows = big[idx == k]
            assert torch.equal(rows, rows[:1].expand_as(rows)), (n, k)
            assert (rows[0] - ref256[k]).abs().max() < 2e-6, (n, k)


def test_compensated_mode_takes_the_fp4_path_on_bench_sized_lanes(small, text_bank):
    """Lanes of >= 32 tiles run fc1 / fc2 as fp16 pass + MX-fp4 correction terms (smaller calls use split products, which
    the other tests cover): 64 tiles = two lanes of 32.  Held to the tolerance, and required to beat the plain fp16 mode."""
    x = synth_tiles(64, seed=91)
    with torch.no_grad():
        ref = O.encode_image(small, x) @ text_bank.t()
    errs = {}
    p18 = {"comp_full_blocks": 1, "comp_mlp_blocks": 8}        # the prefix plan 1 / 8: every MLP GEMM of this depth-2 model on the fp16 + MX-fp4 kernels
    for name, precision, opts in (("fp16", "fp16", {}), ("comp", "comp", p18), ("comp, attention side plain", "comp", {**p18, "comp_full_blocks": 0}),
                                  ("comp, one lane", "comp", {**p18, "streams": 1}),
                                  ("comp, qkv of block 0 compensated instead of split", "comp", {**p18, "comp_qkv": 1}), ("comp, the handle's initial plan", "comp", {})):
        m = KEEPModel(precision=precision, towers=towers_of(small))
        m.auto_calibrate = False                  # explicit plans: this test is about the fp4 path, not about what calibrate() picks
        m.load_state_dict(small, strict=True)
        m.to("cuda:0")
        for k, v in opts.items():
            m.set_option(k, v)
        d = (m.encode_image(x.cuda()).cpu() @ text_bank.t() - ref).abs()
        errs[name] = (d.max().item(), d.pow(2).mean().sqrt().item())
        print(f"[64 tiles d2 {name}] max|dcos|={errs[name][0]:.3e} rms={errs[name][1]:.3e}")
    assert errs["comp"][0] < COS_TOL and errs["comp, one lane"][0] < COS_TOL and errs["comp, qkv of block 0 compensated instead of split"][0] < COS_TOL
    assert errs["comp, the handle's initial plan"][0] < COS_TOL and errs["comp, the handle's initial plan"][1] < 0.5 * errs["fp16"][1]
    assert errs["comp"][1] < 0.5 * errs["fp16"][1]
    assert errs["comp, attention side plain"][1] < 0.8 * errs["fp16"][1]


def test_dual_tower_similarity_full_depth():
    """Config 3 in miniature: 16 tiles x 8 prompts through both towers, sim matrix + argmax."""
    sd = synth_state_dict(KEEPShape(), seed=31)
    x, toks = synth_tiles(16, seed=32), synth_prompts(8, 256, seed=33)
    with torch.no_grad():
        ri, rt = O.encode_image(sd, x), O.encode_text(sd, toks)
    ref = O.similarity(ri, rt)
    for precision in MODES:
        ptol = tol(precision, 5e-6)
        m = make_model(sd, precision)
        sim, lab = m.similarity(m.encode_image(x.cuda()), m.encode_text({k: v.cuda() for k, v in toks.items()}), mode="argmax")
        d = (sim.cpu() - ref).abs().max().item()
        print(f"[dual {precision}] max|dcos|={d:.3e}")
        assert d < ptol
        assert torch.equal(lab.cpu(), O.sim_argmax(ref))


def test_config3_fixture_first_chunk(golden_dir):
    """BASELINE config 3 against the committed fp32-oracle fixture (tools/make_golden.py c3; bench.py runs all 4096 tiles):
    the first 256 tiles x 64 prompts through both towers = 16 384 cosines.  Default mode inside 1e-4, and the labels of the
    default path -- ``classify``: default-precision encode, tiles with a top-2 margin below ``label_margin`` encoded again with
    split products -- are EXACTLY the oracle's (north star: "tile argmax labels bit-exact"; the chunk holds oracle margins down
    to 1e-6).  The plain ``similarity(mode='argmax')`` of default-precision features can only promise a label where the oracle's
    margin exceeds twice the cosine error: kept as the statement of why the second look exists."""
    g = np.load(os.path.join(golden_dir, "c3_dual_tower.npz"))
    sd = synth_state_dict(KEEPShape(), seed=int(g["weight_seed"]))
    chunk = int(g["chunk"])
    x = synth_tiles(chunk, seed=int(g["tile_seed0"]))
    assert abs(float(x.double().abs().sum()) - float(g["tiles0_checksum"])) < 1e-3 * float(g["tiles0_checksum"])
    toks = {"input_ids": torch.from_numpy(g["input_ids"].astype(np.int64)), "attention_mask": torch.from_numpy(g["attention_mask"].astype(np.int64))}
    toks["token_type_ids"] = torch.zeros_like(toks["input_ids"])
    ref = torch.from_numpy(g["sims"][:chunk])
    ref_lab, margin = torch.from_numpy(g["argmax"][:chunk].astype(np.int64)), torch.from_numpy(g["margin"][:chunk])
    assert int((margin < 2.5e-4).sum()) >= 8 and float(margin.min()) > 5e-7          # the chunk does hold near ties, none below the strict mode's error
    for precision in ("comp", "strict"):
        m = make_model(sd, precision)
        txt = m.encode_text({k: v.cuda() for k, v in toks.items()})
        sim, lab = m.similarity(m.encode_image(x.cuda()), txt, mode="argmax")
        d = (sim.cpu() - ref).abs()
        differ = lab.cpu().long() != ref_lab
        assert d.max() < tol(precision, 5e-6)
        assert not (differ & (margin > 2 * d.max())).any()
        csim, clab = m.classify(x.cuda(), txt)
        dc = (csim.cpu() - ref).abs()
        print(f"[c3 first chunk {precision}] max|dcos|={d.max():.3e} rms={d.pow(2).mean().sqrt():.3e}; plain argmax: {int(differ.sum())} labels differ; "
              f"classify: {m.last_rechecked} of {chunk} tiles encoded twice, max|dcos|={dc.max():.3e}, labels differing {int((clab.cpu().long() != ref_lab).sum())}")
        assert dc.max() < tol(precision, 5e-6)
        assert torch.equal(clab.cpu().long(), ref_lab)
        if precision == "comp":
            assert 0 <= m.last_rechecked < chunk // 4                # (with the margin scaled by this bank's prompt distances the first chunk may hold no tile to look at twice)
            lm = m.get_option("label_margin")                       # set by calibrate(): sqrt 2 x the predicted worst cosine error (<= 1.42e-4), not a fixed 2.5e-4
            assert 5e-5 < lm <= 2 ** 0.5 * COS_TOL * 1.001 and lm == pytest.approx(m.calibration["label_margin"], rel=1e-3)
            # classify scales the calibrated per-unit margin by THIS bank's largest prompt distance (<= 2; sqrt 2 is the engine's own default)
            diam = float((2.0 - 2.0 * (txt @ txt.t()).min()).clamp_min(0).sqrt())
            assert m.last_margin == pytest.approx(m.calibration["label_margin_per_unit_prompt_distance"] * diam, rel=1e-3) and m.last_margin <= lm * 2 ** 0.5 * 1.001
            lm = m.last_margin
            flagged = (csim.topk(2, dim=1).values.diff(dim=1).abs().squeeze(1) < 0.9 * lm).cpu()      # rows that were looked at again carry strict-grade cosines
            assert (dc[flagged].max() < 5e-6) if flagged.any() else True
        else:
            assert m.last_rechecked == 0


def test_near_tie_argmax(text_bank):
    """Two prompts ~1e-4 apart in cosine for a typical tile (t2 = t1 nudged by 3e-3 along a random direction; 65 of the 96 tiles
    have an oracle margin below 1e-4, the smallest is 5.4e-6): ``classify`` must return exactly the oracle's labels in the default
    and in the strict mode; the plain argmax of default-precision features may differ only below twice its cosine error."""
    sd = synth_state_dict(small_shape(2, 2), seed=5, text=False)
    x = synth_tiles(96, seed=123)
    with torch.no_grad():
        ref_f = O.encode_image(sd, x)
    gen = torch.Generator().manual_seed(10)
    u = torch.nn.functional.normalize(torch.randn(768, generator=gen), dim=0)
    pairs = []
    for t1 in text_bank[:8]:
        pairs += [t1, torch.nn.functional.normalize(t1 + 3e-3 * u, dim=0)]
    bank = torch.stack(pairs)
    ref = ref_f @ bank.t()
    ref_lab = ref.argmax(1)
    top2 = ref.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    assert (margin < 1e-4).sum() > 50 and margin.min() > 3e-6                 # near ties, all decidable by the strict arithmetic
    for precision in ("comp", "strict"):
        m = make_model(sd, precision)
        sim, lab = m.similarity(m.encode_image(x.cuda()), bank.cuda(), mode="argmax")
        err = (sim.cpu() - ref).abs().max().item()
        differ = lab.cpu().long() != ref_lab
        assert not (differ & (margin > 2 * err)).any()
        csim, clab, cfeat = m.classify(x.cuda(), bank.cuda(), return_features=True)
        print(f"[near tie {precision}] plain: max|dcos|={err:.2e}, {int(differ.sum())} of {len(x)} labels differ; classify: {m.last_rechecked} tiles encoded twice, "
              f"{int((clab.cpu().long() != ref_lab).sum())} differ")
        assert torch.equal(clab.cpu().long(), ref_lab)
        assert (cfeat.cpu() @ bank.t() - csim.cpu()).abs().max() < 1e-6     # the returned features are the ones the similarity was taken from
        # scale: margins are compared in cosine units; labels do not change with the logit scale (keep_inference.py:52, exp = 25)
        _, clab25 = m.classify(x.cuda(), bank.cuda(), scale=25.0)
        assert torch.equal(clab25, clab)
        # margin 0 switches the second look off: then classify IS encode + similarity
        s0, l0 = m.classify(x.cuda(), bank.cuda(), margin=0.0)
        assert m.last_rechecked == 0 and torch.equal(l0, lab) and torch.equal(s0, sim)


def test_classify_edge_cases(small, text_bank):
    m = make_model(small, "comp")
    bank = text_bank[:5].cuda()
    s, l = m.classify(torch.empty(0, 3, 224, 224).cuda(), bank)
    assert s.shape == (0, 5) and l.shape == (0,) and m.last_rechecked == 0
    x = synth_tiles(3, seed=1)
    s1, l1 = m.classify(x, text_bank[:5])                                   # host in -> host out
    assert s1.device.type == "cpu" and l1.dtype == torch.int32
    with torch.no_grad():
        ref = O.encode_image(small, x) @ text_bank[:5].t()
    assert (s1 - ref).abs().max() < COS_TOL and torch.equal(l1.long(), ref.argmax(1))
    s_one, l_one = m.classify(x.cuda(), text_bank[:1].cuda())              # a single prompt: nothing to decide
    assert m.last_rechecked == 0 and int(l_one.abs().sum()) == 0
    u8 = torch.randint(0, 256, (4, 224, 224, 3), dtype=torch.uint8)
    su, lu = m.classify(u8.cuda(), bank, margin=1.0)                        # every tile flagged: all rows come from the strict pass
    assert m.last_rechecked == 4
    ms = make_model(small, "strict")
    assert (ms.encode_image_uint8(u8.cuda()) @ bank.t() - su).abs().max() < 2e-6
    with pytest.raises(ValueError):
        m.classify(x.cuda(), bank, scale=0.0)
    with pytest.raises(ValueError):
        m.classify(x.cuda()[:, :, :100], bank)


# ------------------------------------------------------------------ similarity modes
@pytest.mark.parametrize("N,P", [(1, 9), (63, 16), (64, 17), (1000, 40), (4096, 64), (130, 64)])
def test_similarity_9_to_64_prompts_fused_kernel(N, P):
    """The config-3 shape (up to 64 prompts): fused fp32-MFMA similarity + argmax / softmax kernel, ragged row and column
    counts, ties resolved like torch.argmax (first maximum), equal to the plain GEMM path to fp32 rounding."""
    m = KEEPModel()
    g = torch.Generator().manual_seed(N * 100 + P)
    img = torch.nn.functional.normalize(torch.randn(N, 768, generator=g), dim=-1)
    txt = torch.nn.functional.normalize(torch.randn(P, 768, generator=g), dim=-1)
    txt[P - 1] = txt[2]                                     # a tie between column 2 and the last column (different lanes / tiles)
    ref = O.similarity(img, txt)
    raw = m.similarity(img, txt).cpu()
    assert (raw - ref).abs().max() < 1e-6
    sim, lab = m.similarity(img, txt, scale=25.0, mode="argmax")
    assert (sim.cpu() - 25.0 * ref).abs().max() < 3e-5
    got, exp = lab.cpu().long(), O.sim_argmax(ref).long()
    top2 = ref.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-6                 # rows whose maximum is unique beyond fp32 rounding
    assert torch.equal(got[clear], exp[clear])
    tied = ref.argmax(1) == 2                                # rows won by the duplicated prompt: the lower index must be reported
    assert (got[tied] == 2).all() and not (got == P - 1).any()
    sm = m.similarity(img, txt, scale=10.0, mode="softmax").cpu()
    assert (sm - O.sim_softmax(ref, 10.0)).abs().max() < 1e-6 and (sm.sum(1) - 1).abs().max() < 1e-5
    sm16 = m.similarity(img, txt, scale=10.0, mode="softmax_f16")
    assert sm16.dtype == torch.float16 and (sm16.float().cpu() - O.sim_softmax(ref, 10.0)).abs().max() < 6e-4
    m.set_option("sgemv_m", 0)                               # plain path: 128x128 fp32 GEMM + row kernels
    assert (m.similarity(img, txt).cpu() - raw).abs().max() < 1e-6
    _, lab2 = m.similarity(img, txt, scale=25.0, mode="argmax")
    assert torch.equal(lab2.cpu().long()[clear], exp[clear])



def test_config5_probability_map_at_full_size_against_the_oracle():
    """BASELINE config 5 at its stated size: the fp16 probability map softmax(10 cos) of 100 000 tiles x 2 classes, EVERY entry against the fp32 CPU oracle
    (`O.sim_softmax(O.similarity(...))`: a 0.3 GFLOP matmul on the host).  Tolerance: half an fp16 ulp at 0.5 (probabilities in [0.5, 1) are stored to 2^-11) + the
    fp32 path's 1e-6; rows sum to 1 within one ulp."""
    import bench
    m = KEEPModel()
    r = bench.config5(m, torch.device("cuda", 0), n=100_000)
    print(f"[config 5, 100 000 x 2] {r['us']} us, {r['GBps']} GB/s ({r['frac_of_hbm_peak']} of the HBM peak); max |err| vs the oracle {r['max_abs_err_vs_oracle']:.3e}")
    assert r["tiles_checked_against_oracle"] == 100_000 and r["within_fp16_rounding_of_oracle"] and r["max_abs_err_vs_oracle"] <= 2.0 ** -12 + 1e-6
    g = torch.Generator().manual_seed(56)
    feats = torch.nn.functional.normalize(torch.randn(4096, 768, generator=g), dim=-1).cuda()
    cls = torch.nn.functional.normalize(torch.randn(2, 768, generator=g), dim=-1).cuda()
    p = m.similarity(feats, cls, scale=10.0, mode="softmax_f16").float()
    assert (p.sum(1) - 1.0).abs().max() <= 2.0 ** -11 + 1e-6


def test_similarity_modes():
    m = KEEPModel()
    g = torch.Generator().manual_seed(5)
    img = torch.nn.functional.normalize(torch.randn(1000, 768, generator=g), dim=-1)
    txt = torch.nn.functional.normalize(torch.randn(64, 768, generator=g), dim=-1)
    ref = O.similarity(img, txt)
    raw = m.similarity(img, txt).cpu()
    assert (raw - ref).abs().max() < 1e-6
    sim, lab = m.similarity(img, txt, scale=25.0, mode="argmax")
    assert (sim.cpu() - 25.0 * ref).abs().max() < 3e-5 and torch.equal(lab.cpu(), O.sim_argmax(ref))
    sm = m.similarity(img, txt[:4], scale=10.0, mode="softmax").cpu()
    assert (sm - O.sim_softmax(O.similarity(img, txt[:4]), 10.0)).abs().max() < 1e-6
    sm16 = m.similarity(img, txt[:2], scale=10.0, mode="softmax_f16")
    assert sm16.dtype == torch.float16
    assert (sm16.float().cpu() - O.sim_softmax(O.similarity(img, txt[:2]), 10.0)).abs().max() < 6e-4
    sc = m.similarity(img, txt[:4], mode="top2score")
    assert abs(sc - O.rank_cls_score(O.similarity(img, txt[:4]))) < 1e-6
    # ties: lowest index wins, as torch.argmax on CPU
    t2 = torch.cat([txt[:1], txt[:1], txt[1:3]])
    _, lab2 = m.similarity(img, t2, mode="argmax")
    assert torch.equal(lab2.cpu(), O.sim_argmax(O.similarity(img, t2)))


# ------------------------------------------------------------------ other weight distributions, calibration, fp16 range
@pytest.mark.parametrize("family", ["heavy_tail", "small_ls"])
def test_weight_families_calibrated_default_mode_within_tolerance(golden_dir, family):
    """The default mode on weights it was NOT tuned on (keep_amd.synth families: massive residual channels + outlier LayerNorm gains +
    heavy-tailed weights; tiny LayerScale), full depth, both towers, against the fp32 oracle (tools/make_golden.py families):
    load_state_dict calibrates the 'comp' setting on a probe batch; with whatever it picked, every cosine is inside 1e-4 and the
    labels of the labelled path are the oracle's."""
    g = np.load(os.path.join(golden_dir, f"family_{family}.npz"))
    sd = synth_state_dict(KEEPShape(), seed=int(g["weight_seed"]), family=family)
    assert abs(float(sd["visual.blocks.0.attn.qkv.weight"].double().abs().sum()) - float(g["qkv0_checksum"])) < 1e-6 * float(g["qkv0_checksum"])
    x = synth_tiles(int(g["n_tiles"]), seed=int(g["tile_seed"]))
    toks = {"input_ids": torch.from_numpy(g["input_ids"].astype(np.int64)), "attention_mask": torch.from_numpy(g["attention_mask"].astype(np.int64))}
    toks["token_type_ids"] = torch.zeros_like(toks["input_ids"])
    ref, ref_lab, margin = torch.from_numpy(g["sims"]), torch.from_numpy(g["argmax"].astype(np.int64)), torch.from_numpy(g["margin"])
    m = make_model(sd, "comp")
    cal = m.calibration
    assert cal is not None and cal["precision"] in ("comp", "strict") and cal["tried"]
    assert cal["precision"] == "strict" or (cal["tried"][-1]["predicted_max_abs_dcos"] <= COS_TOL and cal["exceedance_probability"] <= 0.01 + 1e-9)
    txt = m.encode_text({k: v.cuda() for k, v in toks.items()})
    sim, lab = m.classify(x.cuda(), txt)
    d = (sim.cpu() - ref).abs()
    print(f"[family {family}] calibrated to {cal['precision']} {cal['plan']} after {len(cal['tried'])} candidate(s) "
          f"(probe errors {[t['max_abs_dcos'] for t in cal['tried']]}); max|dcos| vs oracle {d.max():.3e} rms {d.pow(2).mean().sqrt():.3e}; "
          f"{m.last_rechecked} tiles encoded twice; smallest oracle margin {float(margin.min()):.2e}")
    assert d.max() < COS_TOL
    decidable = margin > 2e-6
    assert torch.equal(lab.cpu().long()[decidable], ref_lab[decidable])
    # the uncalibrated built-in setting, for the record (it may or may not hold on this family -- that is why calibrate() exists)
    m2 = KEEPModel(precision="comp", towers=towers_of(sd))
    m2.auto_calibrate = False
    m2.load_state_dict(sd, strict=True)
    m2.to("cuda:0")
    assert m2.calibration is None
    d2 = (m2.encode_image(x.cuda()).cpu() @ txt.cpu().t() - ref).abs()
    print(f"[family {family}] the handle's initial plan without calibration: max|dcos| {d2.max():.3e}")


@pytest.mark.parametrize("budget", ["ladder", "measured"])
def test_calibrate_walks_its_candidates_and_reports(small, budget):
    from keep_amd.model import plan_string
    m = make_model(small, "comp")
    assert m.calibration["precision"] == "comp" and m.calibration["budget"] == m.calibration_budget == "measured"      # what load_state_dict ran
    assert plan_string(m.get_plan()) == m.calibration["plan"]

    def cost(c):                                               # a plan's price in knobs: an attention side counts double
        if c["precision"] != "comp":
            return 99
        a, mm = (part.split(":")[1] for part in c["plan"].split())
        return sum(2 * (ch != "0") for ch in a) + sum(ch != "0" for ch in mm)

    # an unreachable target tries its candidates and ends in the split-product mode
    cal = m.calibrate(n_tiles=64, tolerance=1e-9, budget=budget)
    assert cal["precision"] == "strict" and cal["budget"] == budget and len(cal["tried"]) >= (2 if budget == "ladder" else 1) and m.get_option("precision") == 1
    errs = [t["max_abs_dcos"] for t in cal["tried"]]
    assert errs[-1] <= errs[0] * 1.2                       # more compensated blocks: not worse
    x = synth_tiles(4, seed=2).cuda()
    with torch.no_grad():
        ref = O.encode_image(small, x.cpu())
    assert (m.encode_image(x).cpu() - ref).abs().max() < 5e-6
    # a generous target keeps the cheapest candidate; explicit probe tiles and prompts are accepted
    m.set_precision("comp")
    cal = m.calibrate(tiles=synth_tiles(40, seed=9).to(torch.bfloat16), text_features=torch.nn.functional.normalize(torch.randn(7, 768), dim=-1), tolerance=1e-2,
                      budget=budget)
    assert cal["precision"] == "comp" and len(cal["tried"]) == 1 and "40 tiles x 7 prompts" in cal["probe"] and cost(cal) == 0
    if budget == "measured":
        sh = cal["variance_shares"]
        assert len(sh["attn"]) == len(sh["mlp"]) == 2 and sh["floor"] < min(sh["mlp"]) and sh["residual_mlp"][2] < 0.3
    # the rule is population-aware: a larger population (more tiles x distinct prompts to compare) asks for a smaller rms, never a larger one
    small_pop, large_pop = m.calibrate(population=1e4, budget=budget), m.calibrate(population=1e9, budget=budget)
    assert small_pop["target_rms_dcos"] > large_pop["target_rms_dcos"] and small_pop["max_sigmas_quantile"] < large_pop["max_sigmas_quantile"]
    assert cost(small_pop) <= cost(large_pop)
    # ... and confidence-aware: the quantile sits above the location of the maximum, the more so the higher the confidence
    lo, hi = m.calibrate(confidence=0.5, budget=budget), m.calibrate(confidence=0.999, budget=budget)
    assert lo["expected_max_sigmas"] < lo["max_sigmas_quantile"] < hi["max_sigmas_quantile"] and lo["target_rms_dcos"] > hi["target_rms_dcos"]
    for c in (lo, hi):
        if c["precision"] == "comp":
            assert c["exceedance_probability"] <= 1.0 - c["confidence"] + 1e-9
            from keep_amd.model import max_sigmas_quantile
            z_ratio = max_sigmas_quantile(c["label_population"], c["confidence"]) / max_sigmas_quantile(c["population"], c["confidence"])
            # one margin per tile: the quantile over the label population of the same per-tile error mixture (equal sigmas: exactly the ratio of the two quantiles)
            assert 0.9 * z_ratio <= c["label_margin"] / (2 ** 0.5 * c["predicted_max_abs_dcos"]) <= 1.0 + 1e-6 and z_ratio <= 1.0
    if hi["precision"] == "comp":
        assert m.get_option("label_margin") == pytest.approx(hi["label_margin"], rel=1e-3)         # the last calibration set the engine's second-look threshold
    # strict_blocks set by the caller survives a calibration (it used to be reset to 0)
    m.set_precision("comp", strict_blocks=1)
    m.calibrate(budget=budget)
    assert m.get_option("strict_blocks") == 1 and m.calibration["strict_blocks"] == 1
    fp = make_model(small, "fp16")
    assert fp.calibration is None and fp.calibrate() is None            # only the compensated mode has something to choose


def test_per_block_plan_is_what_the_prefix_options_stand_for():
    """keep_set_block_precision: the (comp_full_blocks, comp_mlp_blocks) shorthand and the same plan set block by block run the same kernels
    (bit-identical features); a plan is read back from the handle, survives being re-applied, is replaced by a later shorthand; every mode of
    the plan lands between the plain and the split arithmetic; out-of-range modes and blocks are refused."""
    from keep_amd import _lib
    from keep_amd.model import plan_prefix, prefix_plan
    depth = 4
    sd = synth_state_dict(small_shape(depth, 2), seed=8, text=False)
    x = synth_tiles(64, seed=31).cuda()                                  # two lanes of 32 tiles: the compensated kernels run
    with torch.no_grad():
        ref = O.encode_image(sd, x.cpu())
    m = KEEPModel(small_shape(depth, 2), precision="comp", towers=("image",))
    m.auto_calibrate = False
    m.load_state_dict(sd, strict=True)
    m.to("cuda:0")
    assert m.get_plan() == [(_lib.ATTN_SPLIT_COMPQKV, _lib.MLP_COMP)] + [(_lib.ATTN_PLAIN, _lib.MLP_CLS)] * (depth - 1)      # what a handle starts with
    assert m.get_option("plan_custom") == 0
    m.set_option("comp_full_blocks", 1); m.set_option("comp_mlp_blocks", 2)
    a = m.encode_image(x)
    m.set_plan(prefix_plan(depth, 1, 2))
    assert m.get_option("comp_full_blocks") == 1 and m.get_option("comp_mlp_blocks") == 2 and torch.equal(m.encode_image(x), a)
    custom = [(_lib.ATTN_SPLIT, _lib.MLP_COMP), (_lib.ATTN_PLAIN, _lib.MLP_COMP), (_lib.ATTN_PLAIN, _lib.MLP_PLAIN), (_lib.ATTN_PLAIN, _lib.MLP_PLAIN)]
    m.set_plan(custom)                                                   # the same plan, written block by block... it IS a prefix: stored as the shorthand
    assert plan_prefix(custom) == (1, 2) and m.get_plan() == custom and torch.equal(m.encode_image(x), a)
    err = lambda f: float((f.cpu() - ref).norm(dim=1).pow(2).mean().sqrt())
    m.set_plan([(0, 0)] * depth); e_plain = err(m.encode_image(x))
    m.set_plan([(1, 1)] * depth); e_split = err(m.encode_image(x))
    assert e_split < 0.05 * e_plain
    got = {}
    for name, plan in {"mlp comp": [(0, 2)] * depth, "mlp comp, W_lo term": [(0, 3)] * depth, "attn split": [(1, 0)] * depth,
                       "attn split, comp qkv": [(2, 0)] * depth, "comp qkv only": [(3, 0)] * depth, "mlp plain + CLS rows split": [(0, 4)] * depth,
                       "mixed": [(1, 2), (3, 3), (0, 4), (2, 0)]}.items():
        m.set_plan(plan)
        assert m.get_plan() == plan and (m.get_option("plan_custom") == 1 or plan_prefix(plan) is not None)
        got[name] = err(m.encode_image(x))
        assert torch.equal(m.encode_image(x), m.encode_image(x))         # a plan is deterministic
    print(f"[plans, depth {depth}] feature-error rms: plain {e_plain:.3e} split {e_split:.3e} " + " ".join(f"{k}: {v:.3e}" for k, v in got.items()))
    assert all(e_split * 0.9 <= v < e_plain for v in got.values())
    assert got["mlp comp"] < got["mlp comp, W_lo term"] < e_plain and got["attn split"] <= got["attn split, comp qkv"] * 1.05 and got["attn split, comp qkv"] < got["comp qkv only"]
    # the CLS rows redone as split products: the pooled feature is one of them, so most of the MLP's error share goes with 0.5 % of the rows
    assert got["mlp plain + CLS rows split"] < 0.93 * e_plain
    # KEEP_ATTN_PROJ_CLS (round 6): plain rows + the CLS rows' proj again as a split product; alone and chained with the CLS-row MLP (N(0,1) tiles: a small
    # share -- what it does on correlated tiles is measured in tests/test_tile_families.py)
    m.set_plan([(4, 0)] * depth); e_pc = err(m.encode_image(x)); assert m.get_plan() == [(4, 0)] * depth
    m.set_plan([(4, 4)] * depth); e_pc_cls = err(m.encode_image(x))
    m.set_plan([(5, 4)] * depth); e_cq_pc = err(m.encode_image(x)); assert m.get_plan() == [(5, 4)] * depth          # + a compensated qkv GEMM: not worse
    assert e_cq_pc < 1.02 * e_pc_cls
    assert torch.equal(m.encode_image(x), m.encode_image(x))
    print(f"[plans, depth {depth}] CLS rows' proj split: {e_pc:.3e}; + CLS-row MLP: {e_pc_cls:.3e}")
    assert e_split * 0.9 <= e_pc < 1.02 * e_plain and e_pc_cls < 1.02 * got["mlp plain + CLS rows split"]
    m.set_plan([(1, 4)] * depth); e_cls = err(m.encode_image(x))
    m.set_plan([(1, 0)] * depth); e_attn_only = err(m.encode_image(x))
    m.set_plan([(1, 2)] * depth); e_comp = err(m.encode_image(x))
    print(f"[plans, depth {depth}] attention side split + MLP: plain {e_attn_only:.3e}, CLS rows split {e_cls:.3e}, compensated {e_comp:.3e}")
    assert e_cls < 0.6 * e_attn_only
    # a shorthand rewrites the whole plan; the model object re-applies a custom plan to a fresh handle
    m.set_plan([(1, 2), (3, 3), (0, 4), (2, 0)])
    b = m.encode_image(x)
    small_b = m.encode_image(x[:3])                                      # graph-replayed small call: the CLS fix-up is captured with it
    assert torch.equal(m.encode_image(x[:3]), small_b) and (small_b - b[:3]).norm(dim=1).max() < 2e-3
    m._destroy(); m._host_sd = sd; m.to("cuda:0")
    assert m.get_plan() == [(1, 2), (3, 3), (0, 4), (2, 0)] and torch.equal(m.encode_image(x), b)
    m.set_option("comp_mlp_blocks", 1)
    assert m.get_plan() == prefix_plan(depth, 0, 1) and m.get_option("plan_custom") == 0
    lib = _lib.load()
    for bad in ((0, 6, 0), (0, 0, 5), (64, 1, 1), (-1, 1, 1)):
        assert lib.keep_set_block_precision(m._handle, *bad) == _lib.KEEP_EINVAL
    with pytest.raises(ValueError):
        m.set_plan([(0, 7)])


def test_mean_input_bias_compensation():
    """keep_calibrate_bias: the row-independent part of the dropped W_lo A_hi term (W_lo @ mean input row) folded into the bias of the plain fp16
    launches.  Full depth, bench weights, everything plain: with the compensation the cosine errors against the split-product mode are smaller than
    without (the CPU study behind it: 8 % of the all-fp16 variance, 25 % of its weight-rounding half); it is held per handle, forgotten on request
    and on reload, switched by an option, and split / compensated launches never see it."""
    sd = synth_state_dict(KEEPShape(), seed=0, text=False)
    m = KEEPModel(KEEPShape(), precision="comp", towers=("image",))
    m.auto_calibrate = False
    m.load_state_dict(sd, strict=True)
    m.to("cuda:0")
    assert m.get_option("bias_ready") == 0                        # nothing calibrated yet (auto_calibrate off)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(128, 3, 224, 224, generator=g).to(torch.bfloat16).cuda()
    bank = torch.nn.functional.normalize(torch.randn(64, 768, generator=g), dim=-1).cuda()
    m.set_precision("strict")
    ref = m.similarity(m.encode_image(x), bank)
    m.set_precision("comp")
    m.set_plan([(0, 0)] * 24)
    off = m.similarity(m.encode_image(x), bank)
    m.calibrate_bias(probe="gaussian")                 # the probe of the distribution it is evaluated on (off that distribution: tests/test_tile_families.py)
    assert m.get_option("bias_ready") == 1 and m.get_option("bias_correction") == 1
    on = m.similarity(m.encode_image(x), bank)
    rms = lambda d: float(d.pow(2).mean().sqrt())
    print(f"[bias compensation] all-plain plan, 128 tiles x 64 random directions vs split products: rms {rms(off - ref):.3e} -> {rms(on - ref):.3e} "
          f"(variance x{(rms(on - ref) / rms(off - ref)) ** 2:.3f})")
    assert rms(on - ref) < 0.99 * rms(off - ref)
    m.set_option("bias_correction", 0)
    assert torch.equal(m.similarity(m.encode_image(x), bank), off)                 # switched off: the checkpoint's biases, bit for bit
    m.set_option("bias_correction", 1)
    assert torch.equal(m.similarity(m.encode_image(x), bank), on)                  # deterministic
    m.calibrate_bias(probe="gaussian"); assert torch.equal(m.similarity(m.encode_image(x), bank), on)      # and reproducible: no atomics in the averages
    # split / compensated launches compute the W_lo term themselves: the all-split plan does not change
    m.set_plan([(1, 1)] * 24)
    sp_on = m.encode_image(x[:32])
    m.calibrate_bias(x[:0])                                                        # zero tiles: forget
    assert m.get_option("bias_ready") == 0 and torch.equal(m.encode_image(x[:32]), sp_on)
    m.set_plan([(0, 0)] * 24)
    assert torch.equal(m.similarity(m.encode_image(x), bank), off)
    with pytest.raises(ValueError):
        m.calibrate_bias(x[:3])                                                    # too few tiles to average
    m.calibrate_bias(x)                                                            # the caller's own tiles are accepted
    assert rms(m.similarity(m.encode_image(x), bank) - ref) < 0.99 * rms(off - ref)
    m.load_state_dict(sd, strict=True)                                             # reload: calibrations belong to the weights
    assert m.get_option("bias_ready") == 0


def test_fp16_range_of_the_qkv_and_hidden_stores(small, text_bank):
    """The qkv and MLP-hidden activations are stored as fp16 (max 65504).  Weights scaled so that those stores reach the upper decades
    of the range still meet the tolerance (fp16's relative precision does not depend on magnitude); weights scaled past the range do
    NOT come back as plausible numbers: conversions do not saturate, the overflow reaches the output as NaN, and the engine raises
    FloatingPointError (immediately with check_token_ids=True, at the next engine call in the lazy default)."""
    x = synth_tiles(6, seed=31)
    big = {k: v.clone() for k, v in small.items()}
    for i in range(2):
        big[f"visual.blocks.{i}.mlp.fc1.weight"] *= 8000.0            # hidden = GELU(fc1(LN(x))) reaches 3.2e4 with these weights (oracle-measured): half the range
        big[f"visual.blocks.{i}.mlp.fc1.bias"] *= 8000.0
        big[f"visual.blocks.{i}.ls2.gamma"] /= 8000.0                 # (fp32 epilogue scale) keeps the block's contribution to the residual stream where it was
        big[f"visual.blocks.{i}.attn.qkv.weight"][2048:] *= 8000.0    # V rows: stored v (and the attention output) up to 3.8e4
        big[f"visual.blocks.{i}.attn.qkv.bias"][2048:] *= 8000.0
        big[f"visual.blocks.{i}.ls1.gamma"] /= 8000.0
    with torch.no_grad():
        tok = O.vit_tokens(big, x, 2)
        ref = O.encode_image(big, x) @ text_bank.t()
    assert bool(torch.isfinite(ref).all())
    for precision in ("comp", "strict"):
        m = make_model(big, precision)
        d = (m.encode_image(x.cuda()).cpu() @ text_bank.t() - ref).abs().max().item()
        print(f"[fp16 range, in range, {precision}] max|dcos| = {d:.3e}")
        assert d < tol(precision, 2e-5)
    over = {k: v.clone() for k, v in big.items()}
    over["visual.blocks.1.mlp.fc1.weight"] *= 5.0                     # hidden reaches 1.6e5: beyond 65504 (the weight itself, max 5.3e3, is fine)
    over["visual.blocks.1.mlp.fc1.bias"] *= 5.0
    with torch.no_grad():
        assert bool(torch.isfinite(O.encode_image(over, x)).all())    # the fp32 reference is fine with these weights
    m = KEEPModel(precision="comp", towers=towers_of(over))
    m.auto_calibrate = False
    m.load_state_dict(over, strict=True)
    m.to("cuda:0")
    m.check_token_ids = True
    with pytest.raises(FloatingPointError):
        m.encode_image(x.cuda())
    m.check_token_ids = "lazy"
    out = m.encode_image(x.cuda())                                      # returns (no host synchronisation) ...
    torch.cuda.synchronize()
    with pytest.raises(FloatingPointError):
        m.similarity(out, text_bank.cuda())                             # ... and the next engine call reports it
    assert not bool(torch.isfinite(out).all())
    with pytest.raises(FloatingPointError):
        m.auto_calibrate = True
        m.calibrate()                                                   # calibration refuses such weights too
    # the operand planes themselves: a GEMM weight beyond fp16's maximum is refused at load ...
    bad = dict(small)
    bad["visual.blocks.1.attn.qkv.weight"] = small["visual.blocks.1.attn.qkv.weight"] * 1e7
    with pytest.raises(ValueError, match="65504"):
        make_model(bad, "comp")


def test_weights_below_the_fp16_window_load(small, text_bank):
    """A projection whose magnitude lives in its LayerScale (fc2 / 8000, ls2 x 8000: the same function) would put fc2's entries into fp16
    subnormals -- 1.2e-3 in cosine if stored as they are, and invisible to calibrate(), whose split-product yardstick loses the same bits.
    proj / fc2 planes are therefore pre-scaled by a power of two with LayerScale / bias adjusted (exact), and the load says so; any other
    tiny weight (a dead layer) loads with a warning, as torch's load_state_dict would (keep_inference.py:83).  Nothing is refused."""
    x = synth_tiles(6, seed=12)
    folded = dict(small)
    folded["visual.blocks.0.mlp.fc2.weight"] = small["visual.blocks.0.mlp.fc2.weight"] / 8192.0
    folded["visual.blocks.0.mlp.fc2.bias"] = small["visual.blocks.0.mlp.fc2.bias"] / 8192.0
    folded["visual.blocks.0.ls2.gamma"] = small["visual.blocks.0.ls2.gamma"] * 8192.0
    with torch.no_grad():
        want = O.encode_image(small, x)                       # powers of two: the folded weights are the same function, bit for bit in fp32
    for precision in ("strict", "comp"):
        with pytest.warns(RuntimeWarning, match="fc2.weight.*stored as 2\\^"):
            m = make_model(folded, precision)
        got = m.encode_image(x.cuda()).cpu()
        d = (got @ text_bank.t() - want @ text_bank.t()).abs().max().item()
        print(f"[fc2 / 8192 {precision}] max|dcos| = {d:.3e}")
        assert d < tol(precision, 3e-6)
    dead = dict(small)
    dead["visual.blocks.1.attn.qkv.weight"] = small["visual.blocks.1.attn.qkv.weight"] * 1e-3
    with pytest.warns(RuntimeWarning, match="qkv.weight.*subnormals"):
        m = make_model(dead, "comp")
    assert bool(torch.isfinite(m.encode_image(x.cuda())).all())


def test_text_graph_replay_covers_a_64_prompt_bank_chunk(small):
    """encode_text of 64 prompts (valid lengths 8-32 -> run at T = 32: 2048 token rows, the classifier-bank chunk of keep_amd.wsi) is
    captured once and replayed as one graph launch: same bits as the kernel-by-kernel path, on every replay, for other ids through the
    same graph, and the token-range flag still works through it."""
    m = make_model(small, "comp")
    toks = {k: v.cuda() for k, v in synth_prompts(64, 256, seed=71).items()}
    m.set_option("graphs", 0)
    ref = m.encode_text(toks)
    assert m.last_text_length == 32
    m.set_option("graphs", 1)
    for _ in range(3):
        assert torch.equal(m.encode_text(toks), ref)
    other = {k: v.cuda() for k, v in synth_prompts(64, 256, seed=72).items()}
    m.set_option("graphs", 0)
    ref2 = m.encode_text(other)
    m.set_option("graphs", 1)
    assert torch.equal(m.encode_text(other), ref2) and not torch.equal(ref2, ref)
    with torch.no_grad():
        want = O.encode_text(small, {k: v.cpu() for k, v in other.items()})
    assert (ref2.cpu() - want).abs().max() < 5e-6
    bad = {k: v.clone() for k, v in toks.items()}
    bad["input_ids"][5, 2] = 10 ** 6
    m.check_token_ids = True
    with pytest.raises(IndexError):
        m.encode_text(bad)


def test_slide_sized_population_stays_inside_the_tolerance():
    """BASELINE configs 4 / 5 put 100 000 tiles against a prompt bank; the tolerance is on EVERY cosine, so the worst of N x P errors matters and it
    grows with the population.  THREE synthetic slides of one eighth of that size (12 500 tiles, the share of one of 8 GPUs) in the calibrated plan
    against the split-product mode: all 12 500 x 64 and 12 500 x 264 cosines of every slide within 1e-4 (none over), max / rms as a Gaussian
    population predicts (that is the rule calibrate() extrapolates with), and each slide's measured rms predicts the full population inside the
    tolerance at the rule's confidence -- no slack: the rule itself carries the probe's sampling margin.  Same slide label, same screened prompt
    sets; the full size (five slides of 100 000 tiles) runs in bench.py (`configs.c4`)."""
    import bench
    from keep_amd.model import CALIBRATION_POPULATION, CONFIDENCE, exceedance_probability, max_sigmas_quantile
    sd = synth_state_dict(KEEPShape(), seed=0)
    m = make_model(sd, "comp")
    cal = m.calibration
    assert cal["population"] == CALIBRATION_POPULATION and cal["precision"] == "comp" and cal["confidence"] == CONFIDENCE
    assert cal["exceedance_probability"] <= 1.0 - CONFIDENCE + 1e-9 and cal["predicted_max_abs_dcos"] <= COS_TOL
    own = m.get_plan()
    r = bench.config4(m, torch.device("cuda", 0), n=12_500, seeds=(1000, 2000, 3000))
    h = r["headline_setting"]
    print(f"[3 x 12 500 tiles, plan {h['plan']}] worst over seeds {r['worst_over_seeds']}; scores {h['screening_scores']}; label {h['slide_label']}; "
          f"tumour ratio {h['tumour_ratio']}; calls differ on {h['tumour_tile_calls']}")
    assert m.get_plan() == own and m.get_option("precision") == 2                      # restored
    assert len(r["per_seed"]) == 3 and r["worst_over_seeds"]["over_1e-4"] == 0 and r["within_1e-4"]
    z = max_sigmas_quantile(CALIBRATION_POPULATION, CONFIDENCE)
    for st_seed in r["per_seed"]:
        for key in ("cos_vs_64_prompts", "cos_vs_264_distinct_prompts"):
            st = st_seed[key]
            assert st["max_abs"] <= COS_TOL and st["over_1e-4"] == 0
            assert st["max_over_rms"] <= st["gaussian_max_over_rms"]["quantile_0.99"]      # no heavier than Gaussian tails: what the population rule assumes
            assert st["rms"] * z <= COS_TOL                                                # ... and its prediction for the full population holds, at the rule's confidence
            assert exceedance_probability(st["rms"], CALIBRATION_POPULATION) <= 1.0 - CONFIDENCE
    assert h["slide_label_equal"] and h["screening_scores"]["same_top_n"] and h["screening_scores"]["max_abs_diff"] < 1e-4
    assert h["tumour_tile_calls"]["largest_strict_cos_margin_of_such_a_tile"] <= 2 * COS_TOL      # a call only moves on a near tie
    assert abs(h["tumour_ratio"][0] - h["tumour_ratio"][1]) <= 2e-3
