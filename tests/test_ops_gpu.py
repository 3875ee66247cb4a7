"""Each HIP kernel against a torch CPU reference of the same operator (through the C ABI)."""
import math

import pytest
import torch

from keep_amd.ops import EPI_F16, EPI_GELU_F16, EPI_RESID_F32, EPI_RESID_LS

pytestmark = pytest.mark.gpu


def r16(t):
    return t.to(torch.float16).to(torch.float64)


def gelu64(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def rand(*shape, seed=0, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32) * std


# ------------------------------------------------------------------ fp16 MFMA GEMM
GEMM_SHAPES = [(394, 3072, 1024), (128, 128, 64), (200, 768, 768), (77, 1024, 4096), (1182, 2304, 768), (333, 4096, 1024),
               (2048, 1024, 1024), (1000, 256, 192)]


@pytest.fixture(params=[128, 256, 0], ids=["v2_256x128", "v2_256x256", "auto_splitk"])
def gemm_impl(ops, request):
    """Run the GEMM tests once per kernel variant of the product build (256-wide falls back to 128-wide for N % 256 != 0).  0 = the product's automatic choice, with the
    small-M split-K kernel taking every shape up to its 1024-row limit."""
    ops.set_option("gemm_impl", request.param)
    skinny = 320
    if request.param == 0:
        ops.set_option("gemm_skinny_m", 1024)
        ops.set_option("gemm_splitk_tiles", 256)           # every larger shape of the list goes through the K-sliced 256x256 path
    yield request.param
    ops.set_option("gemm_impl", 0)
    ops.set_option("gemm_skinny_m", skinny)
    ops.set_option("gemm_splitk_tiles", 64)


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("split", [False, True])
def test_linear_bias(ops, gemm_impl, M, N, K, split):
    a, w, b = rand(M, K, seed=1), rand(N, K, seed=2, std=0.05), rand(N, seed=3, std=0.1)
    out = ops.linear(a, w, b, EPI_F16, split).cpu().double()
    if split:
        ref = a.double() @ w.double().t() + b.double()
        assert (out - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())
    else:
        ref = r16(a) @ r16(w).t() + b.double()
        # output is rounded to fp16: half an ulp = 2^-11 relative, plus fp32 accumulation noise
        assert ((out - ref).abs() <= 5e-4 * ref.abs() + 2e-5 * math.sqrt(K)).all()


@pytest.mark.parametrize("split", [False, True])
def test_linear_gelu(ops, gemm_impl, split):
    M, N, K = 394, 4096, 1024
    a, w, b = rand(M, K, seed=4), rand(N, K, seed=5, std=0.03), rand(N, seed=6, std=0.1)
    out = ops.linear(a, w, b, EPI_GELU_F16, split).cpu().double()
    if split:
        ref = gelu64(a.double() @ w.double().t() + b.double())
        assert (out - ref).abs().max() < 2e-5
    else:
        ref = gelu64(r16(a) @ r16(w).t() + b.double())
        assert ((out - ref).abs() <= 5e-4 * ref.abs() + 1e-4).all()


@pytest.mark.parametrize("split", [False, True])
def test_linear_layerscale_residual(ops, gemm_impl, split):
    M, N, K = 394, 1024, 4096
    a, w, b = rand(M, K, seed=7), rand(N, K, seed=8, std=0.02), rand(N, seed=9, std=0.1)
    ls, resid = torch.rand(N, generator=torch.Generator().manual_seed(10)) * 0.45 + 0.05, rand(M, N, seed=11)
    out = ops.linear(a, w, b, EPI_RESID_LS, split, ls=ls, resid=resid).cpu().double()
    A, W = (a.double(), w.double()) if split else (r16(a), r16(w))
    ref = resid.double() + ls.double() * (A @ W.t() + b.double())
    assert (out - ref).abs().max() < 3e-5


@pytest.mark.parametrize("split", [False, True])
def test_linear_residual_sum(ops, gemm_impl, split):
    M, N, K = 512, 768, 3072
    a, w, b, resid = rand(M, K, seed=12), rand(N, K, seed=13, std=0.02), rand(N, seed=14, std=0.1), rand(M, N, seed=15)
    out = ops.linear(a, w, b, EPI_RESID_F32, split, resid=resid).cpu().double()
    A, W = (a.double(), w.double()) if split else (r16(a), r16(w))
    ref = resid.double() + A @ W.t() + b.double()
    assert (out - ref).abs().max() < 5e-5


# ------------------------------------------------------------------ compensated product (fp16 pass + MX-fp4 correction terms)
def _rel_rms(out, ref):
    return ((out - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


@pytest.mark.parametrize("M,N,K,epi", [(1000, 1024, 1024, EPI_F16), (394, 4096, 1024, EPI_GELU_F16), (2048, 1024, 4096, EPI_RESID_LS),
                                       (257, 256, 256, EPI_F16), (513, 512, 384, EPI_F16), (300, 768, 3072, EPI_RESID_LS)])
def test_compensated_linear_recovers_the_fp16_rounding(ops, M, N, K, epi):
    """split=2: A_hi W_hi on the fp16 pipe + Q4(A_hi) Q4(W_lo) + Q4(A_lo) Q4(W_hi) on the MX-fp4 pipe.  The correction terms
    only need 2-3 bits: the result must sit >= 4x closer to the fp64 product than the plain fp16-operand product does
    (tools/precision_study.py: ~96 % of the rounding variance removed), on every tile position incl. the ragged M edge."""
    a, w, b = rand(M, K, seed=21), rand(N, K, seed=22, std=0.04), rand(N, seed=23, std=0.1)
    ls = torch.rand(N, generator=torch.Generator().manual_seed(24)) * 0.45 + 0.05
    resid = rand(M, N, seed=25)
    kw = dict(ls=ls, resid=resid) if epi == EPI_RESID_LS else {}
    acc = a.double() @ w.double().t() + b.double()
    acc16 = r16(a) @ r16(w).t() + b.double()
    if epi == EPI_GELU_F16:
        ref, ref16 = gelu64(acc), gelu64(acc16)
    elif epi == EPI_RESID_LS:
        ref, ref16 = resid.double() + ls.double() * acc, resid.double() + ls.double() * acc16
    else:
        ref, ref16 = acc, acc16
    out = ops.linear(a, w, b, epi, 2, **kw).cpu().double()
    e_comp, e_fp16 = (out - ref).abs(), (ref16 - ref).abs()
    print(f"[comp linear {M}x{N}x{K} epi{epi}] rms err comp {e_comp.pow(2).mean().sqrt():.3e} vs fp16 operands {e_fp16.pow(2).mean().sqrt():.3e}; "
          f"max {e_comp.max():.3e} vs {e_fp16.max():.3e}")
    assert e_comp.pow(2).mean().sqrt() < 0.25 * e_fp16.pow(2).mean().sqrt()
    # no tile / row / column is left uncorrected or corrupted: blockwise rms over 32 x 32 patches
    Mb, Nb = M // 32 * 32, N // 32 * 32
    blk = lambda e: e[:Mb, :Nb].reshape(Mb // 32, 32, Nb // 32, 32).pow(2).mean(dim=(1, 3)).sqrt()
    assert (blk(e_comp) < 0.5 * blk(e_fp16).clamp_min(1e-12)).all()
    assert (e_comp[Mb:] .max() if Mb < M else torch.tensor(0.)) < 4 * e_fp16.max()


@pytest.mark.parametrize("M,N,K,epi", [(1000, 1024, 1024, EPI_F16), (394, 4096, 1024, EPI_GELU_F16), (2048, 1024, 4096, EPI_RESID_LS),
                                       (257, 256, 512, EPI_F16), (300, 768, 3072, EPI_RESID_LS)])
def test_one_term_compensated_linear_removes_the_weight_rounding(ops, M, N, K, epi):
    """split=3 (KEEP_MLP_COMP_W's GEMM): A_hi W_hi on the fp16 pipe + Q4(A_hi) Q4(W_lo) on the MX-fp4 pipe, K = 128 per chunk.  What it must
    deliver is the product of the fp16-ROUNDED activations with the UNROUNDED weights: against that reference the error has to be a small
    fraction of the weight-rounding error it removes, on every 32 x 32 patch (a wrong plane, block order or scale byte adds error instead);
    against the exact product it keeps the activation half of the fp16 rounding error -- no more, no less."""
    a, w, b = rand(M, K, seed=51), rand(N, K, seed=52, std=0.04), rand(N, seed=53, std=0.1)
    ls = torch.rand(N, generator=torch.Generator().manual_seed(54)) * 0.45 + 0.05
    resid = rand(M, N, seed=55)
    kw = dict(ls=ls, resid=resid) if epi == EPI_RESID_LS else {}
    fin = {EPI_GELU_F16: gelu64, EPI_RESID_LS: lambda acc: resid.double() + ls.double() * acc}.get(epi, lambda acc: acc)
    ref = fin(a.double() @ w.double().t() + b.double())                    # exact
    ref_w = fin(r16(a) @ w.double().t() + b.double())                      # activations rounded, weights exact: the one-term target
    ref16 = fin(r16(a) @ r16(w).t() + b.double())                          # both rounded: the plain fp16 product
    out = ops.linear(a, w, b, epi, 3, **kw).cpu().double()
    rms = lambda e: e.pow(2).mean().sqrt().item()
    e_t, w_err, e_x, e16 = out - ref_w, ref16 - ref_w, out - ref, ref16 - ref
    print(f"[one-term linear {M}x{N}x{K} epi{epi}] vs A_hi W: rms {rms(e_t):.3e} (weight-rounding error it removes: {rms(w_err):.3e}); "
          f"vs exact: {rms(e_x):.3e} (plain fp16 product {rms(e16):.3e})")
    out_round = 0.0 if epi == EPI_RESID_LS else rms(ref_w) * 2.0 ** -12   # an fp16 output cannot be closer than its own rounding
    assert rms(e_t) < 0.3 * rms(w_err) + 1.5 * out_round
    assert 0.5 * rms(e16) < rms(e_x) + out_round and rms(e_x) < 0.9 * rms(e16) + 1.5 * out_round
    if epi == EPI_RESID_LS:
        Mb, Nb = M // 32 * 32, N // 32 * 32
        blk = lambda e: e[:Mb, :Nb].reshape(Mb // 32, 32, Nb // 32, 32).pow(2).mean(dim=(1, 3)).sqrt()
        assert (blk(e_t) < 0.6 * blk(w_err).clamp_min(1e-12)).all()


@pytest.mark.parametrize("K", [256, 384, 448])
def test_one_term_compensated_linear_needs_four_chunks_of_128(ops, K):
    with pytest.raises(ValueError, match="K%128"):
        ops.linear(rand(256, K, seed=1), rand(256, K, seed=2), torch.zeros(256), EPI_F16, 3)


@pytest.mark.parametrize("K", [64, 192, 320])
def test_compensated_linear_rejects_k_that_does_not_fill_the_ring(ops, K):
    """The fp4 phase's first chunks are staged into the fp16 ring's stages as they retire: K must be a multiple of 4 K steps (128)
    and at least 256.  Anything else is refused, not silently computed another way."""
    with pytest.raises(ValueError, match="K%128"):
        ops.linear(rand(256, K, seed=1), rand(256, K, seed=2), torch.zeros(256), EPI_F16, 2)


def test_compensated_linear_is_not_transposed(ops):
    """Asymmetric operands whose fp16 rounding error is large and structured: a swapped / permuted correction operand
    (rows, k blocks, scale bytes) would add error instead of removing it."""
    M, N, K = 512, 512, 256
    g = torch.Generator().manual_seed(31)
    a = (torch.arange(M * K, dtype=torch.float32).reshape(M, K) % 977) / 977.0 + 1.0 + torch.rand(M, K, generator=g) * 1e-3
    w = ((torch.arange(N * K, dtype=torch.float32).reshape(N, K) * 7) % 1013) / 1013.0 - 0.3 + torch.rand(N, K, generator=g) * 1e-3
    ref = a.double() @ w.double().t()
    out = ops.linear(a, w, torch.zeros(N), EPI_RESID_LS, 2, ls=torch.ones(N), resid=torch.zeros(M, N)).cpu().double()
    e16 = (r16(a) @ r16(w).t() - ref).abs().max().item()
    e = (out - ref).abs().max().item()
    print(f"[comp transposition probe] max err {e:.3e} vs fp16 operands {e16:.3e}")
    assert e < 0.3 * e16


@pytest.mark.parametrize("D,F,M", [(1024, 4096, 1576), (768, 3072, 520)])
def test_mlp_block_modes(ops, D, F, M):
    """LayerNorm -> fc1 + GELU -> fc2 + LayerScale + residual through the tower's kernels.  Mode 2 consumes the MX-fp4 side
    planes written by the LayerNorm kernel and by the GELU epilogue (the producers of the compensated path): if either wrote
    a wrong layout the corrections would add noise, so the error must drop well below the plain fp16 mode's and approach the
    split mode's."""
    x = rand(M, D, seed=41)
    ln_w, ln_b = 1.0 + rand(D, seed=42, std=0.1), rand(D, seed=43, std=0.05)
    w1, b1 = rand(F, D, seed=44, std=0.025), rand(F, seed=45, std=0.02)
    w2, b2 = rand(D, F, seed=46, std=0.02), rand(D, seed=47, std=0.02)
    ls = torch.rand(D, generator=torch.Generator().manual_seed(48)) * 0.45 + 0.05
    xd = x.double()
    h = torch.nn.functional.layer_norm(xd, (D,), ln_w.double(), ln_b.double(), 1e-6)
    ref = xd + ls.double() * (gelu64(h @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double())
    err = {}
    for mode in (0, 1, 2, 3):
        out = ops.mlp(x, ln_w, ln_b, w1, b1, w2, b2, ls, mode).cpu().double()
        err[mode] = (out - ref).pow(2).mean().sqrt().item()
    print(f"[mlp D{D} M{M}] rms err: fp16 {err[0]:.3e}  split {err[1]:.3e}  compensated {err[2]:.3e}  compensated, W_lo term only {err[3]:.3e}")
    assert err[1] < 0.05 * err[0]
    assert err[2] < 0.25 * err[0]
    # mode 3 (LayerNorm and the GELU epilogue write Q(x_hi) only; both GEMMs add Q4(A_hi) Q4(W_lo)): the weight half of the rounding variance goes
    assert err[2] < err[3] < 0.85 * err[0] and err[3] > 0.55 * err[0]


@pytest.mark.parametrize("M,N,K,epi", [(16640, 1024, 1024, EPI_RESID_LS), (8300, 3072, 1024, EPI_F16), (9000, 4096, 1024, EPI_GELU_F16),
                                       (16900, 1024, 4096, EPI_RESID_LS), (66000, 1024, 256, EPI_F16)])
def test_persistent_gemm_is_bit_identical_to_one_tile_per_workgroup(M, N, K, epi):
    """gemm_persistent=1: one workgroup per CU walks the tile list and stages the next tile's first three K steps from the tail of the
    current K loop (they land under the epilogue, which then bounces through what is left of the LDS).  Same products in the same order:
    every output bit must match the one-tile-per-workgroup launch -- on full tiles, on the ragged last row tile, with 2 to 4 tiles per
    workgroup, and with the shortest K the path accepts (8 steps)."""
    a, w, b = rand(M, K, seed=51), rand(N, K, seed=52, std=0.04), rand(N, seed=53, std=0.1)
    ls = torch.rand(N, generator=torch.Generator().manual_seed(54)) * 0.45 + 0.05
    resid = rand(M, N, seed=55)
    kw = dict(ls=ls, resid=resid) if epi == EPI_RESID_LS else {}
    from keep_amd.ops import Ops
    outs = []
    for pers in (0, 1, 0, 1):
        o = Ops("cuda:0")                      # options are per handle: a fresh one per setting leaves the session's handle alone
        o.set_option("gemm_persistent", pers)
        outs.append(o.linear(a, w, b, epi, False, **kw).cpu())
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3])       # each path is deterministic
    assert torch.equal(outs[0], outs[1]), f"max diff {(outs[0] - outs[1]).abs().max().item():.3e}"
    ref = a.double() @ w.double().t() + b.double()
    if epi == EPI_RESID_LS:
        ref = resid.double() + ls.double() * ref
    elif epi == EPI_GELU_F16:
        ref = gelu64(ref)
    assert (outs[1].double() - ref).abs().max().item() < 0.05


def test_linear_is_not_transposed(ops, gemm_impl):
    """A = I-like probe with an asymmetric W: catches row/col swaps in the C fragment mapping."""
    M = N = K = 256
    a = torch.eye(M, K)
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) / 1024.0
    out = ops.linear(a, w, torch.zeros(N), EPI_F16, True).cpu()
    assert (out - w.t()).abs().max() < 1e-3


def test_gelu_epilogue_accuracy_sweep(ops, gemm_impl):
    """The branch-free erf in the epilogue: drive acc+bias over [-6, 6] through an identity GEMM."""
    M, N, K = 256, 256, 256
    a = torch.eye(M, K)
    xs = torch.linspace(-6, 6, N * K).reshape(N, K)
    out = ops.linear(a, xs, torch.zeros(N), EPI_GELU_F16, True).cpu().double()     # split: fp32-class output
    ref = gelu64(xs.double()).t()
    assert (out - ref).abs().max() < 2e-6


def test_gelu_epilogue_fp16_mode_sweep(ops, gemm_impl):
    """Single-pass mode uses the degree-6 polynomial; its error must stay under the fp16 rounding of the output."""
    M, N, K = 256, 256, 256
    xs = torch.linspace(-6, 6, N * K).reshape(N, K)
    out = ops.linear(torch.eye(M, K), xs, torch.zeros(N), EPI_GELU_F16, False).cpu().double()
    ref = gelu64(r16(xs)).t()
    assert ((out - ref).abs() <= 5e-4 * ref.abs() + 3e-6).all()


def test_linear_rejects_bad_shapes(ops):
    with pytest.raises(ValueError):
        ops.linear(rand(8, 100), rand(128, 100), rand(128))
    with pytest.raises(ValueError):
        ops.linear(rand(8, 64), rand(100, 64), rand(100))


# ------------------------------------------------------------------ attention
def attn_ref(qkv, B, T, heads, mask, round_ops):
    D = heads * 64
    x = qkv.reshape(B, T, 3, heads, 64).permute(2, 0, 3, 1, 4).double()
    if round_ops:
        x = x.to(torch.float16).double()
    q, k, v = x[0], x[1], x[2]
    s = q @ k.transpose(-1, -2) * 0.125
    if mask is not None:
        s = s + (1.0 - mask[:, None, None, :].double()) * -1e30
    p = torch.softmax(s, -1)
    return (p @ v).transpose(1, 2).reshape(B * T, D)


@pytest.mark.parametrize("B,T,heads", [(3, 197, 16), (2, 256, 12), (2, 64, 12), (1, 100, 2), (2, 128, 4), (1, 17, 1)])
@pytest.mark.parametrize("split", [False, True])
def test_attention_unmasked(ops, B, T, heads, split):
    qkv = rand(B * T, 3 * heads * 64, seed=20, std=1.5)
    out = ops.attention(qkv, B, T, heads, None, split).cpu().double()
    ref = attn_ref(qkv, B, T, heads, None, not split)
    tol = 3e-5 if split else 4e-3
    assert (out - ref).abs().max() < tol


@pytest.mark.parametrize("B", [33, 40, 64])
def test_persistent_attention_is_bit_identical(B):
    """Image-tower attention (197 tokens, 16 heads, no mask) with >= 512 (image, head) pairs runs on the persistent kernel: one 16-wave workgroup
    per CU walks the pairs, 13 waves compute one query tile each while 3 waves stage the next pair's K / V into the other half of a double buffer.
    Same arithmetic and summation order per tile as the one-pair-per-workgroup kernel: equal bit for bit, also when the pairs do not divide
    evenly over the CUs (33 x 16 = 528 pairs on 256 workgroups) -- and equal to the fp64 reference within the fp16 budget."""
    from keep_amd.ops import Ops
    T, heads = 197, 16
    qkv = rand(B * T, 3 * heads * 64, seed=24, std=1.5)
    outs = {}
    for waves in (8, 16):
        o = Ops("cuda:0")
        o.set_option("attn_waves", waves)
        outs[waves] = o.attention(qkv, B, T, heads, None, False)
    assert torch.equal(outs[8], outs[16])
    assert (outs[16][: 2 * T].cpu().double() - attn_ref(qkv[: 2 * T], 2, T, heads, None, True)).abs().max() < 4e-3


@pytest.mark.parametrize("T", [256, 40])
@pytest.mark.parametrize("split", [False, True])
def test_attention_key_padding_mask(ops, T, split):
    B, heads = 4, 12
    qkv = rand(B * T, 3 * heads * 64, seed=21, std=1.5)
    lens = [T, 9, 1, T // 2]
    mask = torch.zeros(B, T, dtype=torch.int64)
    for i, L in enumerate(lens):
        mask[i, :L] = 1
    mask[3, 3] = 0      # a hole, not just a prefix
    out = ops.attention(qkv, B, T, heads, mask, split).cpu().double()
    ref = attn_ref(qkv, B, T, heads, mask, not split)
    assert (out - ref).abs().max() < (3e-5 if split else 4e-3)


def test_attention_fully_masked_row_is_uniform(ops):
    """HF adds finfo.min to masked keys: a row with no valid key degenerates to a uniform average."""
    B, T, heads = 1, 32, 1
    qkv = rand(B * T, 3 * 64, seed=22)
    mask = torch.zeros(B, T, dtype=torch.int64)
    out = ops.attention(qkv, B, T, heads, mask, True).cpu().double()
    v = qkv[:, 128:192].double()
    assert (out - v.mean(0, keepdim=True)).abs().max() < 1e-5


def test_attention_512_tokens(ops):
    B, T, heads = 1, 512, 2
    qkv = rand(B * T, 3 * heads * 64, seed=23)
    out = ops.attention(qkv, B, T, heads, None, False).cpu().double()
    assert (out - attn_ref(qkv, B, T, heads, None, True)).abs().max() < 4e-3


@pytest.mark.parametrize("T", [512, 257, 400])
def test_split_attention_over_two_key_windows(ops, T):
    """Split products above 256 keys: K / V hi + lo of one window fill the LDS, so the keys are processed as [0, 256) and [256, T) in two launches,
    the second merging the first one's (unnormalised output, running maximum, sum) per query -- the exact softmax over all keys.  Unmasked, with
    key-padding masks ending in either window, with a hole, and with a second window that is padding only."""
    B, heads = 4, 3
    qkv = rand(B * T, 3 * heads * 64, seed=26, std=1.5)
    out = ops.attention(qkv, B, T, heads, None, True).cpu().double()
    assert (out - attn_ref(qkv, B, T, heads, None, False)).abs().max() < 3e-5
    mask = torch.ones(B, T, dtype=torch.int64)
    mask[0, 100:] = 0                       # everything in the second window (and most of the first) is padding
    mask[1, 256:] = 0                       # exactly the second window is padding
    mask[2, T - 1:] = 0
    mask[3, 7] = 0; mask[3, 300 if T > 300 else 256] = 0      # holes in both windows
    out = ops.attention(qkv, B, T, heads, mask, True).cpu().double()
    assert (out - attn_ref(qkv, B, T, heads, mask, False)).abs().max() < 3e-5
    # no valid key at all: HF adds finfo.min everywhere -> a uniform average over ALL T keys, across both windows
    none = torch.zeros(1, T, dtype=torch.int64)
    q1 = qkv[:T, :192].contiguous()
    out = ops.attention(q1, 1, T, 1, none, True).cpu().double()
    assert (out - q1[:, 128:192].double().mean(0, keepdim=True)).abs().max() < 1e-5
    with pytest.raises(ValueError):
        ops.attention(rand(600, 192), 1, 600, 1)


# ------------------------------------------------------------------ LayerNorm / sgemm / l2norm
@pytest.mark.parametrize("D,eps", [(1024, 1e-6), (768, 1e-12)])
def test_layernorm(ops, D, eps):
    x = rand(301, D, seed=30, std=3.0) + 0.7
    add = rand(301, D, seed=31)
    g, b = 1 + rand(D, seed=32, std=0.1), rand(D, seed=33, std=0.1)
    for a in (None, add):
        out = ops.layernorm(x, g, b, eps, add=a).cpu()
        ref = torch.nn.functional.layer_norm((x if a is None else x + a).double(), (D,), g.double(), b.double(), eps)
        assert (out.double() - ref).abs().max() < 5e-6


@pytest.mark.parametrize("M,N,K", [(256, 768, 1024), (5, 3, 768), (300, 64, 768), (130, 7128 // 8, 768)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_sgemm_f32(ops, M, N, K, act):
    a, b, bias = rand(M, K, seed=40), rand(N, K, seed=41, std=0.05), rand(N, seed=42, std=0.1)
    out = ops.sgemm(a, b, bias, 0.5, act).cpu().double()
    ref = 0.5 * (a.double() @ b.double().t()) + bias.double()
    ref = gelu64(ref) if act == 1 else (torch.tanh(ref) if act == 2 else ref)
    assert (out - ref).abs().max() < 2e-5


def test_sgemm_not_transposed(ops):
    a = torch.eye(64, 64)
    b = torch.arange(48 * 64, dtype=torch.float32).reshape(48, 64)
    assert torch.equal(ops.sgemm(a, b).cpu(), b.t())


def test_l2norm(ops):
    x = rand(37, 768, seed=50).cuda()
    x[5] = 0
    ref = torch.nn.functional.normalize(x.cpu(), dim=-1)
    assert (ops.l2norm_(x).cpu() - ref).abs().max() < 1e-6


def test_matrix_pipe_ceiling_probe():
    """keep_mfma_probe (measurement aid, as keep_clock_probe): MFMAs back to back on every SIMD with no memory traffic.  The pipes reach the nominal
    dense peak only with operands that do not toggle the multipliers; with N(0, 1) operands the socket's power cap holds them far below it -- the
    context bench.py prints next to its roofline fractions (profiles/r04_mfma_power_ceiling.txt: 1 590 vs 2 480 TFLOP/s)."""
    from keep_amd import KEEPModel
    m = KEEPModel()
    m._ready_device()
    rnd = m.mfma_ceiling(iters=20_000, reps=2)
    zer = m.mfma_ceiling(torch.zeros(4096, dtype=torch.float16, device="cuda"), iters=20_000, reps=2)
    print(f"matrix-pipe ceiling: random operands {rnd:.0f} TFLOP/s, zeros {zer:.0f} TFLOP/s")
    assert 500.0 < rnd < 2600.0 and 1500.0 < zer < 2600.0 and zer > rnd
    with pytest.raises(ValueError):
        m.mfma_ceiling(iters=0)
