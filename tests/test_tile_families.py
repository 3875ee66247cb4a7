"""Parity OFF the calibration's probe distribution (round-5 review, item 1): structured tile families -- crops of a real H & E image, Beer-Lambert
stain fields, glass background, half / half -- instead of the i.i.d. N(0,1) pixels of BASELINE config 2.

CPU tests: the generators (determinism, slide-index addressing, what the normalised pixels look like) and the statistics of the population rule.
GPU tests: the bench weights, calibrated exactly as `load_state_dict` leaves them (default probe: N(0,1) + stain fields + background + half / half;
`he_crops` -- the one family made of REAL pixels -- is held out of the probe on purpose), evaluated on every family
  (i)   64 tiles per family in 'comp' and 'strict' against the fp32 CPU oracle: cosines within 1e-4 / 5e-6, labels through `classify` equal;
  (ii)  12 500 tiles per family (one GPU's share of a 100 000-tile slide) in 'comp' against 'strict': none over 1e-4, and the slide's own per-tile
        errors predict the calibration population inside the tolerance at the rule's confidence;
  (iii) the same with the mean-input bias compensation calibrated on tiles of ANOTHER distribution (the round-5 default): what it does off its own
        distribution is measured and reported, and the rule -- which verifies whatever the compensation leaves -- still holds on every family.
"""
import math

import pytest
import torch

from keep_amd.model import (CALIBRATION_POPULATION, CONFIDENCE, max_sigmas_quantile, mixture_exceedance, mixture_max_quantile)
from keep_amd.synth import PROBE_FAMILIES, TILE_FAMILIES, calibration_probe, normalise_u8, synth_tile_family

COS_TOL = 1e-4
_ORACLE_CACHE = {}
EVAL_FAMILIES = ("he_crops", "stain_field", "background", "half")


# ------------------------------------------------------------------------------------------------ CPU: generators and statistics
@pytest.mark.parametrize("family", TILE_FAMILIES)
def test_family_tiles_depend_only_on_their_slide_index(family):
    a = synth_tile_family(family, 0, 24, "cpu", seed=11, unit=16)
    assert a.dtype == torch.uint8 and tuple(a.shape) == (24, 224, 224, 3)
    b = synth_tile_family(family, 5, 21, "cpu", seed=11, unit=16)                      # another window of the same slide, across a unit boundary
    assert torch.equal(a[5:21], b)
    assert not torch.equal(a[:16], synth_tile_family(family, 0, 16, "cpu", seed=12, unit=16))      # another slide
    assert synth_tile_family(family, 7, 7, "cpu").shape[0] == 0


def test_families_look_like_what_the_reference_transform_feeds():
    """ImageNet-normalised H & E is NOT N(0,1): non-zero channel means, spatially correlated pixels, near-constant glass."""
    stats = {}
    for family in EVAL_FAMILIES:
        x = normalise_u8(synth_tile_family(family, 0, 16, "cpu", seed=3, unit=16))
        assert x.dtype == torch.float32 and tuple(x.shape) == (16, 3, 224, 224)
        lag1 = float(((x[..., 1:] - x[..., 1:].mean()) * (x[..., :-1] - x[..., :-1].mean())).mean() / x.var().clamp_min(1e-12))
        stats[family] = (x.mean(dim=(0, 2, 3)), float(x.std(dim=(2, 3)).mean()), lag1)
    assert all(float(m.abs().max()) > 0.5 for m, _, _ in stats.values())                # channel means far from 0
    assert stats["he_crops"][2] > 0.8 and stats["stain_field"][2] > 0.8                # neighbouring pixels correlated (N(0,1) tiles: 0)
    assert stats["background"][1] < 0.2 < stats["stain_field"][1]                      # glass is near-constant inside a tile
    u8 = synth_tile_family("background", 0, 64, "cpu", seed=3, unit=64)
    flat = u8.reshape(64, -1)
    assert int((flat.min(1).values == 255).sum()) >= 4                                 # saturated tiles exist: every pixel 255
    # normalise_u8 is ToTensor + Normalize of the reference transform (keep_inference.py:91-92)
    from keep_amd.preprocess import IMAGENET_MEAN, IMAGENET_STD
    px = u8[0, 0, 0].float() / 255.0
    want = (px - torch.tensor(IMAGENET_MEAN)) / torch.tensor(IMAGENET_STD)
    assert torch.allclose(normalise_u8(u8[:1])[0, :, 0, 0], want, atol=1e-6)


def test_calibration_probe_groups():
    tiles, groups, names = calibration_probe(4, "cpu", seed=5)
    assert names == PROBE_FAMILIES and "he_crops" not in names                          # the real-image family is held out of the probe
    assert tuple(tiles.shape) == (16, 3, 224, 224) and tiles.dtype == torch.bfloat16
    assert groups.tolist() == [0] * 4 + [1] * 4 + [2] * 4 + [3] * 4
    again = calibration_probe(4, "cpu", seed=5)[0]
    assert torch.equal(tiles, again)


def test_mixture_rule_reduces_to_the_gaussian_rule_and_follows_the_hardest_tiles():
    n = CALIBRATION_POPULATION
    z = max_sigmas_quantile(n, CONFIDENCE)
    assert mixture_max_quantile([1e-5] * 7, n, CONFIDENCE) == pytest.approx(1e-5 * z, rel=1e-9)            # equal sigmas: rms x z
    assert mixture_exceedance([1e-5] * 7, n, 1e-5 * z) == pytest.approx(1.0 - CONFIDENCE, rel=1e-6)
    # a tenth of the tiles twice as hard: the maximum is theirs -- 2 sigma x the quantile of a tenth of the population (not rms x z, which is 1.14 x)
    mixed = [1e-5] * 9 + [2e-5]
    want = 2e-5 * max_sigmas_quantile(n / 10, CONFIDENCE)
    assert mixture_max_quantile(mixed, n, CONFIDENCE) == pytest.approx(want, rel=1e-3)
    rms = math.sqrt(sum(v * v for v in mixed) / 10)
    assert mixture_max_quantile(mixed, n, CONFIDENCE) > 1.5 * rms * z
    assert mixture_exceedance([], n, 1e-4) == 0.0 and mixture_exceedance([0.0, 0.0], n, 1e-4) == 0.0
    # monotone in the population and in the confidence
    assert mixture_max_quantile(mixed, 1e4) < mixture_max_quantile(mixed, 1e8) and mixture_max_quantile(mixed, n, 0.5) < mixture_max_quantile(mixed, n, 0.999)


def test_greedy_walk_serves_the_worst_group():
    """Two tile groups whose error lives in different places: every step of the walk lowers the WORST group's predicted variance, and the plan that
    ends the walk treats both."""
    from keep_amd import _lib
    from keep_amd.model import DEFAULT_KNOBS, KEEPModel
    depth = 4
    res_m = {0: 1.0, 1: 0.0, 2: 0.04, 3: 0.5, 4: [0.05] * depth}
    res_a = {0: 1.0, 1: 0.0, 2: 0.015, 3: 0.6}
    g1 = {"attn": [8.0, 0.1, 0.1, 0.1], "mlp": [1.0, 1.0, 0.5, 0.5], "floor": 0.01, "residual_mlp": res_m, "residual_attn": res_a}      # block 0's attention side
    g2 = {"attn": [0.5, 0.5, 0.5, 0.5], "mlp": [3.0, 3.0, 3.0, 3.0], "floor": 0.01, "residual_mlp": res_m, "residual_attn": res_a}      # every MLP
    walk = KEEPModel._greedy_walk({"by_group": {"a": g1, "b": g2}}, depth, DEFAULT_KNOBS)
    worst = [v for _, v in walk]
    assert worst[0] == pytest.approx(max(0.01 + 8.3 + 3.0, 0.01 + 2.0 + 12.0))
    assert all(b <= a + 1e-12 for a, b in zip(worst, worst[1:])) and worst[-1] < 0.05 * worst[0]
    first = walk[1][0]
    assert any(m != _lib.MLP_PLAIN for _, m in first) and all(a == _lib.ATTN_PLAIN for a, _ in first)      # group b governs at the start: an MLP knob goes first
    assert walk[-1][0][0][0] != _lib.ATTN_PLAIN                                                             # ... and group a's block-0 attention side is treated before the end
    # one group, read from the top level (the pre-round-6 form of the shares)
    solo = KEEPModel._greedy_walk(g1, depth, DEFAULT_KNOBS)
    assert solo[1][0][0][0] != _lib.ATTN_PLAIN


# ------------------------------------------------------------------------------------------------ GPU: the bench weights on every family
@pytest.fixture(scope="module")
def bench_model():
    from keep_amd import KEEPModel
    from keep_amd.config import KEEPShape
    from keep_amd.synth import synth_state_dict
    sd = synth_state_dict(KEEPShape(), seed=0)
    m = KEEPModel(KEEPShape())
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    return m, sd


@pytest.mark.gpu
def test_default_calibration_probes_the_mixture(bench_model):
    m, _ = bench_model
    cal = m.calibration
    assert cal["precision"] == "comp" and cal["probe_distribution"] == "mixture" and tuple(cal["probe_groups"]) == PROBE_FAMILIES
    assert cal["predicted_max_abs_dcos"] <= COS_TOL and cal["exceedance_probability"] <= 1.0 - CONFIDENCE + 1e-9
    per = {k: v for k, v in cal["per_group"].items() if isinstance(v, dict)}
    assert set(per) == set(PROBE_FAMILIES) and cal["governing_group"] in per
    z = max_sigmas_quantile(CALIBRATION_POPULATION, CONFIDENCE)
    for name, v in per.items():                                                        # EVERY group is inside the tolerance, not only the average
        assert v["effective_rms"] * z <= COS_TOL * (1 + 1e-3) and v["exceedance_probability"] <= 1.0 - CONFIDENCE + 1e-9, name
    assert not cal["bias_correction"]                                                  # the mean-input compensation is opt-in since round 6
    # ... and the plan is the CHEAPEST that verifies, not just one that does: its price in measured knob costs stays a small part of the 33 ms all-plain step
    # (a search that landed in the expensive end of the walk once kept a plan of 14 ms)
    from keep_amd import _lib
    from keep_amd.model import KNOB_COST_MS
    ca = {_lib.ATTN_PLAIN: 0.0, _lib.ATTN_PROJ_CLS: KNOB_COST_MS["attn_proj_cls"], _lib.ATTN_COMPQKV: KNOB_COST_MS["attn_compqkv"], _lib.ATTN_COMPQKV_PROJ_CLS: KNOB_COST_MS["attn_compqkv_proj_cls"],
          _lib.ATTN_SPLIT_COMPQKV: KNOB_COST_MS["attn_split_compqkv"], _lib.ATTN_SPLIT: KNOB_COST_MS["attn_split"]}
    cm = {_lib.MLP_PLAIN: 0.0, _lib.MLP_CLS: KNOB_COST_MS["mlp_cls"], _lib.MLP_COMP_W: KNOB_COST_MS["mlp_comp_w"], _lib.MLP_COMP: KNOB_COST_MS["mlp_comp"], _lib.MLP_SPLIT: 3 * KNOB_COST_MS["mlp_comp"]}
    price = sum(ca[a] + cm[mm] for a, mm in m.get_plan())
    print(f"calibrated plan {cal['plan']}: {price:.2f} ms in knob costs; governed by {cal['governing_group']}, predicted {cal['predicted_max_abs_dcos']:.3e}, {len(cal['tried'])} candidates verified")
    assert price <= 4.5
    sh = cal["variance_shares"]
    assert set(sh["by_group"]) == set(PROBE_FAMILIES) and len(sh["attn"]) == len(sh["mlp"]) == 24


@pytest.mark.gpu
@pytest.mark.parametrize("family", EVAL_FAMILIES)
def test_64_tiles_per_family_against_the_cpu_oracle(bench_model, family):
    from keep_amd.synth import synth_prompts
    from oracle import keep_oracle as O
    m, sd = bench_model
    x = synth_tile_family(family, 0, 64, "cuda:0", seed=7001)
    toks = synth_prompts(64, 64, seed=1)
    import bench
    torch.set_num_threads(min(bench.usable_cpus(), 32))            # (the box reports 256 CPUs and grants 16: torch's default oversubscribes them)
    with torch.no_grad():
        if "txt" not in _ORACLE_CACHE:
            _ORACLE_CACHE["txt"] = O.encode_text(sd, toks)
        ref_txt = _ORACLE_CACHE["txt"]
        ref = O.similarity(O.encode_image(sd, normalise_u8(x).cpu()), ref_txt)
    ref_lab, top2 = ref.argmax(1), ref.topk(2, dim=1).values
    decidable = (top2[:, 0] - top2[:, 1]) > 2e-6
    own = m.get_plan()
    try:
        for mode, tolerance in (("comp", COS_TOL), ("strict", 5e-6)):
            m.set_precision(mode)
            m.set_plan(own)
            txt = m.encode_text({k: v.cuda() for k, v in toks.items()})
            d_float = (m.similarity(m.encode_image(normalise_u8(x)), txt).cpu() - ref).abs()        # the float path the reference transform feeds
            sim, lab = m.classify(x, txt)                                                            # ... and the fused uint8 path, with the second look
            d = (sim.cpu() - ref).abs()
            print(f"[{family} / {mode}] 64 tiles x 64 prompts vs the fp32 oracle: max|dcos| {float(d.max()):.3e} (float pixels {float(d_float.max()):.3e}) "
                  f"rms {float(d.pow(2).mean().sqrt()):.3e}; {m.last_rechecked} tiles looked at twice; {int(decidable.sum())} decidable labels")
            assert float(d.max()) <= tolerance and float(d_float.max()) <= tolerance
            assert torch.equal(lab.cpu().long()[decidable], ref_lab[decidable])
    finally:
        m.set_precision("comp")
        m.set_plan(own)


def _slide_checks(r, family):
    z99 = r["cos_vs_264_distinct_prompts"]["gaussian_max_over_rms"]["quantile_0.99"]
    print(f"[{family}: {r['tiles']} tiles, plan {r['plan']}] max|dcos| {r['max_abs_dcos']:.3e} over {r['over_1e-4']}; rms vs 264 prompts "
          f"{r['cos_vs_264_distinct_prompts']['rms']:.3e} isotropic {r['isotropic_rms']:.3e}; max / rms {r['cos_vs_264_distinct_prompts']['max_over_rms']} (Gaussian 0.99: {z99}); "
          f"hardest tile / rms {r['hardest_tile_over_rms']}; population exceedance {r['population_exceedance_from_this_slide']:.2e}; "
          f"mean pairwise feature cosine {r['mean_pairwise_feature_cos_first_512']}")
    assert r["over_1e-4"] == 0 and r["max_abs_dcos"] <= COS_TOL
    # what THIS slide's per-tile errors predict for the calibration population (100 000 tiles x 264 prompts): inside the tolerance at the rule's confidence
    assert r["population_exceedance_from_this_slide"] <= 1.0 - CONFIDENCE


@pytest.mark.gpu
@pytest.mark.parametrize("family", EVAL_FAMILIES + ("mixed",))
def test_12500_tiles_per_family_stay_inside_the_tolerance(bench_model, family):
    import bench
    m, _ = bench_model
    own = m.get_plan()
    r = bench.structured_slide_parity(m, torch.device("cuda", 0), family, n=12_500, seed=7000)
    assert m.get_plan() == own and m.get_option("precision") == 2                      # restored
    _slide_checks(r, family)


@pytest.mark.gpu
def test_bias_compensation_off_its_own_distribution():
    """The mean-input compensation (`calibrate_bias`) is exact for the mean input row of the tiles it saw.  Calibrated on N(0,1) tiles (round 5's
    default) and used on glass background it ADDS error at a fixed plan; `calibrate()` run after it verifies whatever it leaves per family, so the
    tolerance holds either way -- which is why the compensation is an opt-in for callers who pass their own tiles."""
    import bench
    from keep_amd import KEEPModel
    from keep_amd.config import KEEPShape
    from keep_amd.synth import synth_state_dict
    dev = torch.device("cuda", 0)
    m = KEEPModel(KEEPShape())
    m.auto_calibrate = False
    m.load_state_dict(synth_state_dict(KEEPShape(), seed=0))
    m.to(dev).eval()
    plain = [(0, 0)] * 24
    m.set_plan(plain)
    base = {f: bench.structured_slide_parity(m, dev, f, n=2048, seed=7000)["isotropic_rms"] for f in ("background", "stain_field")}
    m.calibrate_bias(probe="gaussian")
    assert m.get_option("bias_ready") == 1
    m.set_option("bias_correction", 1)
    m.set_plan(plain)
    with_g = {f: bench.structured_slide_parity(m, dev, f, n=2048, seed=7000)["isotropic_rms"] for f in ("background", "stain_field")}
    own_tiles = normalise_u8(synth_tile_family("background", 50_000, 50_064, dev, seed=7000))
    m.calibrate_bias(tiles=own_tiles)                                                  # the caller's own distribution (other tiles of the same slide)
    m.set_plan(plain)
    with_own = bench.structured_slide_parity(m, dev, "background", n=2048, seed=7000)["isotropic_rms"]
    print(f"all-plain isotropic rms, no compensation {base}; compensation calibrated on N(0,1) tiles {with_g}; on 64 background tiles {with_own:.3e}")
    assert with_own < 0.9 * base["background"]                                         # on its own distribution it helps ...
    # ... and whatever it does elsewhere, the plan calibrate() picks AFTER it holds every family of the probe and the held-out one
    m.calibrate_bias(probe="gaussian")
    cal = m.calibrate()
    assert cal["precision"] == "comp" and cal["bias_correction"] and cal["predicted_max_abs_dcos"] <= COS_TOL
    for family in ("background", "he_crops"):
        _slide_checks(bench.structured_slide_parity(m, dev, family, n=12_500, seed=7000), family + " (bias compensation from N(0,1) tiles)")


@pytest.mark.gpu
def test_proj_cls_knob(bench_model):
    """KEEP_ATTN_PROJ_CLS: every row plain, the CLS rows' proj again as a split product on the hi + lo attention output the attention kernel keeps for
    them.  On correlated tiles it removes a good part of the attention side's share (CPU emulation: tools/attn_site_study.py); it composes with
    KEEP_MLP_CLS (one chain on the compact rows); small batches (the generic attention kernel, the small-M GEMM path) take it too."""
    from keep_amd import _lib
    m, _ = bench_model
    own = m.get_plan()
    x = synth_tile_family("he_crops", 0, 1024, "cuda:0", seed=7003)
    iso = lambda f, ref: float((f - ref).pow(2).sum(1).mean().div(768).sqrt())
    enc = lambda t: torch.cat([m.encode_image_uint8(t[i:i + 256]) for i in range(0, t.shape[0], 256)])
    try:
        m.set_precision("strict")
        ref = enc(x)
        m.set_precision("comp")
        res = {}
        for name, plan in (("plain", (_lib.ATTN_PLAIN, _lib.MLP_PLAIN)), ("proj_cls", (_lib.ATTN_PROJ_CLS, _lib.MLP_PLAIN)), ("mlp_cls", (_lib.ATTN_PLAIN, _lib.MLP_CLS)),
                           ("both_cls", (_lib.ATTN_PROJ_CLS, _lib.MLP_CLS)), ("proj_cls+mlp_split", (_lib.ATTN_PROJ_CLS, _lib.MLP_SPLIT))):
            m.set_plan([plan] * 24)
            assert m.get_plan() == [plan] * 24
            res[name] = iso(enc(x), ref)
        print("[he_crops, 1024 tiles, isotropic rms vs strict] " + "  ".join(f"{k} {v:.3e}" for k, v in res.items()))
        assert res["proj_cls"] < 0.92 * res["plain"] and res["both_cls"] < 0.85 * res["mlp_cls"] and res["both_cls"] < 0.6 * res["plain"]
        # small batches: 7 tiles (generic attention kernel, everything on the small-M path) and 40 (one lane of the big kernels)
        m.set_plan([(_lib.ATTN_PROJ_CLS, _lib.MLP_CLS)] * 24)
        for n in (1, 7, 40):
            f = m.encode_image_uint8(x[:n])
            assert iso(f, ref[:n]) < 1.5 * res["both_cls"] + 2e-6, n
    finally:
        m.set_precision("comp")
        m.set_plan(own)
