"""The statistics behind KEEPModel.calibrate and the plan helpers (host logic: no GPU, no library call)."""
import math

import pytest

from keep_amd import _lib
from keep_amd.model import (CALIBRATION_POPULATION, COMP_LADDER, CONFIDENCE, TOLERANCE, exceedance_probability, expected_max_sigmas, max_sigmas_gumbel,
                            max_sigmas_quantile, plan_prefix, plan_string, prefix_plan)


def test_location_mean_and_quantiles_of_the_maximum():
    # values quoted in the docstrings / DESIGN.md
    assert expected_max_sigmas(16384) == pytest.approx(4.03, abs=0.01)
    assert expected_max_sigmas(262144) == pytest.approx(4.63, abs=0.01)
    assert expected_max_sigmas(2.64e7) == pytest.approx(5.51, abs=0.01)
    assert max_sigmas_quantile(2.64e7, 0.99) == pytest.approx(6.26, abs=0.01)
    a, b = max_sigmas_gumbel(2.64e7)
    # the location is (asymptotically) the 1/e quantile: the maximum exceeds it ~63 % of the time -- not the mean, not a bound
    assert max_sigmas_quantile(2.64e7, math.exp(-1.0)) == pytest.approx(b, abs=0.03)
    assert b < b + 0.5772 / a < max_sigmas_quantile(2.64e7, 0.9) < max_sigmas_quantile(2.64e7, 0.99) < max_sigmas_quantile(2.64e7, 0.999)
    # the extreme-value asymptote sits slightly above the exact quantile (the conservative side), within 0.1 sigma
    for n in (16384, 262144, 2.64e7):
        asym = b_q = max_sigmas_gumbel(n)[1] - math.log(-math.log(0.99)) / max_sigmas_gumbel(n)[0]
        assert 0.0 < asym - max_sigmas_quantile(n, 0.99) < 0.1
    # monotone in the population
    assert max_sigmas_quantile(1e4) < max_sigmas_quantile(1e6) < max_sigmas_quantile(1e9)
    # one sample: the plain two-sided normal quantile
    assert max_sigmas_quantile(1, 0.95) == pytest.approx(1.95996, abs=1e-4)


def test_quantile_against_a_monte_carlo_maximum():
    """The quantile against a direct simulation of max |N(0,1)| over 16 384 samples (the probe's size)."""
    import torch
    g = torch.Generator().manual_seed(0)
    mx = torch.stack([torch.randn(16384, generator=g).abs().max() for _ in range(4000)])
    for q, tol in ((0.5, 0.02), (0.9, 0.03), (0.99, 0.08)):
        assert float(mx.quantile(q)) == pytest.approx(max_sigmas_quantile(16384, q), abs=tol)
    assert float((mx > expected_max_sigmas(16384)).float().mean()) == pytest.approx(1.0 - math.exp(-1.0), abs=0.04)


def test_exceedance_probability_is_the_inverse_of_the_quantile():
    for q in (0.5, 0.9, 0.99):
        rms = TOLERANCE / max_sigmas_quantile(CALIBRATION_POPULATION, q)
        assert exceedance_probability(rms, CALIBRATION_POPULATION) == pytest.approx(1.0 - q, rel=1e-6)
    # round 4's committed run: rms 1.797e-5 at the config-4 population was a coin flip, as the review computed
    assert exceedance_probability(1.797e-5, CALIBRATION_POPULATION) == pytest.approx(0.5, abs=0.03)
    assert exceedance_probability(0.0, 1e6) == 0.0 and exceedance_probability(1e-3, 1e6) == pytest.approx(1.0)
    assert CONFIDENCE == 0.99


def test_prefix_plans_round_trip():
    for full, mlp in COMP_LADDER:
        p = prefix_plan(24, min(full, 24), min(mlp, 24))
        assert len(p) == 24 and plan_prefix(p) == (min(full, 24), min(mlp, 24))
    p = prefix_plan(24, 1, 8)
    assert plan_string(p) == "attn:1" + "0" * 23 + " mlp:" + "2" * 8 + "0" * 16
    q = list(p)
    q[12] = (_lib.ATTN_PLAIN, _lib.MLP_COMP_W)
    assert plan_prefix(q) is None and plan_string(q)[34 + 12] == "3"
    q = list(p)
    q[3] = (_lib.ATTN_SPLIT, _lib.MLP_COMP)                # a split-attention block that is not part of a prefix
    assert plan_prefix(q) is None


def test_ladder_is_ordered_by_cost():
    from keep_amd.model import KNOB_COST_MS
    cost = [a * KNOB_COST_MS["attn_split"] + m * KNOB_COST_MS["mlp_comp"] for a, m in COMP_LADDER]
    assert cost == sorted(cost)
