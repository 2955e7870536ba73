"""CPU: numeric claims the kernels' comments make, replayed in numpy float32 (no device, no library call)."""
import numpy as np


def _recip_estimate(u):
    """the plain exponent-flip reciprocal: bits(1/u) ~ 0x7EF311C7 - bits(u)"""
    return (np.uint32(0x7EF311C7) - u.view(np.uint32)).view(np.float32)


def _log2e_over_u_estimate(u):
    """family_math.hpp nich_eval_log2_est: bits(log2e / u) ~ kLog2eOverU - bits(u)"""
    return (np.uint32(0x7F35D5C7) - u.view(np.uint32)).view(np.float32)


def test_exponent_flip_log2e_over_u_is_within_1_6_percent_for_every_u_above_one():
    """with log2e folded into the constant the estimate's piecewise-linear error straddles zero better than the plain
    reciprocal's (5.1 %): one multiplication fewer AND a third of the error"""
    rng = np.random.default_rng(0)
    mant = np.concatenate([1.0 + rng.random(1_000_000), 1.0 + np.arange(4096) / 4096.0]).astype(np.float32)
    worst = 0.0
    for e in (0, 1, 2, 7, 23, 24, 60, 100, 120):
        u = np.ldexp(mant, e).astype(np.float32)
        rel = _log2e_over_u_estimate(u).astype(np.float64) * u.astype(np.float64) / 1.4426950408889634 - 1.0
        worst = max(worst, np.abs(rel).max())
    assert worst < 0.016, worst


def test_exponent_flip_reciprocal_is_within_5_percent_for_every_u_above_one():
    rng = np.random.default_rng(0)
    mant = np.concatenate([1.0 + rng.random(1_000_000), 1.0 + np.arange(4096) / 4096.0]).astype(np.float32)
    worst = 0.0
    for e in (0, 1, 2, 7, 23, 24, 60, 100, 126):
        u = np.ldexp(mant, e).astype(np.float32)
        rel = _recip_estimate(u).astype(np.float64) * u.astype(np.float64) - 1.0
        worst = max(worst, np.abs(rel).max())
    assert worst < 0.051, worst


def test_compensated_log1p_with_the_estimate_keeps_the_error_far_below_the_hardware_logs():
    """log2(1 + t) = log2(u) + log2e (t - (u - 1)) / u with u = fl(1 + t): the term is <= 2^-24 log2e, so a 5% reciprocal
    leaves <= 4.4e-9; without the term the error is 8.6e-8 (times a posterior's (nu_n + 1) / 2 of several thousand)"""
    rng = np.random.default_rng(1)
    t = np.exp(rng.uniform(-40.0, np.log(2.0 ** 23), 2_000_000)).astype(np.float32)
    u = np.float32(1.0) + t
    r = t - (u - np.float32(1.0))                                  # exact in float32 for u < 2^24
    exact = np.log2(1.0 + t.astype(np.float64))
    l2 = np.log2(u.astype(np.float64))
    with_est = l2 + (r * _recip_estimate(u)).astype(np.float64) * 1.4426950408889634
    assert np.abs(l2 - exact).max() > 5e-8                         # what the term is there for
    assert np.abs(with_est - exact).max() < 4.4e-9


def test_compensated_log1p_of_a_square_through_two_fused_multiply_adds():
    """family_math.hpp log1p_sq_parts / nich_eval_log2_est: u = fma(a, a, 1) is 1 + a^2 rounded once and fma(a, a, -(u - 1)) is
    what that rounding dropped (u - 1 is exact), the rounding of a^2 included -- one instruction fewer than t = a * a,
    u = 1 + t, t - (u - 1), and no less accurate: emulated in float64 (a^2 is exact there), with either reciprocal"""
    rng = np.random.default_rng(2)
    a = (np.exp(rng.uniform(-20.0, np.log(2.0 ** 11.4), 2_000_000)) * rng.choice([-1.0, 1.0], 2_000_000)).astype(np.float32)
    a64 = a.astype(np.float64)
    u = (1.0 + a64 * a64).astype(np.float32)                       # fma(a, a, 1): one rounding
    r = (a64 * a64 - (u.astype(np.float64) - 1.0)).astype(np.float32)
    exact = np.log2(1.0 + a64 * a64)
    l2 = np.log2(u.astype(np.float64))
    with_rcp = l2 + (r.astype(np.float64) / u.astype(np.float64)) * 1.4426950408889634
    with_est = l2 + (r * _recip_estimate(u)).astype(np.float64) * 1.4426950408889634
    with_folded = l2 + (r * _log2e_over_u_estimate(u)).astype(np.float64)          # one fused multiply-add on the device
    assert np.abs(l2 - exact).max() > 5e-8
    assert np.abs(with_rcp - exact).max() < 1e-9
    assert np.abs(with_est - exact).max() < 4.4e-9
    assert np.abs(with_folded - exact).max() < 1.5e-9
