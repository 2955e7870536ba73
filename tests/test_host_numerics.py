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


# ---- nich BLOCKS (family_math.hpp): sum_f c1 log(1 + t_f) as c1 log(1 + P), P = prod (1 + t_f) - 1 carried relative to P ----
def _f32(x):
    return np.asarray(x, dtype=np.float32)


def _fma(a, b, c):
    """float32 fused multiply-add: the product of two floats is exact in double, one more double rounding before the float
    one (a double rounding, one case in 2^29: immaterial for a bound)"""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def _join(p, q):
    return _fma(p, q, _f32(p + q))


def _block_log1p(ts):
    """log(1 + P) from the float arithmetic of nich_block_product / nich_block_finish (the score kernels' form, the
    logarithm and the reciprocal exact: their own error is the hardware's, ~1 ulp, in either form)"""
    if len(ts) == 2:
        P = _join(ts[0], ts[1])
    elif len(ts) == 3:
        P = _join(_join(ts[0], ts[1]), ts[2])
    else:
        P = _join(_join(ts[0], ts[1]), _join(ts[2], ts[3]))
    u = _f32(np.float32(1.0) + P)
    e = _f32(P - _f32(u - np.float32(1.0)))
    return np.log(u.astype(np.float64)) + e.astype(np.float64) / u.astype(np.float64), P


def test_a_block_of_nich_features_as_one_log1p_of_their_product():
    """the block form against the sum of the features' own log1p, in double, over values from deep inside a group
    (t ~ 1e-8) to the edge of a far row (|a| = 2^15, t = 2^30: four of them multiply to 2^120, inside the float range):
    the error of c1 log1p(P) relative to ITSELF stays below 6 eps (t_f: a's two roundings and the square's half; two
    levels of joins; 5.5 eps seen in 400k draws, the median below one) -- a third of the 16.8 eps per feature that the gate on a sum of feature scores allows"""
    rng = np.random.default_rng(7)
    n = 400_000
    eps = 2.0 ** -24
    for m in (2, 3, 4):
        # a = fl(fl(x s - smu_hi) - smu_lo): the true a with two roundings of half an ulp each -- modelled as one of them
        # at random and the float conversion's own
        a_true = np.exp(rng.uniform(np.log(1e-4), np.log(2.0 ** 15), (m, n))) * rng.choice([-1.0, 1.0], (m, n))
        a = (a_true * (1.0 + rng.uniform(-0.5, 0.5, (m, n)) * eps)).astype(np.float32)    # within one eps of the truth
        ts = [_f32(a[j] * a[j]) for j in range(m)]
        got, P = _block_log1p(ts)
        assert np.isfinite(P).all() and P.max() < 2.0 ** 121
        want = sum(np.log1p(a_true[j] ** 2) for j in range(m))
        rel = np.abs(got - want) / want
        assert rel.max() < 6.0 * eps, (m, rel.max() / eps)
        # and the typical error is far smaller than the worst case
        assert np.median(rel) < 1.0 * eps


def test_the_far_row_bound_keeps_a_block_of_four_inside_the_float_range():
    """|x| <= xlim = (2^15 - max|s mu|) / max s bounds |a| = |s x - s mu| by 2^15 for EVERY group, so t <= 2^30 and
    (1 + t)^4 - 1 < 2^121: no overflow anywhere in the joins"""
    t = _f32(np.full(4, 2.0 ** 30))
    P = _join(_join(t[0:1], t[1:2]), _join(t[2:3], t[3:4]))
    assert np.isfinite(P).all() and float(P[0]) < 2.0 ** 121
    rng = np.random.default_rng(8)
    s = rng.uniform(1e-3, 50.0, 1000)
    smu = rng.normal(0, 300.0, 1000)
    xlim = (2.0 ** 15 - np.abs(smu).max()) / s.max()
    x = rng.uniform(-xlim, xlim, 200)
    assert (np.abs(s[None, :] * x[:, None] - smu[None, :]) <= 2.0 ** 15).all()
