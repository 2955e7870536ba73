"""CPU: pin the oracle (oracle/msc_oracle.c) against the scipy known-answer vectors.

The double twin must reproduce scipy to ~1e-10; the float restatement (the
reference's own precision) is only checked loosely -- its error is what the 1e-6
GPU tolerance is measured against, not a gate.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.conftest import load_golden

FAMS = {"bb": orc.BB, "gp": orc.GP, "dd": orc.DD, "nich": orc.NICH, "niw": orc.NIW, "bnb": orc.BNB, "dm": orc.DM}


def _ss_from_case(fam, case, prec):
    dim = case.get("dim", 0)
    ss = np.zeros(1, dtype=orc.ss_dtype(fam, dim, prec))
    for k, v in case["ss"].items():
        ss[k][0] = np.asarray(v)
    return ss


def _close(a, b, rtol, atol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert np.all(np.abs(a - b) <= atol + rtol * np.abs(b)), (a, b, np.abs(a - b).max())


@pytest.mark.parametrize("name", sorted(FAMS))
@pytest.mark.parametrize("prec,rtol,atol", [("f64", 1e-10, 1e-9), ("f32", 2e-3, 2e-3)])
def test_score_value_and_data_match_scipy(name, prec, rtol, atol):
    fam = FAMS[name]
    for case in load_golden(name):
        dim = case.get("dim", 0)
        F = orc.Family(fam, case["hp"], dim, prec)
        ss = _ss_from_case(fam, case, prec)
        got = [F.score_value(ss, 0, np.asarray(v)) for v in case["probe"]]
        if prec == "f32" and name == "nich" and abs(case["ss"]["mean"]) > 100:
            continue  # float Welford state at |mean|~1e3 is outside any float tolerance
        _close(got, case["score_value"], rtol, atol)
        _close(F.score_data(ss, 0), case["score_data"], rtol, atol * max(1, len(case["rows"])))


@pytest.mark.parametrize("name", sorted(FAMS))
def test_sequential_add_reproduces_two_pass_suffstats(name):
    fam = FAMS[name]
    for case in load_golden(name):
        dim = case.get("dim", 0)
        F = orc.Family(fam, case["hp"], dim, "f64")
        rows = np.asarray(case["rows"])
        z = np.zeros(len(rows), dtype=np.int32)
        vals = rows.reshape((len(rows),) + orc.value_dtype(fam, dim).shape)
        ss = F.accumulate(1, vals.astype(orc.value_dtype(fam, dim).base), z)
        for k, v in case["ss"].items():
            v = np.asarray(v, dtype=np.float64)
            got = np.asarray(ss[k][0], dtype=np.float64)
            if np.issubdtype(ss.dtype[k].base, np.integer):
                assert np.array_equal(got, v), (name, k)
            else:
                _close(got, v, 1e-9, 1e-9 * (1 + np.abs(v).max()))


@pytest.mark.parametrize("name", sorted(FAMS))
def test_add_then_remove_restores_state(name):
    fam = FAMS[name]
    rng = np.random.default_rng(5)
    for case in load_golden(name):
        dim = case.get("dim", 0)
        for prec, tol in (("f64", 1e-9), ("f32", 2e-3)):
            F = orc.Family(fam, case["hp"], dim, prec)
            ss = _ss_from_case(fam, case, prec)
            before = ss.copy()
            v = np.asarray(case["probe"][int(rng.integers(len(case["probe"])))])
            if name == "nich" and prec == "f32":
                v = np.float32(case["ss"]["mean"] + 0.25)
            F.add_value(ss, 0, v)
            F.remove_value(ss, 0, v)
            for k in ss.dtype.names:
                a, b = np.asarray(ss[k][0], np.float64), np.asarray(before[k][0], np.float64)
                if np.issubdtype(ss.dtype[k].base, np.integer):
                    assert np.array_equal(a, b)
                else:
                    _close(a, b, tol, tol * (1 + np.abs(b).max()))


@pytest.mark.parametrize("name", sorted(FAMS))
def test_chain_rule_score_data_increment_equals_score_value(name):
    fam = FAMS[name]
    for case in load_golden(name):
        dim = case.get("dim", 0)
        F = orc.Family(fam, case["hp"], dim, "f64")
        ss = _ss_from_case(fam, case, "f64")
        v = np.asarray(case["probe"][0])
        sv = F.score_value(ss, 0, v)
        sd0 = F.score_data(ss, 0)
        F.add_value(ss, 0, v)
        sd1 = F.score_data(ss, 0)
        _close(sd1 - sd0, sv, 1e-8, 1e-8 * (1 + abs(sd0)))


def test_score_matrix_and_loo_agree_with_per_value_calls():
    rng = np.random.default_rng(11)
    F = orc.Family(orc.NICH, dict(mu=0., kappa=1., sigmasq=1., nu=1.), 0, "f64")
    x = rng.normal(0, 3, 64).astype(np.float32)
    z = rng.integers(0, 5, 64).astype(np.int32)
    ss = F.accumulate(5, x, z)
    M = F.score_matrix(ss, x)
    L = F.score_matrix(ss, x, z)
    for n in (0, 7, 63):
        for k in range(5):
            assert M[n, k] == F.score_value(ss, k, x[n])
            if k != z[n]:
                assert L[n, k] == M[n, k]
        t = ss.copy()
        F.remove_value(t, int(z[n]), x[n])
        assert L[n, z[n]] == F.score_value(t, int(z[n]), x[n])


def test_scores_to_probs_and_sample_discrete():
    s = np.array([-1.0, -2.0, -0.5, -30.0])
    p = orc.scores_to_probs(s)
    e = np.exp(s - s.max())
    np.testing.assert_allclose(p, e / e.sum(), rtol=1e-14)
    pf = p.astype(np.float32)
    assert orc.sample_discrete(pf, 0.0) == 0
    assert orc.sample_discrete(pf, float(pf[0]) + 1e-4) == 1
    assert orc.sample_discrete(pf, 0.999999) in (2, 3)
    assert orc.sample_discrete(np.array([0.5, 0.25], np.float32), 0.99) == 1  # falls off the end


def test_pseudocount_and_score_assignment():
    assert orc.pseudocount(3, 2.0, 4) == 3.0
    assert orc.pseudocount(0, 2.0, 4) == 0.5
    # CRP(alpha) sequential probability, group_manager.hpp:250-272
    z = [0, 0, 1, 0, 2, 1]
    alpha = 2.0
    want, counts = 0.0, {}
    for i, g in enumerate(z):
        if i:
            want += np.log((counts.get(g, 0) or alpha) / (i + alpha))
        counts[g] = counts.get(g, 0) + 1
    assert abs(orc.score_assignment(z, alpha) - want) < 1e-12


def test_philox_known_answer():
    # Random123 kat_vectors: philox4x32-10 with zero counter/key and the "pi" vector
    np.testing.assert_array_equal(orc.philox([0, 0], [0, 0, 0, 0]),
                                  np.array([0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8], np.uint32))
    np.testing.assert_array_equal(
        orc.philox([0xa4093822, 0x299f31d0], [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]),
        np.array([0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1], np.uint32))
    u = [orc.uniform01(73, 0, r) for r in range(1000)]
    assert 0.0 <= min(u) and max(u) < 1.0 and 0.4 < np.mean(u) < 0.6
