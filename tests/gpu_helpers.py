"""Shared builders for the -m gpu parity tests: seeded synthetic features, the
float state the device is given, and the same state widened for the double twin."""
import numpy as np

from oracle import oracle as orc

TOL = 1e-6  # north_star: float log-scores within 1e-6 relative of the double evaluation


_AUDIT = {}


def audit(name, err, gate):
    """every gate that is not the plain TOL goes through here: the largest error seen per gate is kept and, with
    MSC_TOL_AUDIT=<file> in the environment, written out when the session ends (profiles/r03_tolerance_audit.json is
    such a run) -- a gate is a budget with a reason, and this is the evidence it is held against"""
    err = float(err)
    rec = _AUDIT.setdefault(name, {"max_err": 0.0, "gate": float(gate), "checks": 0})
    rec["max_err"] = max(rec["max_err"], err)
    rec["gate"] = max(rec["gate"], float(gate))
    rec["checks"] += 1
    assert err <= gate, (name, err, gate)


def _dump_audit():
    import json
    import os
    path = os.environ.get("MSC_TOL_AUDIT")
    if path and _AUDIT:
        with open(path, "w") as fh:
            json.dump(_AUDIT, fh, indent=1, sort_keys=True)


import atexit  # noqa: E402
atexit.register(_dump_audit)


def rel_err(got, want):
    """|got - want| / max(1, |want|): relative for |score| >= 1, absolute below."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    return np.abs(got - want) / np.maximum(1.0, np.abs(want))


def make_feature(family, N, K, rng, dim=0, hp=None):
    """-> dict(family, dim, hp, values, np_dtype) with group structure in the data."""
    z_true = rng.integers(0, K, N)
    if family in (orc.BB, orc.BBNC):
        p = rng.random(K)
        vals = (rng.random(N) < p[z_true]).astype(np.bool_)
        hp = hp or dict(alpha=1.0, beta=1.0)
        dt = np.bool_
    elif family == orc.GP:
        lam = rng.gamma(2.0, 2.0, K)
        vals = rng.poisson(lam[z_true]).astype(np.uint32)
        hp = hp or dict(alpha=1.0, inv_beta=1.0)
        dt = np.uint32
    elif family == orc.BNB:
        pr = rng.uniform(0.15, 0.8, K)
        r = int((hp or {}).get("r", 3))
        vals = rng.negative_binomial(r, pr[z_true]).astype(np.uint32)
        hp = hp or dict(alpha=1.5, beta=2.0, r=3)
        dt = np.uint32
    elif family == orc.DM:
        th = rng.dirichlet(np.ones(dim) * 0.7, K)
        tot = rng.integers(0, 40, N)
        vals = np.array([rng.multinomial(t, th[g]) for t, g in zip(tot, z_true)], dtype=np.int32).reshape(N, dim)
        hp = hp or dict(alphas=list(rng.uniform(0.3, 2.0, dim).astype(np.float32)))
        dt = np.dtype((np.int32, (dim,)))
    elif family == orc.DD:
        th = rng.dirichlet(np.ones(dim), K)
        cdf = th.cumsum(1)
        vals = (rng.random(N)[:, None] > cdf[z_true]).sum(1).clip(0, dim - 1).astype(np.int32)
        hp = hp or dict(alphas=[1.0] * dim)
        dt = np.int32
    elif family == orc.NICH:
        centres = rng.normal(0, 10, K)
        vals = (centres[z_true] + rng.normal(0, 1, N)).astype(np.float32)
        hp = hp or dict(mu=0.0, kappa=1.0, sigmasq=1.0, nu=1.0)
        dt = np.float32
    elif family == orc.NIW:
        centres = rng.normal(0, 3, (K, dim))
        A = rng.normal(0, 1, (K, dim, dim)) / np.sqrt(dim)
        vals = (centres[z_true] + np.einsum("nij,nj->ni", A[z_true], rng.normal(0, 1, (N, dim)))).astype(np.float32)
        hp = hp or dict(mu=np.zeros(dim), kappa=1.0, psi=np.eye(dim), nu=float(dim))
        dt = np.dtype((np.float32, (dim,)))
    else:
        raise ValueError(family)
    return dict(family=family, dim=dim, hp=hp, values=vals, np_dtype=dt)


def recarray_of(features):
    dt = np.dtype([("f%d" % i, f["np_dtype"]) for i, f in enumerate(features)])
    arr = np.zeros(len(features[0]["values"]), dtype=dt)
    for i, f in enumerate(features):
        arr["f%d" % i] = f["values"]
    return arr


def state_from_assignment(features, K, z):
    """Suff-stats of every feature given z: accumulate in double, round the float
    fields to float (what the device stores), widen again for the twin.
    -> list of (Family64, ss64_of_float_state, ss32)."""
    out = []
    for f in features:
        F = orc.Family(f["family"], f["hp"], f["dim"], "f64")
        init = None
        if f["family"] == orc.BBNC:     # the group's explicit p is state, drawn once per group
            init = np.zeros(K, dtype=orc.ss_dtype(orc.BBNC, 0, "f64"))
            init["p"] = np.random.default_rng(K).uniform(0.05, 0.95, K).astype(np.float32)
        ss64 = F.accumulate(K, f["values"], z, ss_init=init)
        ss32 = orc.narrow_ss(f["family"], ss64, f["dim"])
        out.append((F, orc.widen_ss(f["family"], ss32, f["dim"]), ss32))
    return out


def load_state(st, feats_state):
    """push hp + float suff-stats of every feature into a common_amd.State"""
    import common_amd
    for i, (F, _, ss32) in enumerate(feats_state):
        st.set_hp(i, F.hp)
        rec = np.zeros(ss32.shape[0], dtype=common_amd.ss_dtype(F.family, F.dim))
        for name in rec.dtype.names:
            rec[name] = ss32[name]
        st.set_ss(i, rec)


def oracle_scores(features, feats_state, z=None, rows=None):
    """double-twin [n, K] matrix summed over features (optionally leave-one-out)."""
    total = None
    for f, (F, ss64, _) in zip(features, feats_state):
        v = f["values"] if rows is None else f["values"][rows]
        zz = None if z is None else (z if rows is None else z[rows])
        m = F.score_matrix(ss64, v, zz)
        total = m if total is None else total + m
    return total


def crp_prior_matrix(counts, alpha, z=None):
    """[n, K] log pseudocounts (group_manager.hpp:274-283); with z: row n removed first."""
    counts = np.asarray(counts, dtype=np.int64)
    K = counts.shape[0]
    ne = int((counts == 0).sum())
    if z is None:
        base = np.where(counts > 0, np.log(np.maximum(counts, 1)), np.log(alpha / max(ne, 1)))
        return base[None, :]
    out = np.empty((len(z), K))
    for n, g in enumerate(z):
        c = counts.copy()
        if g >= 0:
            c[g] -= 1
        ne_n = int((c == 0).sum())
        out[n] = np.where(c > 0, np.log(np.maximum(c, 1)), np.log(alpha / max(ne_n, 1)))
    return out
