#!/usr/bin/env python
"""Generate the known-answer vectors under tests/golden/ (SURVEY.md section 8c).

The real reference cannot be built or imported here (its arithmetic lives in the
absent third-party `distributions` library), so these vectors come from scipy
closed forms evaluated in float64 -- an implementation independent of both the
oracle (oracle/msc_oracle.c) and the HIP kernels:

  bb    scipy.stats.betabinom(1, a+h, b+t).logpmf          / betaln ratio
  gp    scipy.stats.nbinom(n=a', p=b'/(1+b')).logpmf        / chain rule
  dd    closed form log((a_v+c_v)/(sum a + n))              / gammaln closed form
  nich  scipy.stats.t(df, loc, scale).logpdf                / chain rule
  niw   scipy.stats.multivariate_t(loc, shape, df).logpdf   / chain rule
  bnb   scipy.stats.betanbinom(r, a', b').logpmf            / chain rule minus the data-only binomial terms
  dm    scipy.stats.dirichlet_multinomial(alpha+n, X).logpmf / chain rule   (src/models/dm.cpp:39-97)

Suff-stats are computed from the raw rows with two-pass numpy formulas, not with
the oracle's sequential updates.  Hyperparameters and data are rounded to float32
first so every implementation sees identical inputs.

Run:  python tests/golden/make_golden.py     (rewrites tests/golden/*.json)
"""
import json
import os

import numpy as np
from scipy import special, stats

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = lambda x: np.asarray(x, dtype=np.float32).astype(np.float64)


def _dump(name, obj):
    def enc(o):
        if isinstance(o, np.ndarray):
            return o.tolist()
        if isinstance(o, (np.floating,)):
            return float(o)
        if isinstance(o, (np.integer,)):
            return int(o)
        raise TypeError(type(o))
    with open(os.path.join(HERE, name), "w") as fh:
        json.dump(obj, fh, default=enc, indent=1, sort_keys=True)
        fh.write("\n")


def golden_bb(rng):
    cases = []
    for _ in range(6):
        a, b = f32(rng.uniform(0.1, 5.0, 2))
        rows = rng.random(int(rng.integers(0, 40))) < rng.random()
        h, t = int(rows.sum()), int((~rows).sum())
        sv = [float(stats.betabinom(1, a + h, b + t).logpmf(v)) for v in (0, 1)]
        sd = float(special.betaln(a + h, b + t) - special.betaln(a, b))
        cases.append(dict(hp=dict(alpha=a, beta=b), rows=rows.astype(int), ss=dict(heads=h, tails=t),
                          probe=[0, 1], score_value=sv, score_data=sd))
    return cases


def golden_gp(rng):
    cases = []
    for _ in range(6):
        a, ib = f32(rng.uniform(0.2, 4.0, 2))
        rows = rng.poisson(rng.gamma(2.0, 2.0), int(rng.integers(0, 40))).astype(np.int64)
        cnt, sm = len(rows), int(rows.sum())
        log_prod = float(special.gammaln(rows + 1.0).sum())
        pa, pb = a + sm, ib + cnt
        probe = [0, 1, 2, 5, 17, 60, 300]
        sv = [float(stats.nbinom(n=pa, p=pb / (1.0 + pb)).logpmf(v)) for v in probe]
        # chain rule for the marginal likelihood
        sd, s, c = 0.0, 0, 0
        for v in rows:
            sd += float(stats.nbinom(n=a + s, p=(ib + c) / (1.0 + ib + c)).logpmf(v))
            s += int(v)
            c += 1
        cases.append(dict(hp=dict(alpha=a, inv_beta=ib), rows=rows,
                          ss=dict(count=cnt, sum=sm, log_prod=log_prod), probe=probe,
                          score_value=sv, score_data=sd))
    return cases


def golden_dd(rng):
    cases = []
    for dim in (2, 5, 32, 128):
        alphas = f32(rng.uniform(0.2, 3.0, dim))
        rows = rng.integers(0, dim, int(rng.integers(0, 200)))
        counts = np.bincount(rows, minlength=dim)
        n = int(counts.sum())
        probe = sorted(set([0, dim - 1, int(rng.integers(0, dim))]))
        sv = [float(np.log((alphas[v] + counts[v]) / (alphas.sum() + n))) for v in probe]
        sd = float((special.gammaln(alphas + counts) - special.gammaln(alphas)).sum()
                   + special.gammaln(alphas.sum()) - special.gammaln(alphas.sum() + n))
        cases.append(dict(dim=dim, hp=dict(alphas=alphas), rows=rows,
                          ss=dict(count_sum=n, counts=counts), probe=probe, score_value=sv,
                          score_data=sd))
    return cases


def golden_bnb(rng):
    cases = []
    for _ in range(6):
        al, be = f32(rng.uniform(0.3, 4.0, 2))
        r = int(rng.integers(1, 6))
        rows = rng.negative_binomial(r, rng.uniform(0.2, 0.8), int(rng.integers(0, 40))).astype(np.int64)
        cnt, sm = len(rows), int(rows.sum())
        a, b = al + r * cnt, be + sm
        probe = [0, 1, 2, 5, 17, 60, 300]
        sv = [float(stats.betanbinom(r, a, b).logpmf(v)) for v in probe]
        # chain rule, minus the terms the suff-stats {count, sum} cannot carry: sum_i log C(r+v_i-1, v_i)
        sd, s, c = 0.0, 0, 0
        for v in rows:
            sd += float(stats.betanbinom(r, al + r * c, be + s).logpmf(v))
            sd -= float(special.gammaln(r + v) - special.gammaln(v + 1.0) - special.gammaln(r))
            s += int(v)
            c += 1
        cases.append(dict(hp=dict(alpha=al, beta=be, r=r), rows=rows, ss=dict(count=cnt, sum=sm),
                          probe=probe, score_value=sv, score_data=sd))
    return cases


def _dm_logpmf(alpha, x):
    n = int(np.sum(x))
    return 0.0 if n == 0 else float(stats.dirichlet_multinomial(alpha, n).logpmf(x))   # scipy rejects n = 0


def golden_dm(rng):
    cases = []
    for dim in (2, 3, 8, 40):
        alphas = f32(rng.uniform(0.2, 3.0, dim))
        nrows = int(rng.integers(0, 30))
        rows = rng.multinomial(int(rng.integers(1, 40)), rng.dirichlet(np.ones(dim)), size=nrows).astype(np.int64)
        rows = rows.reshape(nrows, dim)
        counts = rows.sum(axis=0) if nrows else np.zeros(dim, dtype=np.int64)
        ratio = float((special.gammaln(rows.sum(axis=1) + 1.0) - special.gammaln(rows + 1.0).sum(axis=1)).sum())
        probe = [rng.multinomial(n, rng.dirichlet(np.ones(dim))).astype(np.int64) for n in (0, 1, 7, 150)]
        sv = [_dm_logpmf(alphas + counts, x) for x in probe]
        sd, c = 0.0, np.zeros(dim)
        for x in rows:
            sd += _dm_logpmf(alphas + c, x)
            c += x
        cases.append(dict(dim=dim, hp=dict(alphas=alphas), rows=rows, ss=dict(counts=counts, ratio=ratio),
                          probe=probe, score_value=sv, score_data=sd))
    return cases


def _nich_post(hp, n, mean, ctv):
    mu, kappa, sigmasq, nu = hp
    kn = kappa + n
    mun = (kappa * mu + n * mean) / kn
    nun = nu + n
    s2n = (nu * sigmasq + ctv + n * kappa * (mu - mean) ** 2 / kn) / nun
    return mun, kn, s2n, nun


def _nich_logpdf(hp, rows, x):
    n = len(rows)
    mean = float(np.mean(rows)) if n else 0.0
    ctv = float(((rows - mean) ** 2).sum()) if n else 0.0
    mun, kn, s2n, nun = _nich_post(hp, n, mean, ctv)
    scale = np.sqrt(s2n * (kn + 1.0) / kn)
    return float(stats.t(df=nun, loc=mun, scale=scale).logpdf(x))


def golden_nich(rng):
    cases = []
    for i in range(8):
        hp = f32([rng.normal(0, 2), rng.uniform(0.1, 3), rng.uniform(0.2, 4), rng.uniform(0.5, 5)])
        n = int(rng.integers(0, 60)) if i else 0
        centre = rng.normal(0, 10) if i % 2 else rng.normal(0, 1000)
        rows = f32(rng.normal(centre, rng.uniform(0.1, 3), n))
        mean = float(rows.mean()) if n else 0.0
        ctv = float(((rows - mean) ** 2).sum()) if n else 0.0
        probe = f32([centre, centre + 0.5, centre - 3.0, centre + 40.0, 0.0])
        sv = [_nich_logpdf(hp, rows, x) for x in probe]
        sd = sum(_nich_logpdf(hp, rows[:j], rows[j]) for j in range(n))
        cases.append(dict(hp=dict(mu=hp[0], kappa=hp[1], sigmasq=hp[2], nu=hp[3]), rows=rows,
                          ss=dict(count=n, mean=mean, count_times_variance=ctv), probe=probe,
                          score_value=sv, score_data=float(sd)))
    return cases


def _niw_logpdf(d, mu, kappa, psi, nu, rows, x):
    n = len(rows)
    sx = rows.sum(0) if n else np.zeros(d)
    sxx = rows.T @ rows if n else np.zeros((d, d))
    kn, nun = kappa + n, nu + n
    mun = (kappa * mu + sx) / kn
    psin = psi + sxx + kappa * np.outer(mu, mu) - kn * np.outer(mun, mun)
    dof = nun - d + 1.0
    shape = psin * (kn + 1.0) / (kn * dof)
    return float(stats.multivariate_t(loc=mun, shape=shape, df=dof).logpdf(x))


def golden_niw(rng):
    cases = []
    for d in (1, 2, 3, 8, 32):
        mu = f32(rng.normal(0, 1, d))
        kappa = float(f32(rng.uniform(0.5, 2)))
        A = rng.normal(0, 1, (d, d))
        psi = f32(A @ A.T / d + np.eye(d))
        psi = (psi + psi.T) / 2
        nu = float(f32(d + rng.uniform(0, 3)))
        n = int(rng.integers(1, 50))
        B = rng.normal(0, 1, (d, d)) / np.sqrt(d)
        rows = f32(rng.normal(0, 1, (n, d)) @ B + rng.normal(0, 3, d))
        probe = f32(np.stack([rows.mean(0), rows.mean(0) + 1.0, np.zeros(d)]))
        sv = [_niw_logpdf(d, mu, kappa, psi, nu, rows, x) for x in probe]
        sd = sum(_niw_logpdf(d, mu, kappa, psi, nu, rows[:j], rows[j]) for j in range(n))
        cases.append(dict(dim=d, hp=dict(mu=mu, kappa=kappa, psi=psi, nu=nu), rows=rows,
                          ss=dict(count=n, sum_x=rows.sum(0), sum_xxT=rows.T @ rows), probe=probe,
                          score_value=sv, score_data=float(sd)))
    return cases


def reference_fixtures():
    """Data the reference's own tests hold for the layout / bookkeeping side of the path."""
    return dict(
        # test/test_dataview.py:31-46
        recarray_bool_f64=dict(dtype=[["f0", "bool"], ["f1", "float64"]],
                               rows=[[False, 32.0], [True, 943.0], [False, -32.0]],
                               offsets=[0, 1], rowsize=9, maskrowsize=2, sum_f0=1),
        # test/test_dataview.py:49-61
        recarray_subarray=dict(dtype=[["f0", "int32"], ["f1", "float32", 2]],
                               rows=[[1, [2.0, 3.0]], [-1, [-3.0, 54.0]]],
                               offsets=[0, 4], rowsize=12, maskrowsize=3),
        # test/test_dataview.py:64-75
        recarray_masked=dict(dtype=[["f%d" % i, "bool"] for i in range(5)],
                             rows=[[True, False, True, True, True]],
                             mask=[[False, False, True, True, True]],
                             offsets=[0, 1, 2, 3, 4], rowsize=5, maskrowsize=5),
        # test/cxx/test_group_manager.cpp:22-66
        group_manager=dict(alpha=2.0, nentities=10, created=7, deleted=[3],
                           assignments=[-1, 2, 1, 0, 6, 1, 2, -1, -1, 5],
                           groups=[0, 1, 2, 4, 5, 6],
                           counts={"0": 1, "1": 2, "2": 2, "4": 0, "5": 1, "6": 1},
                           empty=[4]),
        # microscopes/models.pyx:185-290 default hyper-parameters
        default_hyperparams=dict(bb=dict(alpha=1.0, beta=1.0), bnb=dict(alpha=1.0, beta=1.0, r=1),
                                 gp=dict(alpha=1.0, inv_beta=1.0),
                                 nich=dict(mu=0.0, kappa=1.0, sigmasq=1.0, nu=1.0)),
    )


def main():
    rng = np.random.default_rng(73)  # bin/perf_group.cpp:19 seeds with 73
    _dump("bb.json", golden_bb(rng))
    _dump("gp.json", golden_gp(rng))
    _dump("dd.json", golden_dd(rng))
    _dump("nich.json", golden_nich(rng))
    _dump("niw.json", golden_niw(rng))
    _dump("bnb.json", golden_bnb(rng))   # appended after the round-1 families so their vectors stay as they were
    _dump("dm.json", golden_dm(rng))
    _dump("reference_fixtures.json", reference_fixtures())


if __name__ == "__main__":
    main()
