"""The Cython boundary on the device: a downstream module (tests/cy/downstream_probe.pyx) takes the shared_ptr[model] out
of `desc.c_desc()`, creates hypers and a group through the C++ virtual API and drives add_value / score_value /
score_data per value -- the calls mixturemodel / irm make -- and the results are held against the oracle's double twin
evaluated on the group's own (float) suff-stats, integers bit for bit."""
import os
import sys

import numpy as np
import pytest

from common_amd import models, wire
from oracle import oracle as orc
from tests.gpu_helpers import TOL, audit, make_feature, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def probe(gpu_ctx):
    from common_amd.cy.build import build_module
    build_module(os.path.join(ROOT, "common_amd", "cy", "_models.pyx"))
    build_module(os.path.join(ROOT, "tests", "cy", "downstream_probe.pyx"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "cy"))
    import downstream_probe
    return downstream_probe


CASES = [("bb", models.bb, orc.BB, 0, dict(alpha=0.7, beta=1.9)),
         ("gp", models.gp, orc.GP, 0, dict(alpha=2.5, inv_beta=0.8)),
         ("bnb", models.bnb, orc.BNB, 0, dict(alpha=1.5, beta=2.0, r=3)),
         ("nich", models.nich, orc.NICH, 0, dict(mu=0.3, kappa=1.7, sigmasq=0.9, nu=2.5)),
         ("dd", models.dd(6), orc.DD, 6, dict(alphas=[0.5, 1.0, 1.5, 2.0, 0.25, 3.0])),
         ("niw", models.niw(3), orc.NIW, 3, None)]


@pytest.mark.parametrize("name,desc,family,dim,hp", CASES, ids=[c[0] for c in CASES])
def test_virtual_api_through_the_cython_handle_matches_the_oracle(probe, name, desc, family, dim, hp):
    rng = np.random.default_rng(31)
    f = make_feature(family, 40, 1, rng, dim)
    if hp is None:
        A = rng.normal(0, 1, (dim, dim))
        hp = dict(mu=rng.normal(0, 1, dim), kappa=1.3, psi=A @ A.T + dim * np.eye(dim), nu=dim + 1.5)
    vals = np.ascontiguousarray(f["values"][:-1])
    pv = np.ascontiguousarray(f["values"][-1:])
    bag = desc.py_desc().shared_dict_to_bytes(hp)
    sv, sd, ss_bag = probe.score_through_virtual_api(desc.c_desc(), bag, vals, pv)
    ss = wire.loads(name + ".group", ss_bag)
    F = orc.Family(family, hp, dim, "f64")
    seq = F.new_groups(1)                                           # the oracle's own sequential add_value, in double
    for v in vals:
        F.add_value(seq, 0, v)
    rec = np.zeros(1, dtype=orc.ss_dtype(family, dim, "f64"))       # the double twin of the group's float state
    for k in rec.dtype.names:
        if k not in ss and k == "count_sum":
            rec[k] = sum(ss["counts"])
            continue
        got = np.asarray(ss[k]).reshape(rec[k][0].shape)
        rec[k] = got
        if np.issubdtype(rec.dtype[k].base, np.integer):
            assert np.array_equal(got, seq[k][0]), (name, k)        # counts: bit-exact
        else:
            # a float field after 39 per-value updates, each rounding it once: 39 half-ulps of its largest value
            audit("cy.float_field_after_39_updates." + name + "." + k, rel_err(got, seq[k][0]).max() / (39 * 2.0 ** -24), 1.0)
    assert rel_err(sv, F.score_value(rec, 0, pv[0])) <= TOL, name
    want_sd = F.score_data(rec, 0)
    audit("cy.score_data." + name, abs(sd - want_sd) / max(1.0, abs(want_sd)), TOL)


def test_downstream_state_from_descriptors_and_a_cython_dataview(probe):
    """What mixturemodel's state.__cinit__ does: shared_ptr[model]s out of the descriptors + the C++ dataview out of
    `numpy_dataview._thisptr` -> hip::mixture_state (columns uploaded and converted at bind time); assign_all, one
    remove_value / score_value / add_value move and batched sweeps, checked against the oracle's sequential double path."""
    from common_amd.cy.build import build_module
    build_module(os.path.join(ROOT, "common_amd", "cy", "_dataview.pyx"))
    from common_amd.cy import _dataview
    rng = np.random.default_rng(77)
    N, K, alpha = 300, 5, 1.7
    descs = [models.bb, models.gp, models.nich, models.dd(5)]
    fams = [(orc.BB, 0), (orc.GP, 0), (orc.NICH, 0), (orc.DD, 5)]
    feats = [make_feature(fam, N, K, rng, dim) for fam, dim in fams]
    y = np.zeros(N, dtype=[("f%d" % i, d.py_desc().get_np_dtype()) for i, d in enumerate(descs)])
    for i, f in enumerate(feats):
        y["f%d" % i] = f["values"]
    labels = rng.integers(0, K, N) * 10 + 3                        # arbitrary labels: groups are made in order of appearance
    eid = 41
    out = probe.mixture_walk([d.c_desc() for d in descs], _dataview.numpy_dataview(y), labels, 16, eid,
                             wire.dumps("crp", {"alpha": alpha}), sweeps=3)
    first = {}
    for lab in labels:
        first.setdefault(int(lab), len(first))
    want_assign = [first[int(lab)] for lab in labels]
    assert out["assignments"] == want_assign and out["groups"] == sorted(set(want_assign) | {out["empty_gid"]})
    assert out["probe_gid"] == want_assign[eid] and out["empty_gid"] == len(first)
    G = len(first) + 1
    want, budget = np.zeros(G), np.zeros(G)
    sizes = np.bincount([a for e, a in enumerate(want_assign) if e != eid], minlength=G)
    for c, (d, (fam, dim)) in enumerate(zip(descs, fams)):
        hp = wire.loads(d.name() + ".shared", d.c_desc().default_hp_bytes())
        F = orc.Family(fam, hp, dim, "f64")
        full, without = F.new_groups(G), F.new_groups(G)
        for e in range(N):
            F.add_value(full, want_assign[e], y["f%d" % c][e])
            if e != eid:
                F.add_value(without, want_assign[e], y["f%d" % c][e])
        for which, recs in (("bags", full), ("bags_without", without)):
            for g in range(G):
                ss = wire.loads(d.name() + ".group", out[which][c][g])
                for k in recs.dtype.names:
                    if k not in ss:
                        continue
                    got = np.asarray(ss[k]).reshape(recs[k][g].shape)
                    if np.issubdtype(recs.dtype[k].base, np.integer):
                        assert np.array_equal(got, recs[k][g]), (which, c, g, k)
                    else:
                        # a float field rebuilt from <= N values in one pass (sums in double on the device, stored
                        # as float): a few float roundings of its largest value, not N of them
                        audit("cy.mixture_state.float_field." + d.name() + "." + k,
                              rel_err(got, recs[k][g]).max() / (8 * 2.0 ** -24), 1.0)
        sc = np.array([F.score_value(without, g, y["f%d" % c][eid]) for g in range(G)])
        want += sc
        budget += np.maximum(1.0, np.abs(sc))                      # the gate: 1e-6 x sum over features of max(1, |score_f|)
    prior = np.log(np.where(sizes > 0, sizes, alpha / 1.0))       # (one empty group on offer)
    assert out["score_ids"] == out["groups"]
    got = np.asarray(out["scores"])
    for g in range(G):
        audit("cy.mixture_state.score_value",
              abs(got[g] - (want[g] + prior[g])) / (budget[g] + max(1.0, abs(prior[g]))), TOL)
    assert len(out["after"]) == N and min(out["after"]) >= 0 and 1 <= out["ngroups_after"] <= 16
