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
