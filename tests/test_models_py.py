"""The Python descriptor surface (common_amd.models) held to what the reference's own tests pin:
test/test_models.py:17-51 (callable identity, niw dtype shape, pickling of all eight descriptors),
test/test_default_parameters.py (default hyper-parameters load), test/test_imports.py:32-70 (the hyper-prior scalar
functions' values), and the byte streams of the in-tree protobuf messages (tests/golden/wire.json, written by Google's
protobuf runtime from microscopes/io/schema.proto:3-46)."""
import copy
import math
import pickle

import numpy as np
import pytest
from scipy.stats import norm

import common_amd
from common_amd import wire
from common_amd.models import bb, bbnc, bnb, dd, dm, gp, nich, niw
from common_amd.scalar_functions import log_exponential, log_noninformative_beta_prior, log_normal


def test_model_callable():                                   # test/test_models.py:17-18
    assert nich() is nich and bb() is bb


def test_models_dtype():                                     # test/test_models.py:21-23
    assert niw(3).py_desc().get_np_dtype().shape == (3,)
    assert niw(5).py_desc().get_np_dtype().shape == (5,)
    assert niw(5).py_desc().get_np_dtype().base == np.float64   # microscopes/models.pyx:259: np.dtype((float, (dim,)))
    assert dm(4).py_desc().get_np_dtype().shape == (4,)


def test_models_pickle():                                    # test/test_models.py:26-51
    for model in (bb, bnb, gp, nich, dd(4), bbnc, niw(3), dm(5)):
        model1 = pickle.loads(pickle.dumps(model))
        assert model.name() == model1.name()
        if model.name() == "dd":
            assert len(model.default_hyperparams()["alphas"]) == len(model1.default_hyperparams()["alphas"]) == 4
        elif model.name() == "niw":
            assert len(model.default_hyperparams()["mu"]) == len(model1.default_hyperparams()["mu"]) == 3
        elif model.name() == "dm":
            assert model.py_desc().get_np_dtype().shape == model1.py_desc().get_np_dtype().shape == (5,)
        else:
            assert model1 is model                             # the module-level descriptors are singletons


def test_default_parameters_load():                          # test/test_default_parameters.py: Shared().load(defaults)
    for m in (bb, bnb, gp, nich, dd(5), bbnc, niw(10)):
        fam, dim = m.c_desc().family, m.c_desc().dim
        blk = common_amd.pack_hp(fam, m.default_hyperparams(), dim)
        assert blk.dtype == np.float32 and np.isfinite(blk).all()
        if m.name() != "niw":                                  # and survive the bytes round trip of the py descriptor
            raw = m.py_desc().shared_dict_to_bytes(m.default_hyperparams())
            back = m.py_desc().shared_bytes_to_dict(raw)
            for k, v in m.default_hyperparams().items():
                assert np.allclose(np.asarray(back[k], dtype=np.float64), np.asarray(v, dtype=np.float64))
    assert bb.default_hyperparams() == {"alpha": 1., "beta": 1.}                      # models.pyx:189
    assert bnb.default_hyperparams() == {"alpha": 1., "beta": 1., "r": 1}             # :200
    assert gp.default_hyperparams() == {"alpha": 1., "inv_beta": 1.}                  # :211
    assert nich.default_hyperparams() == {"mu": 0., "kappa": 1., "sigmasq": 1., "nu": 1.}   # :223
    assert niw(4).default_hyperparams()["nu"] == 4.0 and np.array_equal(niw(4).default_hyperparams()["psi"], np.eye(4))


def test_default_hyperpriors_are_the_references_callables():  # models.pyx:185-229
    assert bb.default_hyperpriors() == {("alpha", "beta"): log_noninformative_beta_prior}
    assert bbnc.default_hyperpriors() is bb.default_hyperpriors() and bnb.default_hyperpriors().keys() == bb.default_hyperpriors().keys()
    assert set(gp.default_hyperpriors()) == {"alpha", "inv_beta"} and set(nich.default_hyperpriors()) == {"mu", "sigmasq"}
    assert dd(3).default_hyperpriors() == {} and niw(2).default_hyperpriors() == {} and dm(3).default_hyperpriors() == {}
    assert abs(gp.default_hyperpriors()["alpha"](3.0) - (-3.0)) < 1e-6            # log_exponential(1.)
    assert abs(nich.default_hyperpriors()["mu"](0.7) - norm.logpdf(0.7)) < 1e-6   # log_normal(0., 1.)
    assert len(bb.default_partial_hypergrid()) == 100 * 100 and len(nich.default_partial_hypergrid()) == 100 * 100


def test_log_exponential():                                  # test/test_imports.py:32-39
    lam, x = 2., 10.
    fn = log_exponential(lam)
    assert abs(math.log(lam * math.exp(-lam * x)) - fn(x)) < 1e-5
    assert math.isinf(fn(-10.)) and fn.input_dim() == 1


def test_log_normal():                                       # test/test_imports.py:42-60
    mu, sigma2, x = 1.5, 3.2, 6.3
    ours = log_normal(mu, sigma2)
    want = norm.logpdf(x, loc=mu, scale=math.sqrt(sigma2))
    assert abs(ours(x) - want) < 1e-6
    for other in (pickle.loads(pickle.dumps(ours)), copy.copy(ours)):
        assert abs(ours._mu - other._mu) < 1e-7 and abs(ours._sigma2 - other._sigma2) < 1e-7
        assert abs(other(x) - want) < 1e-6


def test_log_noninformative_beta_prior():                    # test/test_imports.py:63-70
    alpha, beta = 0.8, 0.2
    assert log_noninformative_beta_prior.input_dim() == 2
    assert abs(log_noninformative_beta_prior(alpha, beta) - (-2.5 * np.log(alpha + beta))) < 1e-5
    assert abs(log_noninformative_beta_prior(3.0, 4.5) - (-2.5 * np.log(7.5))) < 1e-5
    assert math.isinf(log_noninformative_beta_prior(0.0, 1.0))
    assert pickle.loads(pickle.dumps(log_noninformative_beta_prior)) is log_noninformative_beta_prior


# ---- golden bytes of the in-tree messages ----------------------------------------------------------------------------
_PY_OF = {"BetaBernoulliNonConj.Shared": ("bbnc", "shared"), "BetaBernoulliNonConj.Group": ("bbnc", "group"),
          "DirichletMultinomial.Shared": ("dm", "shared"), "DirichletMultinomial.Group": ("dm", "group")}


def test_in_tree_messages_serialise_to_protobufs_own_bytes(golden):
    seen = set()
    for vec in golden("wire"):
        name, fields, want = vec["message"], vec["fields"], bytes.fromhex(vec["hex"])
        if name in _PY_OF:
            model, kind = _PY_OF[name]
            desc = (bbnc if model == "bbnc" else dm(len(fields.get("alphas", fields.get("counts", [0]))))).py_desc()
            to_bytes = desc.shared_dict_to_bytes if kind == "shared" else desc.group_dict_to_bytes
            from_bytes = desc.shared_bytes_to_dict if kind == "shared" else desc.group_bytes_to_dict
            assert to_bytes(fields) == want, (name, fields)
            back = from_bytes(want)
            for k, v in fields.items():
                assert np.array_equal(np.asarray(back[k], dtype=np.float64), np.asarray(v, dtype=np.float64)), (name, k)
        elif name == "CRP":
            assert wire.dumps("crp", fields) == want and wire.loads("crp", want) == fields
        elif name == "GroupManager":
            groups = [(int(k), v.encode()) for k, v in sorted(fields["groups"].items(), key=lambda kv: int(kv[0]))]
            assert wire.group_manager_dumps(fields["alpha"], fields["assignments"], groups) == want
            alpha, assignments, back = wire.group_manager_loads(want)
            assert alpha == fields["alpha"] and assignments == fields["assignments"] and back == groups
        elif name == "GroupData":
            assert wire.put_varint(1, fields["id"]) + wire.put_bytes(2, bytes.fromhex(fields["data_hex"])) == want
        else:
            pytest.fail("golden vector for an unknown message: " + name)
        seen.add(name)
    assert seen == {"CRP", "GroupManager", "GroupData"} | set(_PY_OF)


def test_truncated_and_malformed_bags_are_refused():
    good = wire.dumps("bbnc.group", {"p": 0.5, "heads": 300, "tails": 2})
    for bad in (good[:-1], good[:3], b"\x12\xff\xff\xff\xff\xff\xff\xff\xff\xff\x01abc", b"\x0b"):
        with pytest.raises(ValueError):
            wire.loads("bbnc.group", bad)
    with pytest.raises(ValueError):
        wire.loads("bbnc.group", wire.put_float(1, 0.5))       # required fields missing
