"""Model descriptors: the Python surface of microscopes/models.pyx:96-290.

Same names (`bb`, `bnb`, `gp`, `nich`, `dd(n)`, `niw(d)`, ...), same accessor
methods and the same default hyper-parameters; `c_desc()` returns the Cython
extension object of common_amd/cy/_models.pyx (`_bb`, `_dd(size)`, ...: a
`shared_ptr[model]` behind `get()` / `create_hypers()`, as microscopes/_models.pyx:16-52),
which also carries the family tag + dimension the batched HIP state consumes.
All eight models of microscopes/models.pyx:185-290 have a HIP kernel family.
"""
import itertools as it

import numpy as np

from . import _lib as L
from . import wire
from .scalar_functions import log_exponential, log_noninformative_beta_prior, log_normal


_CY_CLASS = {L.BB: "_bb", L.BNB: "_bnb", L.GP: "_gp", L.NICH: "_nich", L.DD: "_dd", L.NIW: "_niw", L.BBNC: "_bbnc",
             L.DM: "_dm", L.NOOP: "_noop"}


class c_model(object):
    """(family, dim) of a descriptor and the factory of its Cython handle: `make()` builds the extension object of
    common_amd/cy/_models.pyx that `_base.get()` / `create_hypers()` live on (microscopes/_models.pyx:16-52)."""

    def __init__(self, family, dim=0):
        self.family, self.dim = int(family), int(dim)

    def make(self):
        try:
            from .cy import _models as cy
        except ImportError as e:
            raise ImportError("common_amd/cy/_models is not built (python common_amd/cy/build.py, or "
                              "__graft_entry__.build()): %s" % e)
        cls = getattr(cy, _CY_CLASS[self.family])
        return cls(self.dim) if self.family in (L.DD, L.NIW, L.DM) else cls()


class py_model(object):
    """dtype carrier and dict <-> protobuf-bytes converters (microscopes/models.pyx:53-94).  `dtype` is the dtype a
    caller's recarray column carries (the reference takes it from the absent library's `Value`; niw's is upstream's own
    float64); a column of any other primitive type is accepted too and converted when a state binds it."""

    def __init__(self, name, dtype):
        self._name = name
        self._dtype = np.dtype(dtype)

    def get_np_dtype(self):
        return self._dtype

    def shared_dict_to_bytes(self, raw):
        return wire.dumps(self._name + ".shared", raw)

    def shared_bytes_to_dict(self, raw):
        return wire.loads(self._name + ".shared", raw)

    def group_dict_to_bytes(self, raw):
        return wire.dumps(self._name + ".group", raw)

    def group_bytes_to_dict(self, raw):
        return wire.loads(self._name + ".group", raw)


class model_descriptor(object):
    def __init__(self, name, py_descriptor, c_descriptor, default_hyperparams, default_hyperpriors,
                 default_partial_hypergrid):
        self._name = name
        self._py_descriptor = py_descriptor
        self._c_spec = c_descriptor          # (family, dim); the Cython handle is made on first use
        self._c_descriptor = None
        self._default_hyperparams = default_hyperparams
        self._default_hyperpriors = default_hyperpriors
        self._default_partial_hypergrid = default_partial_hypergrid

    def name(self):
        return self._name

    def py_desc(self):
        return self._py_descriptor

    def c_desc(self):
        """the model's Cython handle (`_bb`, `_dd`, ...): what downstream cdef code calls `.get()` on"""
        if self._c_descriptor is None:
            self._c_descriptor = self._c_spec.make()
        return self._c_descriptor

    def default_hyperparams(self):
        return self._default_hyperparams

    def default_hyperpriors(self):
        return self._default_hyperpriors

    def default_partial_hypergrid(self):
        return self._default_partial_hypergrid

    # convenience for State(...)
    @property
    def family(self):
        return self._c_spec.family

    @property
    def dim(self):
        return self._c_spec.dim

    def _param(self):
        name = self.name()
        if name in ("dd", "dm"):
            return len(self._default_hyperparams["alphas"])
        if name == "niw":
            return len(self._default_hyperparams["mu"])
        return None

    def __reduce__(self):
        return (_reconstruct_model_descriptor, (self._name, self._param()))

    def __call__(self):
        return self  # nich() == nich


def _reconstruct_model_descriptor(name, param):
    desc = globals()[name]
    return desc if param is None else desc(param)


def _grid2(k0, k1):
    pts = np.logspace(-1, 1, num=100)
    return [{k0: a, k1: b} for a, b in it.product(pts, pts)]


def _nich_grid():
    return [{"mu": m, "sigmasq": s}
            for m, s in it.product(np.linspace(-2., 2., num=100), np.logspace(-1, 1, num=100))]


# default hyper-priors as microscopes/models.pyx:185-229 lists them (callables; test/test_imports.py:32-70 pins their values)
bb = model_descriptor("bb", py_model("bb", np.bool_), c_model(L.BB), {"alpha": 1., "beta": 1.},
                      {("alpha", "beta"): log_noninformative_beta_prior}, _grid2("alpha", "beta"))
bnb = model_descriptor("bnb", py_model("bnb", np.uint32), c_model(L.BNB), {"alpha": 1., "beta": 1., "r": 1},
                       {("alpha", "beta"): log_noninformative_beta_prior}, bb._default_partial_hypergrid)
gp = model_descriptor("gp", py_model("gp", np.uint32), c_model(L.GP), {"alpha": 1., "inv_beta": 1.},
                      {"alpha": log_exponential(1.), "inv_beta": log_exponential(1.)}, _grid2("alpha", "inv_beta"))
nich = model_descriptor("nich", py_model("nich", np.float32), c_model(L.NICH),
                        {"mu": 0., "kappa": 1., "sigmasq": 1., "nu": 1.},
                        {"mu": log_normal(0., 1.), "sigmasq": log_exponential(1.)}, _nich_grid())
bbnc = model_descriptor("bbnc", py_model("bbnc", np.bool_), c_model(L.BBNC), bb._default_hyperparams,
                        bb._default_hyperpriors, bb._default_partial_hypergrid)
noop = model_descriptor("noop", py_model("noop", np.bool_), c_model(L.NOOP), {}, {}, [])


def dd(size):
    if size <= 0:
        raise ValueError("size must be positive")
    return model_descriptor("dd", py_model("dd", np.int32), c_model(L.DD, size), {"alphas": [1.] * size}, {}, [])


def niw(dim):
    if dim <= 0:
        raise ValueError("dim must be positive")
    # (float64, as upstream: microscopes/models.pyx:259 `np.dtype((float, (dim,)))` against a TYPE_F32[dim] model -- the
    # column is converted when a state binds it, as runtime_cast::cast does per value upstream)
    return model_descriptor("niw", py_model("niw", np.dtype((np.float64, (dim,)))), c_model(L.NIW, dim),
                            {"mu": np.array([0.] * dim), "kappa": 1.0, "psi": np.eye(dim),
                             "nu": float(dim)}, {}, [])


def dm(categories):
    if categories <= 0:
        raise ValueError("categories must be positive")
    d = dd(categories)
    return model_descriptor("dm", py_model("dm", np.dtype((np.int32, (categories,)))), c_model(L.DM, categories),
                            d._default_hyperparams, d._default_hyperpriors, d._default_partial_hypergrid)
