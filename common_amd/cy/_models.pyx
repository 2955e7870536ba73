# cython: language_level=3, embedsignature=True
# distutils: language = c++
"""The model handles behind model_descriptor.c_desc() (reference: microscopes/_models.pyx:16-52): one extension type per
component model, each owning the C++ `model` object of include/microscopes/models/*.hpp.  cdef-level API = the
reference's (`_thisptr`, `get()`, `create_hypers()`); the Python-level attributes (`family`, `dim`,
`get_runtime_type()`, `default_hp_bytes()`) are what common_amd.State and the tests read."""
from libcpp.memory cimport shared_ptr

from common_amd.cy._models_h cimport (
    BetaBernoulli as c_bb,
    BetaNegativeBinomial as c_bnb,
    GammaPoisson as c_gp,
    NormalInverseChiSq as c_nich,
    distributions_model as c_distributions_model,
    distributions_model_dd128 as c_distributions_model_dd128,
    distributions_model_niwv as c_distributions_model_niwv,
    bbnc_model as c_bbnc,
    dm_model as c_dm,
    noop_model as c_noop,
    model, hypers,
)
from common_amd.cy._runtime_type_h cimport runtime_type

# family tags of the C ABI (include/microscopes_hip.h msc_family)
cdef enum:
    F_BB = 0
    F_GP = 1
    F_DD = 2
    F_NICH = 3
    F_NIW = 4
    F_NOOP = 5
    F_BBNC = 6
    F_BNB = 7
    F_DM = 8


cdef class _base:
    cdef shared_ptr[model] get(self):
        return self._thisptr

    cdef shared_ptr[hypers] create_hypers(self):
        return self._thisptr.get().create_hypers()

    # -- Python-visible (not upstream): what the batched device state needs to know about the model --
    @property
    def family(self):
        """msc_family tag of the kernel family"""
        return self._family

    @property
    def dim(self):
        """dd: categories, niw: dimension, dm: categories, else 0"""
        return self._dim

    def get_runtime_type(self):
        """(primitive type, element count) of model::get_runtime_type() (distributions.hpp:398-403,497-505)"""
        cdef runtime_type t = self._thisptr.get().get_runtime_type()
        return int(t.t()), int(t.n())

    def runtime_type_str(self):
        cdef runtime_type t = self._thisptr.get().get_runtime_type()
        return t.str().decode()

    def default_hp_bytes(self):
        """create_hypers().get_hp(): the freshly created hypers' protobuf bag"""
        cdef shared_ptr[hypers] h = self.create_hypers()
        return <bytes>h.get().get_hp()

    def __repr__(self):
        return "<%s family=%d dim=%d>" % (type(self).__name__, self._family, self._dim)


cdef class _bb(_base):
    def __cinit__(self):
        self._thisptr.reset(new c_distributions_model[c_bb]())
        self._family, self._dim = F_BB, 0

cdef class _bnb(_base):
    def __cinit__(self):
        self._thisptr.reset(new c_distributions_model[c_bnb]())
        self._family, self._dim = F_BNB, 0

cdef class _gp(_base):
    def __cinit__(self):
        self._thisptr.reset(new c_distributions_model[c_gp]())
        self._family, self._dim = F_GP, 0

cdef class _nich(_base):
    def __cinit__(self):
        self._thisptr.reset(new c_distributions_model[c_nich]())
        self._family, self._dim = F_NICH, 0

cdef class _dd(_base):
    def __cinit__(self, int size):
        if size <= 0:
            raise ValueError("size must be positive")
        self._thisptr.reset(new c_distributions_model_dd128(size))
        self._family, self._dim = F_DD, size

cdef class _niw(_base):
    def __cinit__(self, int dim):
        if dim <= 0:
            raise ValueError("dim must be positive")
        self._thisptr.reset(new c_distributions_model_niwv(dim))
        self._family, self._dim = F_NIW, dim

cdef class _bbnc(_base):
    def __cinit__(self):
        self._thisptr.reset(new c_bbnc())
        self._family, self._dim = F_BBNC, 0

cdef class _dm(_base):
    def __cinit__(self, int categories):
        if categories <= 0:
            raise ValueError("categories must be positive")
        self._thisptr.reset(new c_dm(categories))
        self._family, self._dim = F_DM, categories

cdef class _noop(_base):
    def __cinit__(self):
        self._thisptr.reset(new c_noop())
        self._family, self._dim = F_NOOP, 0
