# The extension types a model_descriptor's c_desc() returns (reference: microscopes/_models.pxd:4-31): `_base` owns a
# shared_ptr[model]; downstream cdef code calls get() / create_hypers() on it.
from libcpp.memory cimport shared_ptr

from common_amd.cy._models_h cimport model, hypers


cdef class _base:
    cdef shared_ptr[model] _thisptr
    cdef int _family
    cdef unsigned _dim
    cdef shared_ptr[model] get(self)
    cdef shared_ptr[hypers] create_hypers(self)

cdef class _bb(_base):
    pass

cdef class _bnb(_base):
    pass

cdef class _gp(_base):
    pass

cdef class _nich(_base):
    pass

cdef class _dd(_base):
    pass

cdef class _bbnc(_base):
    pass

cdef class _niw(_base):
    pass

cdef class _dm(_base):
    pass

cdef class _noop(_base):
    pass
