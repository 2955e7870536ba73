# The dataview extension types downstream Cython state objects take (reference: microscopes/common/recarray/_dataview.pxd):
# `abstract_dataview` owns a shared_ptr[dataview]; `numpy_dataview` keeps the numpy memory it points into alive.
from libcpp.memory cimport shared_ptr
from libcpp.vector cimport vector

from common_amd.cy._dataview_h cimport dataview
from common_amd.cy._runtime_type_h cimport runtime_type

cdef vector[runtime_type] get_c_types(dtype) except *


cdef class abstract_dataview:
    cdef shared_ptr[dataview] _thisptr


cdef class numpy_dataview(abstract_dataview):
    cdef readonly int _n
    cdef readonly object _data
    cdef readonly object _mask
