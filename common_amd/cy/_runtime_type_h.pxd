# runtime_type / primitive types of the plugin surface (include/microscopes/common/runtime_type.hpp -> our types.hpp;
# reference declaration: microscopes/common/_runtime_type_h.pxd)
from libcpp cimport bool as cbool
from libcpp.string cimport string

cdef extern from "microscopes/common/type_info.h":
    ctypedef enum primitive_type:
        TYPE_B
        TYPE_I8
        TYPE_U8
        TYPE_I16
        TYPE_U16
        TYPE_I32
        TYPE_U32
        TYPE_I64
        TYPE_U64
        TYPE_F32
        TYPE_F64
        TYPE_NELEMS

cdef extern from "microscopes/common/runtime_type.hpp" namespace "microscopes::common":
    cdef cppclass runtime_type:
        runtime_type() except +
        runtime_type(primitive_type) except +
        runtime_type(primitive_type, unsigned) except +
        primitive_type t()
        unsigned psize()
        unsigned n()
        unsigned size()
        cbool vec()
        string str()
