# C++ declarations of the packed-record dataviews for Cython (include/microscopes/common/recarray/dataview.hpp); the
# reference's counterpart is microscopes/common/recarray/_dataview_h.pxd.
from libcpp cimport bool as cbool
from libcpp.vector cimport vector
from libc.stdint cimport uint8_t
from libc.stddef cimport size_t

from common_amd.cy._runtime_type_h cimport runtime_type

cdef extern from "microscopes/common/recarray/dataview.hpp" namespace "microscopes::common::recarray":
    cdef cppclass row_accessor:
        row_accessor() except +
        cbool ismasked(size_t)
        const runtime_type &curtype()
        unsigned curshape()
        void bump()
        cbool end()
        unsigned tell()

    cdef cppclass dataview:
        row_accessor get() except +
        const vector[runtime_type] &types()
        size_t size()
        size_t index()
        void next()
        void reset()
        cbool end()

    cdef cppclass row_major_dataview(dataview):
        row_major_dataview(const uint8_t *, const cbool *, size_t, const vector[runtime_type] &) except +
