# cython: language_level=3, embedsignature=True
# distutils: language = c++
"""numpy structured arrays as packed-record dataviews (reference: microscopes/common/recarray/_dataview.pyx:61-92,
microscopes/common/_dataview.pyx:6-53): `numpy_dataview(npd)` wraps a 1-D structured (optionally masked) array in the
C++ `row_major_dataview` of include/microscopes/common/recarray/dataview.hpp -- no copy, the extension object keeps the
array alive -- which downstream state objects read through `_thisptr` (and `hip::mixture_state` uploads to the device
as typed columns, `row_major_dataview::to_device`).  Same argument rules and messages as the reference's."""
import hashlib

import numpy as np

from libc.stdint cimport uint8_t
from libcpp cimport bool as cbool
from libcpp.vector cimport vector

from common_amd.cy._dataview_h cimport dataview, row_major_dataview, row_accessor
from common_amd.cy._runtime_type_h cimport runtime_type, primitive_type

from common_amd.runtime import runtime_types_of


cdef vector[runtime_type] get_c_types(dtype) except *:
    """structured dtype -> vector[runtime_type] (microscopes/common/_dataview.pyx:27-44)"""
    cdef vector[runtime_type] out
    dt = np.dtype(dtype)
    for i, (t, n) in enumerate(runtime_types_of(dt)):
        sub = dt[i].subdtype
        if sub is None:
            out.push_back(runtime_type(<primitive_type> <int> t))
        else:
            out.push_back(runtime_type(<primitive_type> <int> t, <unsigned> n))
    return out


cdef class abstract_dataview:
    def __iter__(self):
        raise NotImplementedError("rows are read through the C++ dataview (or common_amd.DataView on the device)")

    def size(self):
        raise NotImplementedError("abstract")

    def digest(self):
        h = hashlib.sha1()
        typ = type(self)
        h.update((typ.__module__ + '.' + typ.__name__).encode())
        self._digest(h)
        return h


cdef class numpy_dataview(abstract_dataview):
    def __cinit__(self, npd):
        if npd is None:
            raise ValueError("npd is None")
        if len(npd.shape) != 1:
            raise ValueError("1D (structural) arrays only")
        self._n = npd.shape[0]
        dtype = npd.dtype
        if len(dtype) == 0:
            raise ValueError("structural arrays only")
        if hasattr(npd, 'mask'):
            self._data = np.ascontiguousarray(npd.data)
            self._mask = np.ascontiguousarray(np.ma.getmaskarray(npd))
        else:
            self._data = np.ascontiguousarray(npd)
            self._mask = None
        cdef vector[runtime_type] ctypes = get_c_types(dtype)
        cdef const uint8_t[::1] raw = self._data.reshape(-1).view(np.uint8) if self._n else np.zeros(1, np.uint8)
        cdef const uint8_t[::1] mraw
        if self._mask is not None and self._n:
            mraw = self._mask.reshape(-1).view(np.uint8)
            self._thisptr.reset(new row_major_dataview(&raw[0], <const cbool *> &mraw[0], self._n, ctypes))
        else:
            self._thisptr.reset(new row_major_dataview(&raw[0], NULL, self._n, ctypes))

    def size(self):
        return self._n

    def __len__(self):
        return self.size()

    def runtime_types(self):
        """[(primitive type, element count)] as the C++ view reports them"""
        cdef const vector[runtime_type] *ts = &self._thisptr.get().types()
        return [(int(ts[0][i].t()), int(ts[0][i].n())) for i in range(ts[0].size())]

    def masked_cells(self):
        """how many (row, feature element) cells the C++ view sees as masked -- a walk through row_accessor"""
        cdef dataview *v = self._thisptr.get()
        cdef row_accessor acc
        cdef size_t total = 0, e
        v.reset()
        while not v.end():
            acc = v.get()
            while not acc.end():
                for e in range(acc.curshape()):
                    if acc.ismasked(e):
                        total += 1
                acc.bump()
            v.next()
        v.reset()
        return total

    def _digest(self, h):
        if self._mask is not None:
            raise NotImplementedError("masked arrays digest not implemented")      # (as upstream)
        h.update(str(self._data.dtype).encode())
        h.update(self._data.tobytes())
