# cython: language_level=3
# distutils: language = c++
"""What a downstream Cython state object (mixturemodel / irm) does with a model descriptor, written against this build's
.pxd files: cimport the extension type, pull the shared_ptr[model] out of `desc.c_desc()`, create hypers and groups
through the C++ virtual API and drive add_value / score_value / score_data per value.  Built by the tests with
common_amd/cy/build.py, exactly as a downstream package would build its own modules."""
from libcpp.memory cimport shared_ptr
from libcpp.vector cimport vector

from common_amd.cy._models cimport _base
from common_amd.cy._models_h cimport model, hypers, group, rng_t, value_accessor
from common_amd.cy._runtime_type_h cimport runtime_type

import numpy as np


cdef dict describe(shared_ptr[model] m):
    """the cdef-level walk: model -> runtime type, model -> hypers -> bag, hypers -> group -> bag"""
    cdef runtime_type t = m.get().get_runtime_type()
    cdef shared_ptr[hypers] h = m.get().create_hypers()
    cdef rng_t rng = rng_t(7)
    cdef shared_ptr[group] g = h.get().create_group(rng)
    return {"type": int(t.t()), "n": int(t.n()), "vec": bool(t.vec()), "size": int(t.size()),
            "hp": <bytes>h.get().get_hp(), "ss": <bytes>g.get().get_ss()}


def probe(_base desc):
    """takes the object model_descriptor.c_desc() returns; no device work"""
    cdef shared_ptr[model] m = desc.get()
    cdef shared_ptr[hypers] h = desc.create_hypers()
    out = describe(m)
    out["hp_via_base"] = <bytes>h.get().get_hp()
    out["use_count"] = int(m.use_count())
    return out


def score_through_virtual_api(_base desc, bytes hp, values, probe_value):
    """hypers.set_hp(bag); group = hypers.create_group(rng); add every value; then score_value(probe) and score_data --
    each call one group::* virtual call on the device (msc_value_op_single).  values: 1-D (scalar models) or 2-D
    (vector models) numpy array of the model's value dtype.  -> (score_value, score_data, suff-stat bag)"""
    cdef shared_ptr[model] m = desc.get()
    cdef runtime_type t = m.get().get_runtime_type()
    cdef shared_ptr[hypers] h = desc.create_hypers()
    h.get().set_hp(hp)
    cdef rng_t rng = rng_t(11)
    cdef shared_ptr[group] g = h.get().create_group(rng)
    arr = np.ascontiguousarray(values)
    cdef const unsigned char[::1] raw = arr.reshape(-1).view(np.uint8)
    cdef size_t stride = t.size(), i, n = arr.shape[0]
    assert raw.shape[0] == n * stride, "values do not have the model's runtime type"
    for i in range(n):
        g.get().add_value(h.get()[0], value_accessor(&raw[i * stride], NULL, t), rng)
    pv = np.ascontiguousarray(probe_value)
    cdef const unsigned char[::1] praw = pv.reshape(-1).view(np.uint8)
    assert praw.shape[0] == stride
    cdef float sv = g.get().score_value(h.get()[0], value_accessor(&praw[0], NULL, t), rng)
    cdef float sd = g.get().score_data(h.get()[0], rng)
    return float(sv), float(sd), <bytes>g.get().get_ss()
