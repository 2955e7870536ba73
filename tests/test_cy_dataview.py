"""The dataview half of the Cython boundary (common_amd/cy/_dataview: this build's counterpart of
microscopes/common/recarray/_dataview.pxd / .pyx:61-92 and microscopes/common/_dataview.pyx:6-53): numpy structured
arrays wrapped in the C++ row_major_dataview, with the reference's argument rules (test/test_dataview.py holds the same
shapes: plain, masked, vector columns).  No device work here."""
import os
import sys

import numpy as np
import numpy.ma as ma
import pytest

from common_amd import runtime

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cy():
    from common_amd.cy.build import build_module
    build_module(os.path.join(ROOT, "common_amd", "cy", "_models.pyx"))
    build_module(os.path.join(ROOT, "common_amd", "cy", "_dataview.pyx"))
    build_module(os.path.join(ROOT, "tests", "cy", "downstream_probe.pyx"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "cy"))
    import downstream_probe
    from common_amd.cy import _dataview
    return _dataview, downstream_probe


def _rows(n=11):
    dt = np.dtype([("a", np.bool_), ("b", np.float32, (3,)), ("c", np.int32), ("d", np.uint32), ("e", np.float64)])
    rng = np.random.default_rng(3)
    y = np.zeros(n, dtype=dt)
    y["a"] = rng.random(n) < 0.5
    y["b"] = rng.normal(0, 1, (n, 3))
    y["c"] = rng.integers(-5, 5, n)
    y["d"] = rng.integers(0, 9, n)
    y["e"] = rng.normal(0, 1, n)
    return y


def test_argument_rules_are_the_references(cy):
    dv, _ = cy
    with pytest.raises(ValueError, match="npd is None"):
        dv.numpy_dataview(None)
    with pytest.raises(ValueError, match="1D"):
        dv.numpy_dataview(np.zeros((3, 2), dtype=[("a", np.bool_)]))
    with pytest.raises(ValueError, match="structural arrays only"):
        dv.numpy_dataview(np.zeros(4, dtype=np.float32))
    with pytest.raises(NotImplementedError):
        dv.abstract_dataview().size()


def test_runtime_types_and_size_as_the_cxx_view_reports_them(cy):
    dv, probe = cy
    y = _rows()
    v = dv.numpy_dataview(y)
    assert isinstance(v, dv.abstract_dataview)
    assert v.size() == len(v) == len(y)
    want = runtime.runtime_types_of(y.dtype)
    assert v.runtime_types() == [(int(t), int(n)) for t, n in want]
    assert probe.view_size(v) == (len(y), len(y.dtype))            # a cdef consumer reading `_thisptr`
    assert v.masked_cells() == 0
    e = dv.numpy_dataview(y[:0])
    assert e.size() == 0 and e.masked_cells() == 0 and probe.view_size(e) == (0, len(y.dtype))


def test_masks_reach_the_cxx_row_accessor(cy):
    dv, _ = cy
    y = _rows(9)
    m = np.zeros(9, dtype=[(k, np.bool_, y.dtype[k].shape) for k in y.dtype.names])
    m["a"][[1, 4]] = True
    m["b"][2] = [True, False, True]
    m["e"][8] = True
    v = dv.numpy_dataview(ma.array(y, mask=m))
    assert v.masked_cells() == 2 + 2 + 1
    assert dv.numpy_dataview(ma.array(y, mask=np.ones_like(m))).masked_cells() == 9 * (1 + 3 + 1 + 1 + 1)
    assert dv.numpy_dataview(ma.array(y)).masked_cells() == 0      # a masked array with nomask


def test_the_view_keeps_the_array_alive_and_sees_non_contiguous_input(cy):
    dv, _ = cy
    y = _rows(20)
    v = dv.numpy_dataview(y[::2])                                   # strided: the view packs its own copy
    assert v.size() == 10 and v.masked_cells() == 0
    d0 = v.digest().hexdigest()
    del y
    assert v.digest().hexdigest() == d0                             # owns what it points into
    assert dv.numpy_dataview(_rows(20)[::2]).digest().hexdigest() == d0
    assert dv.numpy_dataview(_rows(20)[1::2]).digest().hexdigest() != d0
    with pytest.raises(NotImplementedError):                        # (as upstream: masked digests are not defined)
        dv.numpy_dataview(ma.array(_rows(4))).digest()
