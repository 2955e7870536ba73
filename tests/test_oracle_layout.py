"""CPU: the oracle's packed-record accessors against the data the reference's own
dataview tests hold (test/test_dataview.py:31-75 via tests/golden/reference_fixtures.json)."""
import numpy as np

from oracle import oracle as orc
from tests.conftest import load_golden

NP2T = {"bool": orc.TYPE_B, "int8": orc.TYPE_I8, "uint8": orc.TYPE_U8, "int16": orc.TYPE_I16,
        "uint16": orc.TYPE_U16, "int32": orc.TYPE_I32, "uint32": orc.TYPE_U32,
        "int64": orc.TYPE_I64, "uint64": orc.TYPE_U64, "float32": orc.TYPE_F32,
        "float64": orc.TYPE_F64}


def _mk(case):
    dt = np.dtype([tuple(f) if len(f) == 2 else (f[0], f[1], (f[2],)) for f in case["dtype"]])
    rows = [tuple(tuple(v) if isinstance(v, list) else v for v in r) for r in case["rows"]]
    arr = np.array(rows, dtype=dt)
    types = [NP2T[np.dtype(f[1]).name] for f in case["dtype"]]
    counts = [f[2] if len(f) == 3 else 1 for f in case["dtype"]]
    return arr, types, counts


def test_offsets_rowsize_maskrowsize_match_reference_cases():
    fx = load_golden("reference_fixtures")
    for key in ("recarray_bool_f64", "recarray_subarray", "recarray_masked"):
        case = fx[key]
        arr, types, counts = _mk(case)
        off, row, mrow = orc.offsets_and_size(types, counts)
        assert list(off) == case["offsets"] and row == case["rowsize"] and mrow == case["maskrowsize"]
        assert arr.dtype.itemsize == row  # numpy packs records without padding, as the view assumes


def test_unpack_reproduces_every_field_of_reference_rows():
    fx = load_golden("reference_fixtures")
    for key in ("recarray_bool_f64", "recarray_subarray", "recarray_masked"):
        case = fx[key]
        arr, types, counts = _mk(case)
        off, row, _ = orc.offsets_and_size(types, counts)
        for f, (t, c) in enumerate(zip(types, counts)):
            for e in range(c):
                col = orc.unpack_column(arr, row, int(off[f]), e, t, t, len(arr))
                want = arr[arr.dtype.names[f]]
                want = want[:, e] if c > 1 else want
                np.testing.assert_array_equal(col, want)
    case = fx["recarray_bool_f64"]
    arr, types, counts = _mk(case)
    assert int(orc.unpack_column(arr, 9, 0, 0, orc.TYPE_B, orc.TYPE_I64, 3).sum()) == case["sum_f0"]


def test_cross_type_casts_follow_implicit_c_conversion():
    rec = np.array([(3, -2.75, 200)], dtype=[("a", np.int16), ("b", np.float64), ("c", np.uint8)])
    assert orc.unpack_column(rec, 11, 0, 0, orc.TYPE_I16, orc.TYPE_F32, 1)[0] == np.float32(3)
    assert orc.unpack_column(rec, 11, 2, 0, orc.TYPE_F64, orc.TYPE_I32, 1)[0] == -2  # truncation
    assert orc.unpack_column(rec, 11, 2, 0, orc.TYPE_F64, orc.TYPE_B, 1)[0] == 1   # nonzero -> true
    assert orc.unpack_column(rec, 11, 10, 0, orc.TYPE_U8, orc.TYPE_F64, 1)[0] == 200.0
    assert orc.unpack_column(rec, 11, 10, 0, orc.TYPE_U8, orc.TYPE_I8, 1)[0] == -56  # wraps


def test_group_manager_fixture_counts():
    gm = load_golden("reference_fixtures")["group_manager"]
    z = np.array(gm["assignments"])
    groups = [g for g in range(gm["created"]) if g not in gm["deleted"]]
    assert groups == gm["groups"]
    counts = {str(g): int((z == g).sum()) for g in groups}
    assert counts == gm["counts"]
    assert [g for g in groups if counts[str(g)] == 0] == gm["empty"]
    # pseudocount of the one empty group is alpha / |empty| (group_manager.hpp:274-283)
    assert orc.pseudocount(0, gm["alpha"], len(gm["empty"])) == 2.0
