"""GPU parity: bulk add/remove_value (msc_accumulate), score_data, the columnar dataview."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.gpu_helpers import (TOL, audit, crp_prior_matrix, load_state, make_feature, recarray_of, rel_err,
                               state_from_assignment)

pytestmark = pytest.mark.gpu

SPECS = [(orc.BB, 0), (orc.GP, 0), (orc.DD, 5), (orc.DD, 128), (orc.NICH, 0), (orc.NIW, 4), (orc.NIW, 32), (orc.NIW, 45)]


def _check_ss(st, fs, int_exact=True):
    for i, (F, ss64, _) in enumerate(fs):
        rec = st.get_ss(i)
        for name in rec.dtype.names:
            got = rec[name].astype(np.float64)
            want = np.asarray(ss64[name], dtype=np.float64)
            if np.issubdtype(rec.dtype[name].base, np.integer):
                assert np.array_equal(got, want), (F.family, name)   # bit-exact integer counts
            else:
                assert rel_err(got, want).max() <= TOL, (F.family, name, rel_err(got, want).max())


@pytest.mark.parametrize("K", [3, 256, 1000])
def test_accumulate_matches_sequential_add_value(gpu_ctx, K):
    import common_amd
    rng = np.random.default_rng(K)
    N = 20000
    feats = [make_feature(f, N, K, rng, d) for f, d in SPECS]
    z = rng.integers(0, K, N).astype(np.int32)
    z[::17] = -1   # unassigned rows are skipped (group_manager.hpp:218-233 never sees them)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, SPECS, K)
    for i, f in enumerate(feats):
        st.set_hp(i, f["hp"])
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.accumulate(view, zt)
    fs = []
    for f in feats:
        F = orc.Family(f["family"], f["hp"], f["dim"], "f64")
        fs.append((F, F.accumulate(K, f["values"], z), None))   # sequential Welford etc. in double
    _check_ss(st, fs)
    want_cnt = np.bincount(z[z >= 0], minlength=K)
    assert np.array_equal(st.get_group_counts(), want_cnt)


@pytest.mark.parametrize("row0,n", [(0, 1), (0, 3), (4, 4), (8, 5), (1, 4099), (2, 8190), (3, 6), (4096, 9001), (12, 16384 + 7)])
def test_accumulate_row_ranges_of_any_alignment(gpu_ctx, row0, n):
    """the row loop of k_accumulate takes four consecutive rows a load where the range starts on a multiple of four rows
    (and z is 16-byte aligned), row by row otherwise and for a partial last quad: ranges of every alignment and length,
    z passed at its own offset, integer tables bit-exact against numpy"""
    import common_amd
    rng = np.random.default_rng(row0 * 131 + n)
    N, K = 26_000, 37
    specs = [(orc.BB, 0), (orc.GP, 0), (orc.DD, 9), (orc.NICH, 0)]
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z = rng.integers(-1, K, N).astype(np.int32)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, specs, K)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    for zarg in (zt[row0:row0 + n].contiguous(), zt[row0:row0 + n]):      # its own allocation; a view at a 4-byte offset
        st.accumulate(view, zarg, row0=row0, nrows=n)
        zz = z[row0:row0 + n]
        ok = zz >= 0
        assert np.array_equal(st.get_group_counts(), np.bincount(zz[ok], minlength=K))
        sl = slice(row0, row0 + n)
        bb = st.get_ss(0)
        v = feats[0]["values"][sl][ok]
        assert np.array_equal(bb["heads"], np.bincount(zz[ok][v], minlength=K))
        assert np.array_equal(bb["tails"], np.bincount(zz[ok][~v], minlength=K))
        gp = st.get_ss(1)
        assert np.array_equal(gp["sum"], np.bincount(zz[ok], weights=feats[1]["values"][sl][ok], minlength=K).astype(np.uint64))
        dd = st.get_ss(2)
        want = np.zeros((K, 9), dtype=np.int64)
        np.add.at(want, (zz[ok], feats[2]["values"][sl][ok]), 1)
        assert np.array_equal(np.asarray(dd["counts"]).reshape(K, -1)[:, :9], want)
        nich = st.get_ss(3)
        assert np.array_equal(nich["count"], np.bincount(zz[ok], minlength=K))


def test_incremental_add_then_remove_rows(gpu_ctx):
    import common_amd
    rng = np.random.default_rng(4)
    N, K = 6000, 50
    feats = [make_feature(f, N, K, rng, d) for f, d in SPECS]
    z = rng.integers(0, K, N).astype(np.int32)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, SPECS, K)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    # first half, then second half added on top, then second half removed again
    st.accumulate(view, zt, row0=0, nrows=N // 2, reset=True)
    st.accumulate(view, zt[N // 2:].contiguous(), row0=N // 2, nrows=N - N // 2, reset=False)
    full = [(orc.Family(f["family"], f["hp"], f["dim"], "f64"), None, None) for f in feats]
    fs = [(F, F.accumulate(K, f["values"], z), None) for (F, _, _), f in zip(full, feats)]
    _check_ss(st, fs)
    st.accumulate(view, zt[N // 2:].contiguous(), row0=N // 2, nrows=N - N // 2, reset=False, subtract=True)
    half = [(F, F.accumulate(K, f["values"][:N // 2], z[:N // 2]), None) for (F, _, _), f in zip(full, feats)]
    _check_ss(st, half)
    assert np.array_equal(st.get_group_counts(), np.bincount(z[:N // 2], minlength=K))


def test_set_ss_then_add_rows_continues_from_host_state(gpu_ctx):
    import common_amd
    rng = np.random.default_rng(12)
    N, K = 3000, 9
    feats = [make_feature(orc.NICH, N, K, rng), make_feature(orc.GP, N, K, rng)]
    z = rng.integers(0, K, N).astype(np.int32)
    fs_half = state_from_assignment([dict(f, values=f["values"][:N // 2]) for f in feats], K, z[:N // 2])
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(orc.NICH, 0), (orc.GP, 0)], K)
    load_state(st, fs_half)
    st.set_group_counts(np.bincount(z[:N // 2], minlength=K).astype(np.uint32))
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.accumulate(view, zt[N // 2:].contiguous(), row0=N // 2, nrows=N - N // 2, reset=False)
    for i, f in enumerate(feats):
        F = orc.Family(f["family"], f["hp"], f["dim"], "f64")
        ss = fs_half[i][1].copy()               # the float state the device started from, widened
        for n in range(N // 2, N):
            F.add_value(ss, int(z[n]), f["values"][n])
        rec = st.get_ss(i)
        for name in rec.dtype.names:
            got, want = rec[name].astype(np.float64), np.asarray(ss[name], np.float64)
            if np.issubdtype(rec.dtype[name].base, np.integer):
                assert np.array_equal(got, want)
            else:
                audit("state.add_on_top_of_a_float_state." + name, rel_err(got, want).max(), TOL)
    assert np.array_equal(st.get_group_counts(), np.bincount(z, minlength=K))


@pytest.mark.parametrize("K", [4, 300])
def test_score_data_matches_twin(gpu_ctx, K):
    import common_amd
    rng = np.random.default_rng(31 + K)
    N = 5000
    feats = [make_feature(f, N, K, rng, d) for f, d in SPECS + [(orc.BBNC, 0)]]
    z = rng.integers(0, max(1, K - 1), N).astype(np.int32)   # last group stays empty
    fs = state_from_assignment(feats, K, z)
    st = common_amd.State(gpu_ctx, SPECS + [(orc.BBNC, 0)], K)
    load_state(st, fs)
    got = st.score_data().cpu().numpy()
    for i, (F, ss64, _) in enumerate(fs):
        want = F.score_data_all(ss64)
        assert rel_err(got[i], want).max() <= TOL, (F.family, rel_err(got[i], want).max())
    assert np.all(got[:len(SPECS), K - 1] == 0.0)   # empty group: marginal likelihood of no data


def test_dataview_unpack_is_bit_exact_with_runtime_cast(gpu_ctx):
    import common_amd
    rng = np.random.default_rng(8)
    N = 1237
    dt = np.dtype([("b", np.bool_), ("i8", np.int8), ("u16", np.uint16), ("i32", np.int32),
                   ("u64", np.uint64), ("f32", np.float32), ("f64", np.float64), ("v", np.float32, (3,)),
                   ("w", np.int16, (2,))])
    arr = np.zeros(N, dtype=dt)
    arr["b"] = rng.random(N) < 0.5
    arr["i8"] = rng.integers(-128, 128, N)
    arr["u16"] = rng.integers(0, 65536, N)
    arr["i32"] = rng.integers(-2**31, 2**31, N)
    arr["u64"] = rng.integers(0, 2**63, N).astype(np.uint64)
    arr["f32"] = rng.normal(0, 100, N)
    arr["f64"] = rng.normal(0, 1e6, N)
    arr["v"] = rng.normal(0, 1, (N, 3))
    arr["w"] = rng.integers(-3000, 3000, (N, 2))
    assert arr.dtype.itemsize == 1 + 1 + 2 + 4 + 8 + 4 + 8 + 12 + 4   # packed, no padding
    view = common_amd.DataView.from_recarray(gpu_ctx, arr)
    assert view.size() == N and len(view) == N
    for f, name in enumerate(dt.names):
        got = view.column_to_numpy(f)
        np.testing.assert_array_equal(got, arr[name])
    # conversion at upload: every feature to float32 / int32 with the implicit C++ conversion
    types = common_amd.runtime_types_of(dt)
    off, row, _ = orc.offsets_and_size([t for t, _ in types], [c for _, c in types])
    for target, ttype in ((np.float32, orc.TYPE_F32), (np.int32, orc.TYPE_I32)):
        v2 = common_amd.DataView.from_recarray(gpu_ctx, arr, col_types=[ttype] * len(types))
        for f, (t, c) in enumerate(types):
            if ttype == orc.TYPE_I32 and t in (orc.TYPE_F32, orc.TYPE_F64, orc.TYPE_U64):
                continue   # out-of-range float->int conversions are undefined in C++ too
            got = v2.column_to_numpy(f).reshape(N, c)
            for e in range(c):
                want = orc.unpack_column(arr, row, int(off[f]), e, t, ttype, N)
                np.testing.assert_array_equal(got[:, e], want)


def test_masked_recarray_keeps_mask_columns(gpu_ctx):
    import common_amd
    from tests.conftest import load_golden
    case = load_golden("reference_fixtures")["recarray_masked"]
    x = np.ma.masked_array(np.array([tuple(case["rows"][0])], dtype=[("f%d" % i, np.bool_) for i in range(5)]),
                           mask=[tuple(case["mask"][0])])
    view = common_amd.DataView.from_recarray(gpu_ctx, x)
    for f in range(5):
        assert bool(view.column_to_numpy(f)[0]) == case["rows"][0][f]


def test_column_of_another_type_is_converted_at_bind(gpu_ctx):
    """runtime_cast::cast (runtime_type.hpp:145-166) converts whatever primitive type a column holds, per value; here the
    column is converted once, when a state binds it.  The reference's own dtypes -- niw's float64 vectors
    (microscopes/models.pyx:259), numpy-default float64 for nich, int64 counts and categories -- must score and accumulate
    exactly like the same values cast beforehand, bit for bit; a wrong element COUNT is still refused."""
    import common_amd
    from common_amd import models
    rng = np.random.default_rng(5)
    N, K, d = 3000, 7, 4
    specs = [(orc.NICH, 0), (orc.GP, 0), (orc.DD, 6), (orc.NIW, d), (orc.BB, 0), (orc.DM, 3)]
    feats = [make_feature(f, N, K, rng, dd_) for f, dd_ in specs]
    z = rng.integers(0, K, N).astype(np.int32)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    native = recarray_of(feats)
    wide_dt = np.dtype([("f0", np.float64), ("f1", np.int64), ("f2", np.int64),
                        ("f3", models.niw(d).py_desc().get_np_dtype()), ("f4", np.uint8), ("f5", np.int64, (3,))])
    assert wide_dt["f3"].base == np.float64 and wide_dt["f3"].shape == (d,)
    wide = np.zeros(N, dtype=wide_dt)
    for name in native.dtype.names:
        wide[name] = native[name]                      # (float32 -> float64 and back is exact; counts fit)

    def run(arr):
        view = common_amd.DataView.from_recarray(gpu_ctx, arr)
        st = common_amd.State(gpu_ctx, specs, K)
        for i, f in enumerate(feats):
            st.set_hp(i, orc.Family(f["family"], f["hp"], f["dim"], "f64").hp)
        st.accumulate(view, zt)
        return view, st, st.score_value(view).cpu().numpy(), [st.get_ss(i) for i in range(len(specs))]
    v0, s0, sc0, ss0 = run(native)
    v1, s1, sc1, ss1 = run(wide)
    assert v1.column_type(0)[0] == orc.TYPE_F64 and v1.column_type(1)[0] == orc.TYPE_I64      # the view keeps the caller's types
    assert np.array_equal(sc0, sc1)
    for a, b in zip(ss0, ss1):
        assert a.tobytes() == b.tobytes()
    # device tensors take the same road (float64 columns from torch)
    cols = [torch.from_numpy(np.ascontiguousarray(wide[n])).to(gpu_ctx.torch_device) for n in wide.dtype.names]
    vt = common_amd.DataView.from_tensors(gpu_ctx, cols)
    assert np.array_equal(s1.score_value(vt).cpu().numpy(), sc0)
    # the element count is the model's: a scalar column cannot feed niw(4)
    st = common_amd.State(gpu_ctx, [(orc.NIW, d)], K)
    with pytest.raises(common_amd.MicroscopesHipError):
        st.score_value(v0, cols=[0])


def test_accumulate_with_more_groups_than_lds_histograms_hold(gpu_ctx):
    """K = 12000: the per-workgroup LDS histograms do not fit; rows are added straight into the additive tables
    (k_accumulate_global).  Same numbers, then a subtraction, then scores and a sweep on that many groups."""
    import common_amd
    K, N = 12000, 40000
    specs = [(orc.BB, 0), (orc.GP, 0), (orc.DD, 5), (orc.NICH, 0), (orc.BNB, 0), (orc.DM, 3)]
    rng = np.random.default_rng(12)
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z = rng.integers(0, K, N).astype(np.int32)
    z[::13] = -1
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, specs, K)
    for i, f in enumerate(feats):
        st.set_hp(i, orc.Family(f["family"], f["hp"], f["dim"], "f64").hp)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.accumulate(view, zt)
    fs = []
    for f in feats:
        F = orc.Family(f["family"], f["hp"], f["dim"], "f64")
        fs.append((F, F.accumulate(K, f["values"], z), None))
    _check_ss(st, fs)
    assert np.array_equal(st.get_group_counts(), np.bincount(z[z >= 0], minlength=K))
    # take the second half out again
    half = N // 2
    st.accumulate(view, zt[half:].contiguous(), row0=half, nrows=N - half, reset=False, subtract=True)
    fs_half = []
    for f in feats:
        F = orc.Family(f["family"], f["hp"], f["dim"], "f64")
        fs_half.append((F, F.accumulate(K, f["values"][:half], z[:half]), None))
    for i, (F, ss64, _) in enumerate(fs_half):
        rec = st.get_ss(i)
        for name in rec.dtype.names:
            if np.issubdtype(rec.dtype[name].base, np.integer):
                assert np.array_equal(rec[name], np.asarray(ss64[name])), (F.family, name)
    # scores of a few rows against all 12000 groups
    rows = np.arange(0, 64)
    got = st.score_value(view, row0=0, nrows=64).cpu().numpy()
    # against the twin on the float suff-stats the device holds after add + subtract (get_ss, widened): what the float
    # fields lost in the two passes belongs to the state, not to the scoring kernel (round 2 gated 5x against the exact
    # half-data statistics instead)
    held = [orc.widen_ss(F.family, st.get_ss(i).astype(orc.ss_dtype(F.family, F.dim, "f32")), F.dim) for i, (F, _, _) in enumerate(fs_half)]
    want = sum(F.score_matrix(h, f["values"][rows]) for f, (F, _, _), h in zip(feats, fs_half, held))
    mag = sum(np.maximum(1.0, np.abs(F.score_matrix(h, f["values"][rows]))) for f, (F, _, _), h in zip(feats, fs_half, held))
    audit("state.scores_after_add_subtract", (np.abs(got - want) / mag).max(), TOL)
    st.set_alpha(1.0)
    z2 = zt.clone()
    st.sweep_step(view, z2, seed=3, sweep=0)
    zn = z2.cpu().numpy()
    assert ((zn >= 0) & (zn < K)).all() and (zn != np.where(z < 0, 0, z)).any()
    assert np.array_equal(st.get_group_counts(), np.bincount(zn, minlength=K))


def test_entity_op_keeps_every_table_current(gpu_ctx):
    """msc_entity_op: one entity joins / leaves one group, the group by value, in one launch -- after a run of such moves
    the suff-stats, the group sizes and the scores are those of a state built from the final assignment in one pass"""
    import common_amd
    rng = np.random.default_rng(12)
    N, K = 400, 9
    specs = [(orc.BB, 0), (orc.GP, 0), (orc.DD, 5), (orc.NICH, 0), (orc.BNB, 0)]
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    rec = recarray_of(feats)
    mask = np.zeros(N, dtype=[(n, np.bool_) for n in rec.dtype.names])
    mask["f3"][::7] = True                                        # masked values take no part
    view = common_amd.DataView.from_recarray(gpu_ctx, np.ma.masked_array(rec, mask=mask))
    st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    for i, f in enumerate(feats):
        st.set_hp(i, f["hp"])
    st.set_alpha(0.7)
    z = np.full(N, -1, dtype=np.int32)
    zt = torch.from_numpy(z.copy()).to(gpu_ctx.torch_device)
    for n in range(N):                                            # everybody joins a group, one by one
        z[n] = int(rng.integers(0, K - 1))                        # (group K - 1 stays empty)
        st.entity_op(view, n, z[n], join=True, z=zt)
    for _ in range(300):                                          # Gibbs-style moves: leave, (score), join
        n = int(rng.integers(0, N))
        st.entity_op(view, n, z[n], join=False, z=zt)
        row = st.score_value(view, row0=n, nrows=1, crp_prior=True)           # tables are current: no prepare pass runs
        assert bool(torch.isfinite(row).all())
        z[n] = int(rng.integers(0, K - 1))
        st.entity_op(view, n, z[n], join=True, z=zt)
    assert np.array_equal(zt.cpu().numpy(), z)
    ref = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    for i, f in enumerate(feats):
        ref.set_hp(i, f["hp"])
    ref.set_alpha(0.7)
    ref.accumulate(view, torch.from_numpy(z).to(gpu_ctx.torch_device))
    assert np.array_equal(st.get_group_counts(), ref.get_group_counts())
    for i in range(len(feats)):
        a, b = st.get_ss(i), ref.get_ss(i)
        for name in a.dtype.names:
            if np.issubdtype(a.dtype[name].base, np.integer):
                assert np.array_equal(a[name], b[name]), (i, name)
            else:
                assert rel_err(a[name], b[name]).max() <= TOL, (i, name)
    got = st.score_value(view, crp_prior=True).cpu().numpy()
    want = ref.score_value(view, crp_prior=True).cpu().numpy()
    # two device states whose float fields came about differently (one move at a time / one pass): each is held against
    # the twin on ITS OWN float fields at the plain gate (round 2 compared them with each other at 2x)
    for name, s_, sc in (("incremental", st, got), ("one_pass", ref, want)):
        held = [orc.widen_ss(f["family"], s_.get_ss(i).astype(orc.ss_dtype(f["family"], f["dim"], "f32")), f["dim"]) for i, f in enumerate(feats)]
        tw = [orc.Family(f["family"], f["hp"], f["dim"], "f64").score_matrix(h, f["values"]) for f, h in zip(feats, held)]
        for t_, f in zip(tw, feats):
            if f is feats[3]:
                t_[mask["f3"]] = 0.0
        total = sum(tw) + crp_prior_matrix(s_.get_group_counts(), 0.7)
        mag = sum(np.maximum(1.0, np.abs(t_)) for t_ in tw)
        audit("state.entity_op_scores." + name, (np.abs(sc - total) / np.maximum(mag, np.abs(total))).max(), TOL)
    with pytest.raises(common_amd.MicroscopesHipError):
        st.entity_op(view, 0, K, join=True)                       # no such group


def test_entity_op_preconditions_are_checked_on_the_device(gpu_ctx):
    """msc_entity_op's preconditions (the reference asserts them in group_manager::add_value / remove_value,
    group_manager.hpp:218-248): a leave needs a non-empty group that the row is in, a join an unassigned row.  The kernel
    that notices skips the update it concerns and reports through the device's error word: the next synchronising or
    launching call returns MSC_EDEVICE (-6), once; and the call is refused outright between msc_sweep_step_begin and
    msc_state_commit_reduce."""
    import common_amd
    rng = np.random.default_rng(3)
    N, K = 64, 5
    specs = [(orc.BB, 0), (orc.NICH, 0)]
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, specs, K)
    zt = torch.full((N,), -1, dtype=torch.int32, device=gpu_ctx.torch_device)
    st.entity_op(view, 0, 2, join=True, z=zt)
    st.entity_op(view, 1, 2, join=True, z=zt)
    gpu_ctx.synchronize()                                         # nothing to report
    assert st.get_group_counts()[2] == 2 and zt[:2].tolist() == [2, 2]

    def expect_device_error(what):
        with pytest.raises(common_amd.MicroscopesHipError) as ei:
            gpu_ctx.synchronize()
        assert ei.value.code == -6 and "msc_entity_op" in str(ei.value), what
        gpu_ctx.synchronize()                                     # reported once

    st.entity_op(view, 5, 3, join=False, z=zt)                    # row 5 is not in group 3 (which is empty)
    expect_device_error("leave from an empty group")
    st.entity_op(view, 0, 4, join=False, z=zt)                    # row 0 is in group 2, not 4
    expect_device_error("leave from another group")
    st.entity_op(view, 1, 3, join=True, z=zt)                     # row 1 is already assigned
    expect_device_error("join of an assigned row")
    # the group sizes and the assignment vector were left alone by the refused moves
    assert st.get_group_counts().tolist() == [0, 0, 2, 0, 0] and zt[:2].tolist() == [2, 2]
    # the tables of a state that saw a refused move are to be rebuilt; after that everything works on
    st.accumulate(view, zt)
    st.entity_op(view, 0, 2, join=False, z=zt)
    gpu_ctx.synchronize()
    assert st.get_group_counts()[2] == 1 and int(zt[0]) == -1
    # between the sharded step's two halves the additive tables hold uncommitted sums: refused on the host
    st.set_alpha(1.0)
    z2 = torch.zeros(N, dtype=torch.int32, device=gpu_ctx.torch_device)
    st.accumulate(view, z2)
    st.sweep_step_begin(view, z2, seed=1, sweep=0)
    with pytest.raises(common_amd.MicroscopesHipError):
        st.entity_op(view, 0, 0, join=False, z=z2)
    st.commit_reduce()
    st.entity_op(view, 0, int(z2[0]), join=False, z=z2)
    gpu_ctx.synchronize()


def test_default_allocator_places_large_buffers(gpu_ctx):
    """msc_device_alloc from 64 MiB on: a probed, chunk-mapped buffer (msc_device_alloc_stats says what was tried), zero
    filled, usable as a score matrix, freed with the tensor; below that a plain allocation"""
    import common_amd
    t = gpu_ctx.alloc((300_000, 64), torch.float32)               # 76.8 MB
    rates, kept = gpu_ctx.alloc_stats()
    assert 1 <= len(rates) <= 24 and 0 <= kept < len(rates) and all(r > 100.0 for r in rates)
    assert rates[kept] == max(rates) and float(t.abs().sum()) == 0.0
    small = gpu_ctx.alloc((1000,), torch.float32)
    assert float(small.sum()) == 0.0
    st = common_amd.State(gpu_ctx, [(orc.NICH, 0)], 64)
    x = torch.randn(300_000, device=gpu_ctx.torch_device)
    view = common_amd.DataView.from_tensors(gpu_ctx, [x])
    st.score_value(view, out=t)
    assert bool(torch.isfinite(t).all()) and float(t.abs().sum()) > 0.0
    with pytest.raises(RuntimeError):
        gpu_ctx.close()                                           # live buffers: refused
    del t, small


def test_a_state_outlives_the_view_it_last_bound(gpu_ctx):
    """The library keeps no reference to a dataview (neither does the reference: recarray/_dataview.pxd:24-27 borrows).
    A state that scored against a view remembers it by pointer and serial, and what re-plans outside a call that was
    handed a view -- msc_state_set_hp on a dd feature -- must not look at a view that has since been destroyed (round 3:
    plan_groups walked bound_view->packed_bits there: a use after free).  Afterwards a new view binds and scores as if
    nothing had happened."""
    import common_amd
    rng = np.random.default_rng(11)
    N, K = 5000, 40
    specs = [(orc.BB, 0), (orc.BB, 0), (orc.BB, 0), (orc.DD, 6), (orc.NICH, 0)]
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z = rng.integers(0, K, N).astype(np.int32)
    fs = state_from_assignment(feats, K, z)
    st = common_amd.State(gpu_ctx, specs, K)
    load_state(st, fs)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    first = st.score_value(view).cpu().numpy()
    st._bound_view = None                                         # (the Python State would keep it alive: take that away)
    view.close()                                                  # msc_dataview_destroy
    del view
    junk = [torch.full((N,), 7, dtype=torch.uint8, device=gpu_ctx.torch_device) for _ in range(8)]   # reuse the freed pages
    hp = dict(alphas=[0.5, 1.0, 1.5, 2.0, 2.5, 3.0])
    st.set_hp(3, hp)                                              # re-plans: no view to look at
    feats[3]["hp"] = hp
    fs = state_from_assignment(feats, K, z)
    view2 = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    got = st.score_value(view2).cpu().numpy()
    want = sum(F.score_matrix(ss64, f["values"]) for f, (F, ss64, _) in zip(feats, fs))
    mag = sum(np.maximum(1.0, np.abs(F.score_matrix(ss64, f["values"]))) for f, (F, ss64, _) in zip(feats, fs))
    audit("state.rebind_after_view_destroyed", (np.abs(got - want) / mag).max(), TOL)
    assert np.abs(got - first).max() > 1e-3                       # (the new alphas took effect)
    del junk


@pytest.mark.parametrize("N", [6000, 40_000])
def test_columns_rewritten_in_place_need_invalidate(gpu_ctx, N):
    """A view over the caller's tensors (msc_dataview_from_device_columns) caches what it derives from them -- bool columns
    packed four to a byte, masked columns with the mask folded in, converted copies, column maxima, the plain nich columns as
    one row-major matrix (what the role-split kernels' nich waves read: 40k rows take those kernels).  After the caller
    refills the tensors in place (minibatches through fixed buffers) msc_dataview_invalidate drops all of it: scores and
    suff-stats follow the new contents."""
    import common_amd
    rng = np.random.default_rng(12)
    K = 70
    specs = [(orc.BB, 0), (orc.BB, 0), (orc.BB, 0), (orc.BB, 0), (orc.GP, 0), (orc.NICH, 0)] + ([(orc.NICH, 0)] * 2 if N > 20_000 else [])
    dev = gpu_ctx.torch_device

    def batch(seed):
        r = np.random.default_rng(seed)
        feats = [make_feature(f, N, K, r, d) for f, d in specs]
        feats[4]["values"] = (feats[4]["values"] % (9 + 40 * seed)).astype(np.uint32)     # (another maximum per batch)
        return feats
    feats = batch(0)
    cols = [torch.from_numpy(f["values"].copy()).to(dev) for f in feats]
    mask3 = torch.from_numpy(rng.random(N) < 0.2).to(dev)
    view = common_amd.DataView.from_tensors(gpu_ctx, cols, masks=[None, None, None, mask3] + [None] * (len(specs) - 4))
    z = rng.integers(0, K, N).astype(np.int32)
    zt = torch.from_numpy(z).to(dev)
    st = common_amd.State(gpu_ctx, specs, K)

    def check(feats, tag):
        m3 = mask3.cpu().numpy()
        st.accumulate(view, zt)
        got = st.score_value(view).cpu().numpy()
        total, mag = 0.0, 0.0
        for i, f in enumerate(feats):
            F = orc.Family(f["family"], f["hp"], f["dim"], "f64")
            zz = z.copy()
            if i == 3:
                zz[m3] = -1                                       # a masked value takes no part in the suff-stats
            ss = orc.widen_ss(f["family"], orc.narrow_ss(f["family"], F.accumulate(K, f["values"], zz), f["dim"]), f["dim"])
            m = F.score_matrix(ss, f["values"])
            if i == 3:
                m[m3] = 0.0                                       # ... nor in the scores
            total, mag = total + m, mag + np.maximum(1.0, np.abs(m))
        audit("state.invalidate." + tag, (np.abs(got - total) / mag).max(), TOL)
        return got
    a = check(feats, "first_batch")
    feats2 = batch(1)
    for c, f in zip(cols, feats2):
        c.copy_(torch.from_numpy(f["values"].copy()).to(dev))     # in place: the view's pointers stay what they were
    view.invalidate()
    b = check(feats2, "second_batch")
    assert np.abs(a - b).max() > 1e-2
