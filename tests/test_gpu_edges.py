"""GPU: degenerate shapes and error behaviour of the C ABI (through common_amd.runtime):
empty and one-row views, one group, row counts that straddle every tile boundary, bad arguments."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.gpu_helpers import TOL, load_state, make_feature, oracle_scores, recarray_of, rel_err, state_from_assignment

pytestmark = pytest.mark.gpu


def _setup(gpu_ctx, specs, N, K, seed):
    import common_amd
    rng = np.random.default_rng(seed)
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z = rng.integers(0, K, N).astype(np.int32)
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    load_state(st, fs)
    st.set_group_counts(np.bincount(z, minlength=K).astype(np.uint32))
    return feats, z, fs, view, st


@pytest.mark.parametrize("specs", [[(orc.NICH, 0)], [(orc.BB, 0), (orc.GP, 0), (orc.DD, 4)], [(orc.NIW, 3)], [(orc.DM, 3)]])
def test_empty_view_is_a_no_op(gpu_ctx, specs):
    import common_amd
    K = 5
    rng = np.random.default_rng(0)
    feats = [make_feature(f, 0, K, rng, d) for f, d in specs]
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    assert view.nrows == 0
    st = common_amd.State(gpu_ctx, specs, K)
    out = st.score_value(view)
    assert tuple(out.shape) == (0, K)
    z = torch.empty(0, dtype=torch.int32, device=gpu_ctx.torch_device)
    st.accumulate(view, z)
    assert np.array_equal(st.get_group_counts(), np.zeros(K, dtype=np.uint32))
    st.sweep_assign(view, z, seed=1, sweep=0)
    sd = st.score_data().cpu().numpy()
    assert np.all(np.abs(sd) < 1e-6)          # empty groups: marginal likelihood of no data


@pytest.mark.parametrize("N", [1, 2, 7, 8, 9, 63, 64, 65, 127, 128, 129, 255, 257, 1023, 1025])
def test_row_counts_across_every_tile_boundary(gpu_ctx, N):
    for specs, K in (([(orc.NICH, 0)], 3), ([(orc.BB, 0), (orc.NICH, 0), (orc.GP, 0)], 5), ([(orc.DM, 3), (orc.BNB, 0)], 4)):
        feats, z, fs, view, st = _setup(gpu_ctx, specs, N, K, seed=N)
        got = st.score_value(view).cpu().numpy()
        assert got.shape == (N, K)
        assert rel_err(got, oracle_scores(feats, fs)).max() <= TOL
        zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
        loo = st.score_value(view, z=zt).cpu().numpy()
        assert rel_err(loo, oracle_scores(feats, fs, z=z)).max() <= TOL


@pytest.mark.parametrize("K", [1, 2, 255, 256, 257, 1024, 1025])
def test_group_counts_across_every_tile_boundary(gpu_ctx, K):
    feats, z, fs, view, st = _setup(gpu_ctx, [(orc.NICH, 0)], 300, K, seed=K)
    got = st.score_value(view).cpu().numpy()
    assert rel_err(got, oracle_scores(feats, fs)).max() <= TOL
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.set_alpha(2.0)
    st.sweep_assign(view, zt, seed=3, sweep=1)
    znew = zt.cpu().numpy()
    assert znew.min() >= 0 and znew.max() < K
    if K == 1:
        assert np.all(znew == 0)


def test_one_group_one_row_one_feature(gpu_ctx):
    feats, z, fs, view, st = _setup(gpu_ctx, [(orc.GP, 0)], 1, 1, seed=5)
    got = st.score_value(view).cpu().numpy()
    assert rel_err(got, oracle_scores(feats, fs)).max() <= TOL
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    loo = st.score_value(view, z=zt).cpu().numpy()     # the only row leaves the only group: prior predictive
    F = orc.Family(orc.GP, feats[0]["hp"], 0, "f64")
    assert rel_err(loo[0, 0], F.score_value(F.new_groups(1), 0, feats[0]["values"][0])) <= TOL


def test_unassigned_rows_are_ignored_by_accumulate_and_scored_without_removal(gpu_ctx):
    import common_amd
    rng = np.random.default_rng(2)
    N, K = 500, 6
    feats = [make_feature(orc.NICH, N, K, rng), make_feature(orc.DD, N, K, rng, 5)]
    z = rng.integers(0, K, N).astype(np.int32)
    z[::3] = -1                                                   # not assigned (entity_state.hpp:57-72 callers)
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(orc.NICH, 0), (orc.DD, 5)], K)
    for i, (F, _, _) in enumerate(fs):
        st.set_hp(i, F.hp)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.accumulate(view, zt)
    assert np.array_equal(st.get_group_counts(), np.bincount(z[z >= 0], minlength=K))
    assert np.array_equal(st.get_ss(1)["counts"], fs[1][2]["counts"])
    loo = st.score_value(view, z=zt).cpu().numpy()
    assert rel_err(loo, oracle_scores(feats, fs, z=z)).max() <= TOL


def test_bad_arguments_are_refused_with_a_message(gpu_ctx):
    import common_amd
    from common_amd._lib import MicroscopesHipError
    rng = np.random.default_rng(1)
    feats = [make_feature(orc.NICH, 10, 2, rng)]
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    with pytest.raises(MicroscopesHipError):
        common_amd.State(gpu_ctx, [(orc.DD, 0)], 3)              # dd needs 1..128 categories
    with pytest.raises(MicroscopesHipError):
        common_amd.State(gpu_ctx, [(orc.NIW, 129)], 3)           # the prepare kernel's LDS triangles hold 128
    with pytest.raises(MicroscopesHipError):
        common_amd.State(gpu_ctx, [(99, 0)], 3)                  # unknown family
    with pytest.raises(MicroscopesHipError):
        common_amd.State(gpu_ctx, [(orc.NICH, 0)], 0)            # no groups
    st = common_amd.State(gpu_ctx, [(orc.NICH, 0)], 2)
    with pytest.raises(MicroscopesHipError):
        st.score_value(view, row0=5, nrows=10)                   # rows outside the view
    with pytest.raises(MicroscopesHipError):
        st.set_hp(0, np.zeros(3, dtype=np.float32))              # wrong hp block size
    st2 = common_amd.State(gpu_ctx, [(orc.NICH, 0), (orc.BB, 0)], 2)
    with pytest.raises(MicroscopesHipError):
        st2.score_value(view)                                    # the view has one column, the state two features


def test_probed_allocation_is_zero_filled_usable_and_outlives_its_handle(gpu_ctx):
    """msc_device_alloc_probed: the best-placed of a few candidate buffers, as a tensor that owns it"""
    import gc
    import common_amd
    t, rates, kept = gpu_ctx.alloc_probed((4096, 256), torch.float32, candidates=3)
    assert t.shape == (4096, 256) and t.dtype == torch.float32 and len(rates) == 3 and 0 <= kept < 3
    assert all(r > 0 for r in rates) and rates[kept] == max(rates)
    assert float(t.abs().sum()) == 0.0
    x = torch.randn(4096, device=gpu_ctx.torch_device)
    view = common_amd.DataView.from_tensors(gpu_ctx, [x])
    st = common_amd.State(gpu_ctx, [(common_amd.NICH, 0)], 256)
    st.score_value(view, out=t)
    want = st.score_value(view)
    assert torch.equal(t, want)
    keep = t[5].clone()
    gc.collect()
    assert torch.equal(t[5], keep)                      # the owner rides on the tensor's storage
    with pytest.raises(common_amd.MicroscopesHipError):
        gpu_ctx.alloc_probed((4,), torch.float32, candidates=0)


def test_score_tune_is_explicit_and_changes_no_value(gpu_ctx):
    """msc_score_tune settles the single-nich pass's launch shape for a buffer (synchronously, on request); the scores a
    pass writes do not depend on the shape, and states that do not take that kernel report -1"""
    import common_amd
    N, K = 300_000, 256                                     # 300k x 256 scores: large enough for every shape to differ
    g = torch.Generator(device=gpu_ctx.torch_device)
    g.manual_seed(5)
    x = torch.randn(N, generator=g, device=gpu_ctx.torch_device)
    z = torch.randint(0, K, (N,), generator=g, device=gpu_ctx.torch_device, dtype=torch.int32)
    view = common_amd.DataView.from_tensors(gpu_ctx, [x])
    st = common_amd.State(gpu_ctx, [(common_amd.NICH, 0)], K)
    st.accumulate(view, z)
    out = torch.empty((N, K), dtype=torch.float32, device=gpu_ctx.torch_device)
    before = st.score_value(view).clone()
    shape, ms = st.score_tune(view, out)
    assert 0 <= shape < 8 and ms > 0
    assert torch.equal(st.score_value(view, out=out), before)            # the tuned buffer
    assert torch.equal(st.score_value(view), before)                     # another buffer of the same size (fallback entry)
    st2 = common_amd.State(gpu_ctx, [(common_amd.NICH, 0), (common_amd.BB, 0)], K)
    view2 = common_amd.DataView.from_tensors(gpu_ctx, [x, x > 0])
    assert st2.score_tune(view2, out)[0] == -1                           # the tile kernel has no shapes to settle


@pytest.mark.parametrize("specs,K", [([(orc.NICH, 0)], 300), ([(orc.NICH, 0)], 40), ([(orc.BB, 0), (orc.NICH, 0), (orc.DD, 4)], 100),
                                     ([(orc.BB, 0), (orc.GP, 0)], 12)])
def test_assignment_ids_outside_the_table_read_as_unassigned(gpu_ctx, specs, K, monkeypatch):
    """z[n] >= K (a caller's slip) must not index the tables: every kernel treats such a row as not assigned, as
    msc_accumulate does -- same scores and same draws as with -1 in its place"""
    import common_amd
    rng = np.random.default_rng(K)
    N = 3000
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z = rng.integers(0, K, N).astype(np.int32)
    bad = rng.choice(N, 200, replace=False)
    z_bad, z_neg = z.copy(), z.copy()
    z_bad[bad] = rng.choice([K, K + 1, 5 * K + 7, 2 ** 20, 2 ** 31 - 1], 200)
    z_neg[bad] = -1
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    outs = {}
    for name, zz in (("bad", z_bad), ("neg", z_neg)):
        st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
        for i, f in enumerate(feats):
            st.set_hp(i, orc.Family(f["family"], f["hp"], f["dim"], "f64").hp)
        zt = torch.from_numpy(zz).to(gpu_ctx.torch_device)
        st.accumulate(view, zt)
        st.set_alpha(1.5)
        counts = st.get_group_counts().copy()
        loo = st.score_value(view, z=zt).cpu().numpy()
        picks = []
        for pin in ("1", "2"):                                   # both single-nich sweep kernels where they apply
            monkeypatch.setenv("MSC_SWEEP_NICH1", pin)
            zs = zt.clone()
            st.sweep_assign(view, zs, seed=3, sweep=1)
            picks.append(zs.cpu().numpy())
        outs[name] = (counts, loo, picks)
    assert np.array_equal(outs["bad"][0], outs["neg"][0])
    assert np.array_equal(outs["bad"][1], outs["neg"][1])
    for a, b in zip(outs["bad"][2], outs["neg"][2]):
        assert np.array_equal(a, b) and a.min() >= 0 and a.max() < K


def test_device_columns_must_be_aligned_to_their_element_size(gpu_ctx):
    import ctypes as C
    from common_amd import _lib as L
    buf = torch.zeros(64, dtype=torch.uint8, device=gpu_ctx.torch_device)
    rt = (L.RuntimeType * 1)(L.RuntimeType(L.TYPE_F32, 1))
    h = C.c_void_p()
    for off, ok in ((0, True), (1, False), (2, False), (4, True)):
        ptrs = (C.c_void_p * 1)(buf.data_ptr() + off)
        rc = gpu_ctx.lib.msc_dataview_from_device_columns(gpu_ctx._h, 8, rt, 1, ptrs, None, C.byref(h))
        assert (rc == 0) == ok, (off, rc)
        if rc == 0:
            gpu_ctx.lib.msc_dataview_destroy(h)


@pytest.mark.parametrize("offset", [0, 1, 2, 3, 5])
def test_bool_columns_at_any_byte_offset(gpu_ctx, offset):
    """the tile kernels fetch a bool value as the aligned dword around its byte: every base alignment, rows up to the
    last byte of the buffer"""
    import common_amd
    rng = np.random.default_rng(offset)
    N, K = 1003, 70                                                   # K > 64: the tile kernel, not k_narrow
    feats = [make_feature(orc.BB, N, K, rng), make_feature(orc.BB, N, K, rng), make_feature(orc.NICH, N, K, rng)]
    z = rng.integers(0, K, N).astype(np.int32)
    fs = state_from_assignment(feats, K, z)
    dev = gpu_ctx.torch_device
    cols = []
    for f in feats[:2]:
        raw = torch.zeros(offset + N, dtype=torch.bool, device=dev)   # the column ends with the buffer
        raw[offset:] = torch.from_numpy(f["values"].astype(np.bool_)).to(dev)
        cols.append(raw[offset:])
    cols.append(torch.from_numpy(feats[2]["values"]).to(dev))
    view = common_amd.DataView.from_tensors(gpu_ctx, cols)
    st = common_amd.State(gpu_ctx, [(orc.BB, 0), (orc.BB, 0), (orc.NICH, 0)], K)
    load_state(st, fs)
    got = st.score_value(view).cpu().numpy()
    assert rel_err(got, oracle_scores(feats, fs)).max() <= TOL
    for row0, nrows in ((1, 129), (5, 998), (1002, 1)):               # ranges that start and end anywhere
        part = st.score_value(view, row0=row0, nrows=nrows).cpu().numpy()
        assert np.array_equal(part, got[row0:row0 + nrows])
