"""GPU parity: msc_score_value through the C-ABI against the double twin of the oracle."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.gpu_helpers import (TOL, audit, crp_prior_matrix, load_state, make_feature, oracle_scores,
                               recarray_of, rel_err, state_from_assignment)

pytestmark = pytest.mark.gpu

SINGLE = [(orc.BB, 0), (orc.BBNC, 0), (orc.GP, 0), (orc.DD, 5), (orc.DD, 128), (orc.NICH, 0), (orc.NIW, 3), (orc.NIW, 32)]


def _setup(gpu_ctx, specs, N, K, seed, empty_groups=0):
    import common_amd
    rng = np.random.default_rng(seed)
    feats = [make_feature(fam, N, K, rng, dim) for fam, dim in specs]
    z = rng.integers(0, K - empty_groups, N).astype(np.int32)
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    load_state(st, fs)
    counts = np.bincount(z, minlength=K).astype(np.uint32)
    st.set_group_counts(counts)
    return feats, z, fs, view, st, counts


@pytest.mark.parametrize("fam,dim", SINGLE)
@pytest.mark.parametrize("K", [1, 5, 256, 300])
def test_score_value_single_feature(gpu_ctx, fam, dim, K):
    N = 777
    feats, z, fs, view, st, _ = _setup(gpu_ctx, [(fam, dim)], N, K, seed=100 + K + fam)
    got = st.score_value(view).cpu().numpy()
    want = oracle_scores(feats, fs)
    assert got.shape == (N, K)
    assert rel_err(got, want).max() <= TOL


@pytest.mark.parametrize("fam,dim", SINGLE)
def test_score_value_leave_one_out(gpu_ctx, fam, dim):
    N, K = 500, 37
    feats, z, fs, view, st, _ = _setup(gpu_ctx, [(fam, dim)], N, K, seed=7 + fam)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    got = st.score_value(view, z=zt).cpu().numpy()
    want = oracle_scores(feats, fs, z=z)
    assert rel_err(got, want).max() <= TOL
    # unassigned rows (z = -1) are scored against the untouched groups
    z2 = z.copy()
    z2[::3] = -1
    got2 = st.score_value(view, z=torch.from_numpy(z2).to(gpu_ctx.torch_device)).cpu().numpy()
    plain = oracle_scores(feats, fs)
    assert rel_err(got2[::3], plain[::3]).max() <= TOL
    assert rel_err(got2[1::3], want[1::3]).max() <= TOL


def test_gp_large_counts_take_the_saddle_point_path(gpu_ctx):
    """counts >= 32 leave the exact table (family_math.hpp GP_TABLE) for Loader's form"""
    import common_amd
    rng = np.random.default_rng(77)
    N, K = 900, 40
    z = rng.integers(0, K - 2, N).astype(np.int32)
    lam = np.concatenate([rng.gamma(2.0, 2.0, K // 2), rng.uniform(20, 3000, K - K // 2)])
    vals = rng.poisson(lam[z]).astype(np.uint32)
    vals[:8] = [0, 31, 32, 33, 1000, 65535, 1 << 20, 5]
    feats = [dict(family=orc.GP, dim=0, hp=dict(alpha=0.7, inv_beta=0.3), values=vals, np_dtype=np.uint32)]
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(orc.GP, 0)], K)
    load_state(st, fs)
    got = st.score_value(view).cpu().numpy()
    assert rel_err(got, oracle_scores(feats, fs)).max() <= TOL
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    got = st.score_value(view, z=zt).cpu().numpy()
    assert rel_err(got, oracle_scores(feats, fs, z=z)).max() <= TOL


def test_niw_mixed_with_scalar_families_and_leave_one_out(gpu_ctx):
    specs = [(orc.NICH, 0), (orc.NIW, 5), (orc.BB, 0), (orc.NIW, 32)]
    N, K = 700, 21            # K not a multiple of 8: exercises the tail of the niw store
    feats, z, fs, view, st, counts = _setup(gpu_ctx, specs, N, K, seed=44, empty_groups=2)
    got = st.score_value(view).cpu().numpy()
    assert rel_err(got, oracle_scores(feats, fs)).max() <= TOL
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    got = st.score_value(view, z=zt, crp_prior=True).cpu().numpy()
    want = oracle_scores(feats, fs, z=z) + crp_prior_matrix(counts, 1.0, z)
    assert rel_err(got, want).max() <= TOL


def test_niw_f32_matrix_pipe_is_the_documented_looser_variant(gpu_ctx):
    feats, z, fs, view, st, _ = _setup(gpu_ctx, [(orc.NIW, 32)], 640, 48, seed=61)
    want = oracle_scores(feats, fs)
    fast = st.score_value(view, niw_f32=True).cpu().numpy()
    exact = st.score_value(view).cpu().numpy()
    assert rel_err(exact, want).max() <= TOL
    assert rel_err(fast, want).max() <= 2e-5      # MSC_SCORE_NIW_F32: c1 * eps(q), see kernels_niw.hip


def test_score_value_mixed_features_sum_over_columns(gpu_ctx):
    specs = [(orc.BB, 0), (orc.GP, 0), (orc.DD, 32), (orc.NICH, 0)] * 3
    N, K = 600, 300
    feats, z, fs, view, st, _ = _setup(gpu_ctx, specs, N, K, seed=3)
    got = st.score_value(view).cpu().numpy()
    want = oracle_scores(feats, fs)
    assert rel_err(got, want).max() <= TOL
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    got = st.score_value(view, z=zt).cpu().numpy()
    assert rel_err(got, oracle_scores(feats, fs, z=z)).max() <= TOL


@pytest.mark.parametrize("specs", [[(orc.NICH, 0)], [(orc.BB, 0), (orc.NICH, 0), (orc.DD, 7)]])
def test_crp_prior_and_empty_groups(gpu_ctx, specs):
    N, K, alpha = 400, 19, 2.5
    feats, z, fs, view, st, counts = _setup(gpu_ctx, specs, N, K, seed=21, empty_groups=3)
    # make one group a singleton so that removing its row empties it
    st.set_alpha(alpha)
    got = st.score_value(view, crp_prior=True).cpu().numpy()
    want = oracle_scores(feats, fs) + crp_prior_matrix(counts, alpha)
    assert rel_err(got, want).max() <= TOL
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    got = st.score_value(view, z=zt, crp_prior=True).cpu().numpy()
    want = oracle_scores(feats, fs, z=z) + crp_prior_matrix(counts, alpha, z)
    assert rel_err(got, want).max() <= TOL


def test_singleton_group_becomes_empty_under_leave_one_out(gpu_ctx):
    import common_amd
    rng = np.random.default_rng(5)
    N, K, alpha = 64, 6, 1.5
    feats = [make_feature(orc.NICH, N, K, rng)]
    z = rng.integers(0, 3, N).astype(np.int32)
    z[10] = 4                      # group 4 holds exactly one row; group 3 and 5 are empty
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(orc.NICH, 0)], K)
    load_state(st, fs)
    counts = np.bincount(z, minlength=K).astype(np.uint32)
    st.set_group_counts(counts)
    st.set_alpha(alpha)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    got = st.score_value(view, z=zt, crp_prior=True).cpu().numpy()
    want = oracle_scores(feats, fs, z=z) + crp_prior_matrix(counts, alpha, z)
    assert rel_err(got, want).max() <= TOL
    # row 10 sees three empty groups (3, 4, 5), every other row two
    assert abs((got[10, 3] - oracle_scores(feats, fs, z=z)[10, 3]) - np.log(alpha / 3)) < 1e-5


def test_rows_subrange_and_leading_dimension(gpu_ctx):
    N, K = 1000, 20
    feats, z, fs, view, st, _ = _setup(gpu_ctx, [(orc.NICH, 0), (orc.BB, 0)], N, K, seed=9)
    out = torch.full((300, 32), -7.0, dtype=torch.float32, device=gpu_ctx.torch_device)
    st.score_value(view, out=out, row0=123, nrows=300)
    got = out.cpu().numpy()
    want = oracle_scores(feats, fs, rows=slice(123, 423))
    assert rel_err(got[:, :K], want).max() <= TOL
    assert np.all(got[:, K:] == -7.0)  # padding of the leading dimension is left alone


def test_golden_vectors_through_the_device(gpu_ctx):
    """the scipy known answers (tests/golden) reproduced by the HIP path itself"""
    import common_amd
    from tests.conftest import load_golden
    for name, fam in (("bb", orc.BB), ("gp", orc.GP), ("dd", orc.DD), ("nich", orc.NICH), ("niw", orc.NIW)):
        for case in load_golden(name):
            dim = case.get("dim", 0)
            st = common_amd.State(gpu_ctx, [(fam, dim)], 1)
            st.set_hp(0, case["hp"])
            rec = np.zeros(1, dtype=common_amd.ss_dtype(fam, dim))
            for k, v in case["ss"].items():
                rec[k][0] = np.asarray(v)
            st.set_ss(0, rec)
            probe = np.asarray(case["probe"]).astype(orc.value_dtype(fam, dim).base)
            if fam == orc.BB:
                probe = probe.astype(np.bool_)
            arr = np.zeros(len(probe), dtype=[("f0", probe.dtype, probe.shape[1:])])
            arr["f0"] = probe
            view = common_amd.DataView.from_recarray(gpu_ctx, arr)
            got = st.score_value(view).cpu().numpy()[:, 0]
            # the device holds the suff-stats in float (as the reference does): compare with the
            # twin fed the same float state, and loosely with scipy on the unrounded state
            F = orc.Family(fam, case["hp"], dim, "f64")
            ss64 = np.zeros(1, dtype=orc.ss_dtype(fam, dim, "f64"))
            for k in rec.dtype.names:
                ss64[k] = rec[k]
            want = F.score_matrix(ss64, probe.astype(orc.value_dtype(fam, dim).base))[:, 0]
            assert rel_err(got, want).max() <= TOL, (name, got, want)
            if not (name == "nich" and abs(case["ss"]["mean"]) > 100):
                assert rel_err(got, case["score_value"]).max() <= 5e-5, name
            sd = st.score_data().cpu().numpy()[0, 0]
            assert rel_err(sd, F.score_data(ss64, 0)) <= TOL


@pytest.mark.parametrize("N,K", [(40_000, 100), (70_000, 300)])
def test_leave_one_out_on_the_large_tiling(gpu_ctx, N, K):
    """N >= 32k rows takes the 8-rows-per-wave tiling, whose own-group entries are written by k_loo_patch
    (K = 300: two k-tiles); a singleton group exercises the per-row empty-group prior"""
    import common_amd
    rng = np.random.default_rng(N)
    specs = [(orc.BB, 0), (orc.GP, 0), (orc.DD, 7), (orc.NICH, 0), (orc.BNB, 0), (orc.NICH, 0)]
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z = rng.integers(0, K - 3, N).astype(np.int32)
    z[z == K - 4] = 0
    z[12345] = K - 4                                       # a group with exactly one member
    z[77] = -1                                             # an unassigned row
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    load_state(st, fs)
    counts = np.bincount(z[z >= 0], minlength=K)
    st.set_group_counts(counts.astype(np.uint32))
    st.set_alpha(0.9)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    got = st.score_value(view, z=zt, crp_prior=True)
    rows = np.concatenate([[12345, 77, 0, N - 1], rng.choice(N, 300, replace=False)])
    want = oracle_scores(feats, fs, z=z, rows=rows) + crp_prior_matrix(counts, 0.9, z[rows])
    assert rel_err(got[torch.from_numpy(rows).to(gpu_ctx.torch_device)].cpu().numpy(), want).max() <= TOL
    plain = st.score_value(view, crp_prior=True)
    other = torch.ones((N, K), dtype=torch.bool, device=gpu_ctx.torch_device)
    ok = zt >= 0
    other[ok.nonzero().squeeze(1), zt[ok].long()] = False
    other[12345] = False                                   # (its empty-group columns carry the other prior)
    assert torch.equal(got[other], plain[other])           # everything but the own entries: the same bits


TAIL_PLANS = {
    "mixed": [(orc.BB, 0), (orc.GP, 0), (orc.DD, 7), (orc.NICH, 0), (orc.NICH, 0), (orc.DD, 33), (orc.BBNC, 0)],
    "lookups_only": [(orc.BB, 0), (orc.DD, 12), (orc.GP, 0)],
    "nich_only": [(orc.NICH, 0), (orc.NICH, 0), (orc.NICH, 0)],
    "many_stages": [(orc.DD, 60)] * 9 + [(orc.NICH, 0)] * 9,      # 540 table rows: three stages of the kernel's slot; two value batches
    "many_nich": [(orc.DD, 60), (orc.GP, 0)] + [(orc.NICH, 0)] * 70,   # the second phase's block takes 17.5 KiB of the slot's 64
}


@pytest.mark.parametrize("plan", sorted(TAIL_PLANS))
@pytest.mark.parametrize("K", [40, 100, 128, 257, 270, 285, 300, 320, 321, 345, 384])
def test_partly_filled_last_tile_on_the_narrow_kernel(gpu_ctx, plan, K, monkeypatch):
    """A last tile of at most 128 groups (K <= 128: the whole state; 256 < K <= 384: the groups beyond the first tile) comes
    from k_score_tail_rows when the rows are many (lane <-> row; launches of 16 / 32 / 48 groups: every register tiling
    of it), whatever the plan holds: lookup features and nich features, one kind only, more table rows than one stage of its
    slot.  Plain and leave-one-out + prior against the oracle on sampled rows; rows whose own group lies in the tail, a
    singleton tail group, an empty tail group and an unassigned row included; a row range of few rows (row0 > 0) -- which
    the tile kernels score -- gives the same bits as the whole."""
    import common_amd
    if plan == "many_nich" and K not in (100, 300):
        pytest.skip("the wide plan at two table sizes")
    monkeypatch.setenv("MSC_TAIL_MIN_ROWS", "16384")       # (the library's own mark grows with the groups: ~1000 rows a group)
    N = 17_000
    rng = np.random.default_rng(K + len(plan))
    specs = TAIL_PLANS[plan]
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z = rng.integers(0, K, N).astype(np.int32)
    z[z == K - 1] = 0                                      # the last group: empty
    if K != 257:
        z[z == K - 2] = 1
        z[1234] = K - 2                                    # a tail group with exactly one member
    z[77] = -1
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    load_state(st, fs)
    counts = np.bincount(z[z >= 0], minlength=K)
    st.set_group_counts(counts.astype(np.uint32))
    st.set_alpha(1.4)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    in_tail = np.flatnonzero(z >= (256 if K > 256 else 0))[:40]
    rows = np.unique(np.concatenate([[1234, 77, 0, N - 1], in_tail, rng.choice(N, 120, replace=False)]))
    rt = torch.from_numpy(rows).to(gpu_ctx.torch_device)
    plain = st.score_value(view)
    assert rel_err(plain[rt].cpu().numpy(), oracle_scores(feats, fs, rows=rows)).max() <= TOL
    got = st.score_value(view, z=zt, crp_prior=True)
    want = oracle_scores(feats, fs, z=z, rows=rows) + crp_prior_matrix(counts, 1.4, z[rows])
    assert rel_err(got[rt].cpu().numpy(), want).max() <= TOL
    part = torch.full((1531, K), -7.0, dtype=torch.float32, device=gpu_ctx.torch_device)
    st.score_value(view, out=part, z=zt[700:700 + 1531].contiguous(), crp_prior=True, row0=700, nrows=1531)
    assert torch.equal(part, got[700:700 + 1531])


@pytest.mark.parametrize("plan", ["mixed", "nich_only", "lookups_only"])
@pytest.mark.parametrize("K", [70, 128, 270, 300, 384])
def test_narrow_kernel_gives_the_tile_kernels_bits(gpu_ctx, plan, K, monkeypatch):
    """k_score_tail_rows sums as score_tile does -- (prior lo + lookups) + (c0 sum, then nich_accum per feature), prior hi
    last -- so the choice between it and the tile kernels is free to depend on the row count: the same state planned with
    and without it (MSC_NO_NARROW_TAIL, read when a state plans its kernels) scores every entry to the same bits."""
    import common_amd
    monkeypatch.setenv("MSC_TAIL_MIN_ROWS", "16384")
    N = 16_400
    rng = np.random.default_rng(K)
    specs = TAIL_PLANS[plan]
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z = rng.integers(0, K - 1, N).astype(np.int32)
    z[11] = -1
    fs = state_from_assignment(feats, K, z)
    counts = np.bincount(z[z >= 0], minlength=K)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    got = {}
    for narrow in (True, False):
        if narrow:
            monkeypatch.delenv("MSC_NO_NARROW_TAIL", raising=False)
        else:
            monkeypatch.setenv("MSC_NO_NARROW_TAIL", "1")
        view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
        st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
        load_state(st, fs)
        st.set_group_counts(counts.astype(np.uint32))
        st.set_alpha(0.7)
        got[narrow] = (st.score_value(view), st.score_value(view, z=zt, crp_prior=True), st.score_value(view, crp_prior=True))
    for a, b in zip(got[True], got[False]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dim", [1, 2, 7, 8, 9, 16, 17, 20, 24, 31, 32, 33, 48, 64, 100, 128])
@pytest.mark.parametrize("K", [3, 70, 130])
def test_niw_every_kernel_by_dimension(gpu_ctx, dim, K):
    """dim <= 8 runs the per-lane-group vector kernel (64-group tiles: K = 70 and 130 end in partial tiles), 9 .. 128
    the f64 MFMA kernel over the 16-blocks of the triangular factor (1 .. 8 blocks; dimensions that are not a multiple
    of 4 / of 16 end in partial steps / blocks): NormalInverseWishart<-1> has no dimension cap upstream
    (distributions.hpp:87-91,481-509).  Plain, leave-one-out, accumulated on top of another feature, masked rows"""
    import common_amd
    N = 900 if dim <= 32 else 260
    if dim > 32 and K == 130:
        K = 40                                                  # (the oracle refactors per pair: keep the CPU side in seconds)
    rng = np.random.default_rng(1000 * dim + K)
    feats = [make_feature(orc.BB, N, K, rng), make_feature(orc.NIW, N, K, rng, dim)]
    z = rng.integers(0, K, N).astype(np.int32)
    z[5] = -1
    fs = state_from_assignment(feats, K, z)
    rec = recarray_of(feats)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    # the niw feature alone (writes the matrix) and after a bb feature (adds to it)
    for cols in ([1], [0, 1]):
        sub = [feats[c] for c in cols]
        view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(sub))
        st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in sub], K)
        load_state(st, [fs[c] for c in cols])
        st.set_group_counts(np.bincount(z[z >= 0], minlength=K).astype(np.uint32))
        assert rel_err(st.score_value(view).cpu().numpy(), oracle_scores(sub, [fs[c] for c in cols])).max() <= TOL
        assert rel_err(st.score_value(view, z=zt).cpu().numpy(), oracle_scores(sub, [fs[c] for c in cols], z=z)).max() <= TOL
    # masked vectors contribute nothing
    mask = np.zeros(N, dtype=[("f0", np.bool_), ("f1", np.bool_, (dim,))])
    hide = rng.random(N) < 0.3
    mask["f1"][hide, rng.integers(0, dim, hide.sum())] = True
    view = common_amd.DataView.from_recarray(gpu_ctx, np.ma.masked_array(rec, mask=mask))
    st = common_amd.State(gpu_ctx, [(orc.BB, 0), (orc.NIW, dim)], K)
    load_state(st, fs)
    got = st.score_value(view).cpu().numpy()
    want = fs[0][0].score_matrix(fs[0][1], feats[0]["values"])
    niw = fs[1][0].score_matrix(fs[1][1], feats[1]["values"])
    niw[hide] = 0.0
    assert rel_err(got, want + niw).max() <= TOL


BITS_PLANS = {
    "mixed": [(orc.BB, 0), (orc.GP, 0), (orc.NICH, 0), (orc.DD, 7), (orc.NICH, 0), (orc.BB, 0), (orc.NICH, 0)],
    "nich_only": [(orc.NICH, 0)] * 5,       # (no first phase at all: k_score_nich_pack, every wave a nich wave; blocks of 3 + 2)
    "lookups_only": [(orc.BB, 0)] * 5 + [(orc.DD, 9), (orc.GP, 0), (orc.BB, 0), (orc.DD, 70)],   # (k_score_lookups: every wave a lookup wave)
}


@pytest.mark.parametrize("plan", sorted(BITS_PLANS))
@pytest.mark.parametrize("K", [65, 100, 127, 128, 256, 300])
def test_a_rows_score_is_the_same_bits_from_every_tile_kernel(gpu_ctx, K, plan):
    """40k rows take the kernel whose waves split the lookup and the nich phase between them (k_score_tile_roles; up to 128
    groups in its PAIR mode: two groups a lane, two rows a float4 of sums) -- a state of plain nich features alone the one
    whose waves are all nich waves (k_score_nich_pack), one of staged lookup features alone the one whose waves are all
    lookup waves (k_score_lookups) --, a few hundred rows the ones that run the phases one after the other: (prior + lookups) + (nich) in all of them, so the same row must come out bit for bit -- plain, leave-one-out,
    with the prior"""
    import common_amd
    rng = np.random.default_rng(K)
    N = 40_000
    specs = BITS_PLANS[plan]
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z = rng.integers(0, K, N).astype(np.int32)
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, specs, K)
    load_state(st, fs)
    st.set_group_counts(np.bincount(z, minlength=K).astype(np.uint32))
    st.set_alpha(1.2)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    for kw in ({}, {"z": zt}, {"z": zt, "crp_prior": True}):
        whole = st.score_value(view, **kw)
        for row0, n in ((0, 200), (12_345, 64), (39_900, 100)):
            kk = dict(kw)
            if "z" in kk:
                kk["z"] = zt[row0:row0 + n].contiguous()
            part = st.score_value(view, row0=row0, nrows=n, **kk)
            assert torch.equal(part, whole[row0:row0 + n]), (kw.keys(), row0)
    got = st.score_value(view).cpu().numpy()
    rows = rng.choice(N, 300, replace=False)
    assert rel_err(got[rows], oracle_scores(feats, fs, rows=rows)).max() <= TOL


@pytest.mark.parametrize("seed", range(4))
def test_wave_role_kernels_on_random_feature_lists(gpu_ctx, seed):
    """random lists of lookup + nich features, group counts over one to four 256-group tiles, row counts just past the
    point where the role-split kernels take over and not a multiple of anything: sampled rows against the oracle,
    slices against the whole (bit for bit), leave-one-out + prior included"""
    import common_amd
    rng = np.random.default_rng(500 + seed)
    K = int(rng.choice([70, 256, 257, 600, 1000]))
    ktiles = (K + 255) // 256
    N = 128 * 256 // ktiles + int(rng.integers(1, 3000))
    fams = [(orc.BB, 0), (orc.GP, 0), (orc.DD, int(rng.integers(2, 40))), (orc.NICH, 0)]
    specs = [fams[i] for i in rng.integers(0, 4, int(rng.integers(3, 12)))] + [(orc.NICH, 0), (orc.BB, 0)]
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z = rng.integers(0, K, N).astype(np.int32)
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, specs, K)
    load_state(st, fs)
    st.set_group_counts(np.bincount(z, minlength=K).astype(np.uint32))
    st.set_alpha(0.7)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    rows = rng.choice(N, 200, replace=False)
    plain = st.score_value(view)
    assert rel_err(plain.cpu().numpy()[rows], oracle_scores(feats, fs, rows=rows)).max() <= TOL
    loo = st.score_value(view, z=zt)
    assert rel_err(loo.cpu().numpy()[rows], oracle_scores(feats, fs, z=z, rows=rows)).max() <= TOL
    for row0, n in ((0, 129), (N - 77, 77), (N // 2 + 3, 500)):
        assert torch.equal(st.score_value(view, row0=row0, nrows=n), plain[row0:row0 + n])
        assert torch.equal(st.score_value(view, row0=row0, nrows=n, z=zt[row0:row0 + n].contiguous(), crp_prior=True),
                           st.score_value(view, z=zt, crp_prior=True)[row0:row0 + n])


@pytest.mark.parametrize("K", [60, 256, 400])
def test_both_leave_one_out_kernels_give_a_row_the_same_bits(gpu_ctx, monkeypatch, K):
    """k_loo_own (a gather per entry) and k_loo_own_lds (the plan's stages through LDS; taken from one workgroup per CU
    on) feed the same `own` value into the score and sweep kernels: same terms, same order of the double sum, so the
    leave-one-out + prior scores must be equal bit for bit -- a shard of any size reproduces the unsharded pass -- and
    within the gate of the oracle.  Mixed features incl. a masked column and a dd wider than the staged rows."""
    import common_amd
    rng = np.random.default_rng(K)
    N = 30_000
    specs = [(orc.BB, 0), (orc.GP, 0), (orc.NICH, 0), (orc.DD, 7), (orc.NICH, 0), (orc.BB, 0), (orc.DD, 40), (orc.BNB, 0),
             (orc.NICH, 0), (orc.BBNC, 0)]
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z = rng.integers(0, K, N).astype(np.int32)
    z[::17] = -1
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, specs, K)
    load_state(st, fs)
    st.set_group_counts(np.bincount(z[z >= 0], minlength=K).astype(np.uint32))
    st.set_alpha(1.3)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    monkeypatch.setenv("MSC_LOO_LDS", "0")
    gathered = st.score_value(view, z=zt, crp_prior=True).clone()
    monkeypatch.setenv("MSC_LOO_LDS", "1")
    staged = st.score_value(view, z=zt, crp_prior=True).clone()
    assert torch.equal(gathered, staged)
    rows = rng.choice(N, 300, replace=False)
    want = oracle_scores(feats, fs, z=z, rows=rows) + crp_prior_matrix(np.bincount(z[z >= 0], minlength=K), 1.3, z[rows])
    lik = oracle_scores(feats, fs, z=z, rows=rows)
    got = staged.cpu().numpy()[rows]
    assert (np.abs(got - want) / np.maximum(1.0, np.maximum(np.abs(want), np.abs(lik)))).max() <= TOL


@pytest.mark.parametrize("nbb,K", [(2, 70), (3, 100), (5, 300), (9, 256), (16, 40)])
def test_fused_bb_columns_follow_every_table_update(gpu_ctx, nbb, K, monkeypatch):
    """The score / sweep plan fuses a state's unmasked bb / bbnc columns four (three, two) at a time -- one byte column of
    their bits, one table of 2^m rows that k_fuse_tables rebuilds from the members' tables at the head of a call
    (abi.cpp plan_groups).  Interleaved with other families, a masked bb column left out, 2 / 3 / 5 / 9 / 16 of them:
    against the oracle after accumulate, after entity moves (k_entity_op updates the members' tables in place), after a
    sweep step (k_commit_prepare) and after set_ss; few rows (tile kernels) and many (the lane <-> row kernel where it
    applies) agree to the bit; the unfused plan (MSC_NO_BB_FUSE) agrees within the gate."""
    import common_amd
    monkeypatch.setenv("MSC_TAIL_MIN_ROWS", "16384")
    rng = np.random.default_rng(nbb * 1000 + K)
    N = 17_000
    specs = []
    for i in range(nbb):
        specs += [(orc.BB, 0) if i % 3 else (orc.BBNC, 0), (orc.GP, 0) if i % 2 else (orc.DD, 6)]
    nmasked = {2: 1, 3: 2, 5: 3, 9: 4, 16: 7}[nbb]                  # masked bb columns: one alone stays unfused; 2 / 3 / 4 / 7
    specs += [(orc.BB, 0)] * nmasked + [(orc.NICH, 0), (orc.NICH, 0)]   # fuse three states a value, three (two) at a time
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z = rng.integers(0, K - 1, N).astype(np.int32)
    masks = [np.zeros(N, dtype=bool) for _ in feats]
    for c in range(len(specs) - 2 - nmasked, len(specs) - 2):
        masks[c] = rng.random(N) < 0.25
    dev = gpu_ctx.torch_device
    cols = [torch.from_numpy(np.ascontiguousarray(f["values"])).to(dev) for f in feats]
    mts = [torch.from_numpy(m.astype(np.uint8)).to(dev) if m.any() else None for m in masks]
    view = common_amd.DataView.from_tensors(gpu_ctx, cols, mts)
    rows = np.unique(np.concatenate([[0, N - 1], rng.choice(N, 150, replace=False)]))
    rt = torch.from_numpy(rows).to(dev)

    def check(st, what):
        """the state's scores against the twin evaluated on the suff-stats the state holds"""
        want, mag = None, None
        for i, (f, m) in enumerate(zip(feats, masks)):
            F = orc.Family(f["family"], st.get_hp(i), f["dim"], "f64")
            held = orc.widen_ss(f["family"], st.get_ss(i).astype(orc.ss_dtype(f["family"], f["dim"], "f32")), f["dim"])
            sc = F.score_matrix(held, f["values"][rows])
            sc[m[rows]] = 0.0
            want = sc if want is None else want + sc
            mag = np.maximum(1.0, np.abs(sc)) if mag is None else mag + np.maximum(1.0, np.abs(sc))
        big = st.score_value(view)
        assert (np.abs(big[rt].cpu().numpy() - want) / mag).max() <= TOL, what
        few = st.score_value(view, row0=700, nrows=900)             # few rows: the tile kernels
        assert torch.equal(few, big[700:1600]), what
        return big

    st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    for i, f in enumerate(feats):
        st.set_hp(i, f["hp"])
    st.set_alpha(1.2)
    zt = torch.from_numpy(z).to(dev)
    st.accumulate(view, zt)
    for i, (fam, _) in enumerate(specs):                            # (a bbnc group's p is the state's own: give it values)
        if fam == orc.BBNC:
            rec = st.get_ss(i)
            rec["p"] = rng.uniform(0.05, 0.95, K).astype(np.float32)
            st.set_ss(i, rec)
    fused = check(st, "accumulate")
    for n in rng.choice(N, 40, replace=False):                      # entity moves
        st.entity_op(view, int(n), int(z[n]), join=False, z=zt)
        z[n] = int(rng.integers(0, K))
        st.entity_op(view, int(n), int(z[n]), join=True, z=zt)
    check(st, "entity_op")
    st.sweep_step(view, zt, seed=5, sweep=0)                        # draw + tables rebuilt (k_commit_prepare)
    check(st, "sweep_step")
    rec = st.get_ss(0)
    rec["heads"] += 3
    st.set_ss(0, rec)
    check(st, "set_ss")
    # the unfused plan: another association of the same terms
    monkeypatch.setenv("MSC_NO_BB_FUSE", "1")
    st2 = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    for i, f in enumerate(feats):
        st2.set_hp(i, f["hp"])
    monkeypatch.delenv("MSC_NO_BB_FUSE")
    st3 = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    for i, f in enumerate(feats):
        st3.set_hp(i, f["hp"])
    for i in range(len(feats)):                                     # the same tables in both
        st2.set_ss(i, st.get_ss(i))
        st3.set_ss(i, st.get_ss(i))
    a, b = st2.score_value(view)[rt].cpu().numpy(), st3.score_value(view)[rt].cpu().numpy()
    assert (np.abs(a - b) / np.maximum(1.0, np.abs(a))).max() <= 1e-5
    assert np.array_equal(b, st.score_value(view)[rt].cpu().numpy())
    assert fused.shape == (N, K)


# ---- nich BLOCKS (family_math.hpp): plain nich features that share c1 are scored as one log1p of a product -------------
def _nich_block_state(gpu_ctx, specs, N, K, rng, hp_of=None, z_of=None, edit=None):
    """state + view of `specs`; hp_of[i]: that feature's hp; z_of[i]: the assignment ITS suff-stats come from (default: z)"""
    import common_amd
    feats = [make_feature(f, N, K, rng, d, hp=(hp_of or {}).get(i)) for i, (f, d) in enumerate(specs)]
    if edit:
        edit(feats)
    z = rng.integers(0, K, N).astype(np.int32)
    fs = []
    for i, f in enumerate(feats):
        fs.append(state_from_assignment([f], K, (z_of or {}).get(i, z))[0])
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, specs, K)
    load_state(st, fs)
    st.set_group_counts(np.bincount(z, minlength=K).astype(np.uint32))
    st.set_alpha(1.1)
    return feats, fs, z, view, st


def _gate_on_sum(got, feats, fs, rows, z=None, prior=None):
    """|got - sum_f twin_f| <= 1e-6 sum_f max(1, |twin_f|) (+ the prior's magnitude): the per-feature tolerances add"""
    tw = [F.score_matrix(ss64, f["values"][rows], None if z is None else z[rows]) for f, (F, ss64, _) in zip(feats, fs)]
    total = sum(tw) + (0.0 if prior is None else prior)
    mag = sum(np.maximum(1.0, np.abs(t)) for t in tw) + (0.0 if prior is None else np.maximum(1.0, np.abs(prior)))
    return (np.abs(got - total) / np.maximum(mag, np.abs(total))).max()


@pytest.mark.parametrize("K,lane_row", [(48, True), (100, True), (100, False), (128, False), (256, True), (300, True)])
@pytest.mark.parametrize("nnich", [2, 3, 4, 5, 9, 16])
def test_nich_blocks_of_every_size_against_the_twin(gpu_ctx, K, lane_row, nnich, monkeypatch):
    """two to sixteen plain nich columns beside lookup columns (blocks of 2, 3, 4; 5 = 3 + 2; 9 = 3 + 3 + 3; 16 = 4 x 4) on
    every kernel that walks the plan: few rows (tile kernels, phases one after the other), many (role-split -- up to 128
    groups in PAIR mode when the lane <-> row kernel is not forced -- / lane <-> row), leave-one-out + prior.  The gate is
    the north star's on a sum of features; slices of the whole come out bit for bit."""
    if lane_row:
        monkeypatch.setenv("MSC_TAIL_MIN_ROWS", "16384")
    rng = np.random.default_rng(1000 * K + nnich)
    N = 40_000
    specs = [(orc.BB, 0), (orc.GP, 0)] + [(orc.NICH, 0)] * nnich + [(orc.DD, 9)]
    feats, fs, z, view, st = _nich_block_state(gpu_ctx, specs, N, K, rng)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    rows = np.unique(np.concatenate([[0, N - 1], rng.choice(N, 250, replace=False)]))
    rt = torch.from_numpy(rows).to(gpu_ctx.torch_device)
    plain = st.score_value(view)
    audit("nich_blocks.sum_of_features.n%d" % nnich, _gate_on_sum(plain[rt].cpu().numpy(), feats, fs, rows), TOL)
    loo = st.score_value(view, z=zt, crp_prior=True)
    counts = np.bincount(z, minlength=K)
    audit("nich_blocks.sum_of_features_loo_prior.n%d" % nnich,
          _gate_on_sum(loo[rt].cpu().numpy(), feats, fs, rows, z=z, prior=crp_prior_matrix(counts, 1.1, z[rows])), TOL)
    for row0, n in ((0, 300), (N // 2 + 1, 129), (N - 64, 64)):     # (few rows: the tile kernels, one phase after the other)
        assert torch.equal(st.score_value(view, row0=row0, nrows=n), plain[row0:row0 + n])
        assert torch.equal(st.score_value(view, row0=row0, nrows=n, z=zt[row0:row0 + n].contiguous(), crp_prior=True), loo[row0:row0 + n])


def test_nich_blocks_follow_the_nu_prior_and_the_counts_the_suffstats_hold(gpu_ctx, monkeypatch):
    """What makes a block: the same nu prior (the host's plan) AND, per group, the same count (the head kernel looks at the
    c1 the suff-stats produced).  Columns 2-5 share nu = 1 but column 4's suff-stats come from ANOTHER assignment (set
    feature by feature, as msc_state_set_ss allows): its block is scored feature by feature; columns 6, 7 have nu = 3.5 and
    form a block of their own; column 8's nu is its own.  Against the twin, and -- for the features that fell back -- the
    bits of a plan without blocks (MSC_NO_NICH_BLOCKS)."""
    import common_amd
    monkeypatch.setenv("MSC_TAIL_MIN_ROWS", "16384")
    K, N = 200, 30_000
    specs = [(orc.BB, 0), (orc.DD, 5)] + [(orc.NICH, 0)] * 7
    hp = lambda nu: dict(mu=0.3, kappa=1.5, sigmasq=0.8, nu=nu)    # noqa: E731
    hp_of = {2: hp(1.0), 3: hp(1.0), 4: hp(1.0), 5: hp(1.0), 6: hp(3.5), 7: hp(3.5), 8: hp(2.25)}
    seed = 4242
    z_other = np.random.default_rng(99).integers(0, K, N).astype(np.int32)
    feats, fs, z, view, st = _nich_block_state(gpu_ctx, specs, N, K, np.random.default_rng(seed), hp_of=hp_of, z_of={4: z_other})
    got = st.score_value(view)
    rows = np.unique(np.random.default_rng(5).choice(N, 300, replace=False))
    audit("nich_blocks.mixed_nu_and_counts", _gate_on_sum(got[torch.from_numpy(rows).to(got.device)].cpu().numpy(), feats, fs, rows), TOL)
    part = st.score_value(view, row0=1000, nrows=200)              # (the tile kernels on few rows: the same bits)
    assert torch.equal(part, got[1000:1200])
    # four nu = 1 columns alone, one of them holding another assignment's counts: the block falls back to nich_accum
    sub = [(orc.NICH, 0)] * 4
    f4, s4, z4, v4, st4 = _nich_block_state(gpu_ctx, sub, N, K, np.random.default_rng(seed + 1), hp_of={i: hp(1.0) for i in range(4)},
                                            z_of={2: z_other})
    a = st4.score_value(v4)
    tw = [F.score_matrix(ss64, f["values"]) for f, (F, ss64, _) in zip(f4, s4)]
    mag = sum(np.maximum(1.0, np.abs(t)) for t in tw)
    audit("nich_blocks.differing_counts_fall_back", (np.abs(a.cpu().numpy() - sum(tw)) / mag).max(), TOL)
    assert torch.equal(st4.score_value(v4, row0=500, nrows=100), a[500:600])


@pytest.mark.parametrize("K,lane_row", [(64, True), (100, False), (256, True), (290, True)])
def test_far_rows_take_the_plain_path_for_all_their_nich_features(gpu_ctx, K, lane_row, monkeypatch):
    """a value 10^7 posterior scales from the groups (|a| beyond 2^15: four such squares multiplied leave the float range)
    makes its ROW a far row: nich_accum feature by feature for that row, whatever the other rows of its wave do -- so the
    row's bits are the same in a slice of 64 rows and in the whole, on the tile kernels and on the lane <-> row kernel, and
    its scores meet the gate like any other's (they are hugely negative: relative).  (K = 100: the role-split kernels'
    PAIR mode, where a far row is one half of a pair of sums.)"""
    if lane_row:
        monkeypatch.setenv("MSC_TAIL_MIN_ROWS", "16384")
    rng = np.random.default_rng(31 * K)
    N = 20_000 if lane_row else 40_000
    specs = [(orc.BB, 0)] + [(orc.NICH, 0)] * 6 + [(orc.GP, 0)]
    far_rows = np.array([5, 129, 4097, 12_345, N - 1])

    def edit(feats):
        for j, r in enumerate(far_rows):
            feats[1 + j % 6]["values"][r] = np.float32((-1.0) ** j * 3.0e7 * (1 + j))
    feats, fs, z, view, st = _nich_block_state(gpu_ctx, specs, N, K, rng, edit=edit)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    plain = st.score_value(view)
    rows = np.unique(np.concatenate([far_rows, far_rows[:-1] + 1, rng.choice(N, 100, replace=False)]))
    got = plain[torch.from_numpy(rows).to(plain.device)].cpu().numpy()
    assert np.isfinite(got).all()
    audit("nich_blocks.far_rows", _gate_on_sum(got, feats, fs, rows), TOL)
    loo = st.score_value(view, z=zt, crp_prior=True)
    for r in far_rows:
        row0 = max(0, int(r) - 20)
        n = min(64, N - row0)
        assert torch.equal(st.score_value(view, row0=row0, nrows=n), plain[row0:row0 + n])
        assert torch.equal(st.score_value(view, row0=row0, nrows=n, z=zt[row0:row0 + n].contiguous(), crp_prior=True), loo[row0:row0 + n])


def test_a_view_keeps_at_most_eight_index_matrices_and_the_ninth_plan_still_scores_right(gpu_ctx):
    """The lookup index matrix and the x matrix of the role-split kernels are copies a view keeps per distinct feature list
    (at most eight of each: abi.cpp kViewPackCap); a ninth list -- or a failed allocation -- plans the state onto the kernels
    that need none.  Ten states over ten different column subsets of ONE view: every one scores its rows like the oracle,
    and the bits of a state do not depend on whether it got its matrices (the tile kernels agree bit for bit): the ninth and
    tenth against fresh views of their own, where they do get them.  (a copy is an optimisation, never a reason
    for a call to fail.)"""
    import common_amd
    rng = np.random.default_rng(2025)
    N, K = 40_000, 200
    specs = [(orc.DD, 6 + i) for i in range(10)] + [(orc.GP, 0), (orc.BB, 0), (orc.NICH, 0), (orc.NICH, 0), (orc.NICH, 0)]
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z = rng.integers(0, K, N).astype(np.int32)
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    rows = rng.choice(N, 200, replace=False)
    for i in range(10):
        cols = [i, 10, 11, 12, 13, 14]                       # a different dd column each time: a different first phase
        st = common_amd.State(gpu_ctx, [specs[c] for c in cols], K)
        load_state(st, [fs[c] for c in cols])
        got = st.score_value(view, cols=cols)
        want = oracle_scores([feats[c] for c in cols], [fs[c] for c in cols], rows=rows)
        assert rel_err(got.cpu().numpy()[rows], want).max() <= TOL, i
        if i >= 8:                                          # (the view's cap is spent: this plan took the kernels without a matrix)
            own_view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of([feats[c] for c in cols]))
            st2 = common_amd.State(gpu_ctx, [specs[c] for c in cols], K)
            load_state(st2, [fs[c] for c in cols])
            assert torch.equal(st2.score_value(own_view), got), i
