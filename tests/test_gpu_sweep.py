"""GPU parity: the fused Gibbs assignment sweep against the oracle's synchronous sweep."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.gpu_helpers import load_state, make_feature, recarray_of, state_from_assignment

pytestmark = pytest.mark.gpu


def _run(gpu_ctx, specs, N, K, seed, sweep_idx, alpha=1.3, empty=2, row_id0=0, small_dm=False):
    import common_amd
    rng = np.random.default_rng(seed)
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    if small_dm:      # dm counts small enough that the feature's tables are staged whole (row totals of at most ~10)
        feats = [dict(f, values=(f["values"] // 5).astype(np.int32)) if f["family"] == orc.DM else f for f in feats]
    z = rng.integers(0, max(1, K - empty), N).astype(np.int32)
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    load_state(st, fs)
    st.set_group_counts(np.bincount(z, minlength=K).astype(np.uint32))
    st.set_alpha(alpha)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.sweep_assign(view, zt, seed=seed, sweep=sweep_idx, row_id0=row_id0)
    got = zt.cpu().numpy()
    # the oracle draws with uniform01(seed, sweep, global row): shift rows by row_id0 via a wrapper
    feats_o = [(F, ss64, f["values"]) for f, (F, ss64, _) in zip(feats, fs)]
    want, scores = orc.sweep(feats_o, K, alpha, z, seed, sweep_idx, "f64", want_scores=True) \
        if row_id0 == 0 else (None, None)
    return got, want, scores, z


def _check_agreement(got, want, scores, seed, sweep_idx, min_agree):
    agree = (got == want).mean()
    assert agree >= min_agree, agree
    # a disagreement may only be a dart landing within rounding of a CDF step
    for n in np.nonzero(got != want)[0]:
        p = orc.scores_to_probs(scores[n])
        cdf = np.cumsum(p)
        u = orc.uniform01(seed, sweep_idx, int(n))
        lo, hi = sorted((int(got[n]), int(want[n])))
        assert abs(cdf[lo] - u) < 1e-5 or p[lo + 1:hi + 1].sum() < 1e-5, (n, lo, hi, cdf[lo], u)


@pytest.fixture(params=["rowwise", "transposed"])
def nich1_kernel(request, monkeypatch):
    """the single-nich sweep has two kernels (k_sweep_nich1: a wave scans one row at a time; k_sweep_nich1_t: lane sums
    through LDS, one lane finishes one row; the launcher picks by size): run the test through each"""
    monkeypatch.setenv("MSC_SWEEP_NICH1", {"rowwise": "1", "transposed": "2"}[request.param])
    return request.param


@pytest.mark.parametrize("K", [1, 2, 7, 64, 65, 100, 256, 257, 300, 512, 1000, 1024])
def test_sweep_single_nich_feature_matches_oracle(gpu_ctx, K, nich1_kernel):
    got, want, scores, _ = _run(gpu_ctx, [(orc.NICH, 0)], 3000, K, seed=11 + K, sweep_idx=3)
    _check_agreement(got, want, scores, 11 + K, 3, 0.998)


def test_sweep_mixed_features_matches_oracle(gpu_ctx):
    specs = [(orc.BB, 0), (orc.GP, 0), (orc.DD, 9), (orc.NICH, 0), (orc.NICH, 0)]
    got, want, scores, _ = _run(gpu_ctx, specs, 2500, 120, seed=5, sweep_idx=0)
    _check_agreement(got, want, scores, 5, 0, 0.998)


def test_sweep_generic_path_for_wide_tables(gpu_ctx):
    specs = [(orc.BB, 0), (orc.NICH, 0)]
    got, want, scores, _ = _run(gpu_ctx, specs, 1500, 700, seed=9, sweep_idx=1)   # K > 256, 2 features
    _check_agreement(got, want, scores, 9, 1, 0.998)
    # K > 1024 with ~900 empty groups: ~900 CDF steps of mass 2.5e-4 each, so a float dart lands
    # within rounding (few 1e-6) of a step for ~0.5% of rows; _check_agreement verifies each one
    got, want, scores, _ = _run(gpu_ctx, [(orc.NICH, 0)], 800, 1500, seed=10, sweep_idx=1)
    _check_agreement(got, want, scores, 10, 1, 0.99)


def test_sweep_with_niw_feature_takes_the_generic_path(gpu_ctx):
    got, want, scores, _ = _run(gpu_ctx, [(orc.NIW, 6), (orc.NICH, 0)], 1200, 30, seed=17, sweep_idx=2)
    _check_agreement(got, want, scores, 17, 2, 0.995)


@pytest.mark.parametrize("N", [1, 31, 32, 33, 63, 64, 65, 1000, 4097])
def test_sweep_single_nich_row_counts_across_chunk_boundaries(gpu_ctx, N, nich1_kernel):
    got, want, scores, _ = _run(gpu_ctx, [(orc.NICH, 0)], N, 20, seed=300 + N, sweep_idx=1)
    _check_agreement(got, want, scores, 300 + N, 1, 0.99 if N > 100 else 1.0)


def test_sweep_single_nich_with_singletons_and_unassigned_rows(gpu_ctx, nich1_kernel):
    """off the main path: rows that are their group's only member (the group becomes empty: every empty group's prior
    changes for that row) and rows not assigned to any group"""
    import common_amd
    N, K, seed, sweep_idx, alpha = 2500, 70, 41, 2, 1.7
    rng = np.random.default_rng(seed)
    f = make_feature(orc.NICH, N, K, rng)
    z = rng.integers(0, 40, N).astype(np.int32)
    z[rng.choice(N, 20, replace=False)] = np.arange(40, 60)          # twenty singletons; groups 60..69 stay empty
    z[rng.choice(np.nonzero(z < 40)[0], 100, replace=False)] = -1    # not assigned
    F = orc.Family(orc.NICH, f["hp"], 0, "f64")
    ss64 = F.accumulate(K, f["values"], z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of([f]))
    st = common_amd.State(gpu_ctx, [(orc.NICH, 0)], K)
    st.set_hp(0, F.hp)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.accumulate(view, zt)
    st.set_alpha(alpha)
    st.sweep_assign(view, zt, seed=seed, sweep=sweep_idx)
    want, scores = orc.sweep([(F, ss64, f["values"])], K, alpha, z, seed, sweep_idx, "f64", want_scores=True)
    _check_agreement(zt.cpu().numpy(), want, scores, seed, sweep_idx, 0.995)


@pytest.mark.parametrize("N,K", [(3000, 70), (50_000, 256), (20_000, 1000)])
def test_both_single_nich_kernels_draw_the_same_assignments(gpu_ctx, monkeypatch, N, K):
    """the two kernels follow the same CDF order with the same uniforms: masked values, singletons, unassigned rows
    and empty groups included they may differ only where a dart lands within rounding of a step"""
    import common_amd
    rng = np.random.default_rng(N + K)
    f = make_feature(orc.NICH, N, K, rng)
    used = K - 6
    z = rng.integers(0, used - 10, N).astype(np.int32)
    z[rng.choice(N, 10, replace=False)] = np.arange(used - 10, used)  # singletons
    z[rng.choice(np.nonzero(z < used - 10)[0], N // 20, replace=False)] = -1
    mask = rng.random(N) < 0.15
    mask[np.nonzero(z >= used - 10)[0][:4]] = True                    # masked singletons too
    rec = np.ma.masked_array(np.zeros(N, dtype=[("f0", np.float32)]), mask=[(bool(m),) for m in mask])
    rec.data["f0"] = f["values"]
    view = common_amd.DataView.from_recarray(gpu_ctx, rec)
    st = common_amd.State(gpu_ctx, [(orc.NICH, 0)], K)
    st.set_hp(0, orc.Family(orc.NICH, f["hp"], 0, "f64").hp)
    z0 = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.accumulate(view, z0)
    st.set_alpha(0.9)
    picks = {}
    for name, pin in (("rowwise", "1"), ("transposed", "2")):
        monkeypatch.setenv("MSC_SWEEP_NICH1", pin)
        zt = z0.clone()
        st.sweep_assign(view, zt, seed=5, sweep=9)
        picks[name] = zt.cpu().numpy()
    a, b = picks["rowwise"], picks["transposed"]
    assert (a != z).mean() > 0.05                                     # (something was drawn)
    assert (a == b).mean() >= 0.9995, (a == b).mean()
    assert np.all(np.abs(a - b)[a != b] <= 1) or (a != b).sum() <= 3  # a step to the neighbouring group, if any


def test_sweep_is_a_function_of_seed_sweep_and_global_row(gpu_ctx, nich1_kernel):
    a, _, _, _ = _run(gpu_ctx, [(orc.NICH, 0)], 2000, 64, seed=21, sweep_idx=4)
    b, _, _, _ = _run(gpu_ctx, [(orc.NICH, 0)], 2000, 64, seed=21, sweep_idx=4)
    c, _, _, _ = _run(gpu_ctx, [(orc.NICH, 0)], 2000, 64, seed=21, sweep_idx=5)
    assert np.array_equal(a, b)
    assert (a != c).mean() > 0.01


def test_sharded_rows_draw_the_same_assignments_as_the_whole(gpu_ctx):
    """rows [lo, hi) swept as a shard with row_id0 = lo reproduce the unsharded sweep"""
    import common_amd
    rng = np.random.default_rng(2)
    N, K = 4000, 100
    feats = [make_feature(orc.NICH, N, K, rng), make_feature(orc.BB, N, K, rng)]
    z = rng.integers(0, K, N).astype(np.int32)
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(orc.NICH, 0), (orc.BB, 0)], K)
    load_state(st, fs)
    st.set_group_counts(np.bincount(z, minlength=K).astype(np.uint32))
    whole = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.sweep_assign(view, whole, seed=8, sweep=2)
    parts = torch.from_numpy(z).to(gpu_ctx.torch_device)
    for lo, n in (common_amd.dist.shard_rows(N, 3, r) for r in range(3)):
        zs = parts[lo:lo + n].contiguous()
        st.sweep_assign(view, zs, seed=8, sweep=2, row0=lo, nrows=n, row_id0=lo)
        parts[lo:lo + n] = zs
    assert torch.equal(whole, parts)


def test_sweep_then_rebuild_tables_gives_suffstats_of_new_assignment(gpu_ctx):
    import common_amd
    rng = np.random.default_rng(14)
    N, K = 5000, 33
    feats = [make_feature(orc.NICH, N, K, rng), make_feature(orc.GP, N, K, rng)]
    z = rng.integers(0, K, N).astype(np.int32)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(orc.NICH, 0), (orc.GP, 0)], K)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    drv = common_amd.dist.ShardedSweep(st, view, zt, first_global_row=0)
    drv.rebuild_tables()
    for i in range(3):
        drv.sweep(seed=1, sweep_index=i)
    znew = zt.cpu().numpy()
    assert (znew != z).mean() > 0.3          # the chain moved
    assert np.array_equal(st.get_group_counts(), np.bincount(znew, minlength=K))
    for i, f in enumerate(feats):
        F = orc.Family(f["family"], f["hp"], f["dim"], "f64")
        want = F.accumulate(K, f["values"], znew)
        rec = st.get_ss(i)
        for name in rec.dtype.names:
            a, b = rec[name].astype(np.float64), np.asarray(want[name], np.float64)
            if np.issubdtype(rec.dtype[name].base, np.integer):
                assert np.array_equal(a, b)
            else:
                assert np.all(np.abs(a - b) <= 1e-6 * np.maximum(1, np.abs(b)))


def test_sweep_draws_follow_the_softmax_of_the_scores(gpu_ctx, nich1_kernel):
    """statistical check: one row repeated, many independent uniforms -> empirical = softmax"""
    import common_amd
    rng = np.random.default_rng(3)
    K, N = 6, 60000
    x = np.full(N, 0.3, dtype=np.float32)
    feats = [dict(family=orc.NICH, dim=0, hp=dict(mu=0., kappa=1., sigmasq=1., nu=1.), values=x, np_dtype=np.float32)]
    base = [make_feature(orc.NICH, 600, K, rng)]
    zb = rng.integers(0, K, 600).astype(np.int32)
    fs = state_from_assignment(base, K, zb)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(orc.NICH, 0)], K)
    load_state(st, fs)
    counts = np.bincount(zb, minlength=K).astype(np.uint32)
    st.set_group_counts(counts)
    zt = torch.full((N,), -1, dtype=torch.int32, device=gpu_ctx.torch_device)   # unassigned rows
    st.sweep_assign(view, zt, seed=99, sweep=0)
    emp = np.bincount(zt.cpu().numpy(), minlength=K) / N
    F, ss64, _ = fs[0]
    s = F.score_matrix(ss64, x[:1])[0] + np.log(counts)
    p = np.exp(s - s.max())
    p /= p.sum()
    assert np.abs(emp - p).max() < 4 * np.sqrt(0.25 / N) + 1e-3, (emp, p)


def test_sweep_rows_far_below_the_score_bound_take_the_exact_maximum(gpu_ctx, nich1_kernel):
    """k_sweep_nich1 normalises with a per-wave upper bound of the scores and falls back to the row's exact maximum
    when that underflows: rows thousands of bits below the bound (outliers), next to ordinary rows in the same wave"""
    import common_amd
    N, K, seed, sweep_idx = 3000, 40, 17, 4
    rng = np.random.default_rng(seed)
    f = make_feature(orc.NICH, N, K, rng)
    far = rng.choice(N, 400, replace=False)
    f["values"][far] = (rng.choice([-1.0, 1.0], 400) * 10.0 ** rng.uniform(3, 7, 400)).astype(np.float32)
    z = rng.integers(0, K - 2, N).astype(np.int32)
    z[far[:5]] = K - 3                                     # a small group made of outliers only
    fs = state_from_assignment([f], K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of([f]))
    st = common_amd.State(gpu_ctx, [(orc.NICH, 0)], K)
    load_state(st, fs)
    st.set_group_counts(np.bincount(z, minlength=K).astype(np.uint32))
    st.set_alpha(2.5)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.sweep_assign(view, zt, seed=seed, sweep=sweep_idx)
    want, scores = orc.sweep([(fs[0][0], fs[0][1], f["values"])], K, 2.5, z, seed, sweep_idx, "f64", want_scores=True)
    _check_agreement(zt.cpu().numpy(), want, scores, seed, sweep_idx, 0.995)


@pytest.mark.parametrize("N,K", [(600_000, 16), (300_000, 20), (140_000, 50)])
def test_narrow_tiling_at_the_sizes_that_take_eight_steps_per_wave(gpu_ctx, N, K):
    """K <= 64 runs k_narrow (4 / 8 / 16 lanes per row); from ~32k wave steps on a wave carries eight steps at a time.
    Scores (plain, leave-one-out + prior) on sampled rows and the whole sweep against the oracle."""
    import common_amd
    from tests.gpu_helpers import crp_prior_matrix, oracle_scores, rel_err, TOL
    rng = np.random.default_rng(K)
    specs = [(orc.BB, 0), (orc.NICH, 0), (orc.GP, 0), (orc.DD, 6)]
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z = rng.integers(0, K - 1, N).astype(np.int32)            # the last group stays empty
    z[1234] = -1
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    load_state(st, fs)
    counts = np.bincount(z[z >= 0], minlength=K)
    st.set_group_counts(counts.astype(np.uint32))
    st.set_alpha(1.7)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    rows = np.concatenate([[0, 1234, N - 1], rng.choice(N, 400, replace=False)])
    rt = torch.from_numpy(rows).to(gpu_ctx.torch_device)
    plain = st.score_value(view)
    assert rel_err(plain[rt].cpu().numpy(), oracle_scores(feats, fs, rows=rows)).max() <= TOL
    loo = st.score_value(view, z=zt, crp_prior=True)
    prior = crp_prior_matrix(counts, 1.7, z[rows])
    want = oracle_scores(feats, fs, z=z, rows=rows) + prior
    # The north-star gate is on the float log-score, i.e. on the likelihood part: leave-one-out WITHOUT the prior at
    # the standard gate.
    like_want = oracle_scores(feats, fs, z=z, rows=rows)
    like = st.score_value(view, z=zt)
    assert rel_err(like[rt].cpu().numpy(), like_want).max() <= TOL
    # With the prior: groups of ~40k rows have log(count) = 10.6 and likelihoods near -10, so the sum is O(1) while
    # the likelihood, a float, is only known to ulp(10) = 9.5e-7 (measured budget at (600000, 16): per-feature float
    # scores 1.2e-6 abs, their float accumulation 1.4e-6 more; profiles/r02_crp_error_budget.txt).  The prior must add
    # NOTHING to that: the kernels carry it as a (hi, lo) pair -- exact to 1e-14 -- and add hi after the last feature,
    # so the error of likelihood + prior is the error of the likelihood, gated relative to the likelihood ...
    got = loo[rt].cpu().numpy().astype(np.float64)
    err = np.abs(got - want)
    assert (err / np.maximum(1.0, np.maximum(np.abs(want), np.abs(like_want)))).max() <= TOL
    # ... and where the row's own group is concerned (one double evaluation per row, k_loo_own) relative to the result
    own = np.zeros(want.shape, dtype=bool)
    for r, n in enumerate(rows):
        if z[n] >= 0:
            own[r, z[n]] = True
    assert (err / np.maximum(1.0, np.abs(want)))[own].max() <= TOL
    st.sweep_assign(view, zt, seed=5, sweep=1)
    got = zt.cpu().numpy()
    feats_o = [(F, ss64, f["values"]) for f, (F, ss64, _) in zip(feats, fs)]
    want_z, scores = orc.sweep(feats_o, K, 1.7, z, 5, 1, "f64", want_scores=True)
    _check_agreement(got, want_z, scores, 5, 1, 0.998)


@pytest.mark.parametrize("dim,K", [(1, 5), (2, 40), (3, 64), (8, 33), (5, 1), (2, 100), (4, 200), (5, 128), (3, 256)])
def test_sweep_single_small_niw_feature_is_one_fused_kernel(gpu_ctx, dim, K):
    """one niw feature, dim <= 8, K <= 64 (a Gaussian mixture on low-dimensional vectors): k_sweep_niw1"""
    got, want, scores, _ = _run(gpu_ctx, [(orc.NIW, dim)], 3000, K, seed=100 * dim + K, sweep_idx=2, empty=min(2, K - 1))
    _check_agreement(got, want, scores, 100 * dim + K, 2, 0.997)


def test_sweep_small_niw_with_masked_vectors_and_singletons(gpu_ctx):
    import common_amd
    N, K, dim, seed = 2500, 20, 3, 31
    rng = np.random.default_rng(seed)
    f = make_feature(orc.NIW, N, K, rng, dim)
    z = rng.integers(0, K - 3, N).astype(np.int32)
    z[7] = K - 3                                            # a group of one: removing the row empties it
    z[11] = -1
    mask = np.zeros(N, dtype=[("f0", np.bool_, (dim,))])
    hide = rng.random(N) < 0.2
    mask["f0"][hide, rng.integers(0, dim, hide.sum())] = True
    keep = ~hide
    fs = state_from_assignment([dict(f, values=f["values"][keep])], K, z[keep])
    view = common_amd.DataView.from_recarray(gpu_ctx, np.ma.masked_array(recarray_of([f]), mask=mask))
    st = common_amd.State(gpu_ctx, [(orc.NIW, dim)], K)
    load_state(st, fs)
    counts = np.bincount(z[z >= 0], minlength=K)
    st.set_group_counts(counts.astype(np.uint32))
    st.set_alpha(0.8)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.sweep_assign(view, zt, seed=seed, sweep=0)
    got = zt.cpu().numpy()
    # the oracle's view: masked rows score with the prior alone
    F, ss64, _ = fs[0]
    from tests.gpu_helpers import crp_prior_matrix
    like = np.zeros((N, K))
    zk = np.where(hide, -1, z)                               # a masked row is not in its group's suff-stats: nothing to leave out
    like[keep] = F.score_matrix(ss64, f["values"][keep], zk[keep])
    total = like + crp_prior_matrix(counts, 0.8, z)
    agree = 0
    for n in range(N):
        p = orc.scores_to_probs(total[n])
        pick = orc.sample_discrete(p, orc.uniform01(seed, 0, n))
        if pick == got[n]:
            agree += 1
        else:
            cdf = np.cumsum(p)
            lo, hi = sorted((int(got[n]), int(pick)))
            assert abs(cdf[lo] - orc.uniform01(seed, 0, n)) < 1e-5 or p[lo + 1:hi + 1].sum() < 1e-5, n
    assert agree >= 0.997 * N


@pytest.mark.parametrize("seed", range(24))
def test_single_nich_kernels_agree_on_random_shapes(gpu_ctx, monkeypatch, seed):
    """differential fuzz of the two single-nich sweep kernels: random row / group counts (every G of the templates,
    chunk remainders), concentration, share of empty groups, singletons, unassigned and masked rows, shard offsets"""
    import common_amd
    rng = np.random.default_rng(1000 + seed)
    K = int(rng.choice([1, 2, 3, 17, 64, 65, 100, 128, 129, 255, 256, 257, 400, 512, 513, 900, 1024]))
    N = int(rng.choice([1, 2, 31, 33, 63, 64, 65, 127, 500, 1000, 4097, 20000]))
    f = make_feature(orc.NICH, N, K, rng)
    used = max(1, K - int(rng.integers(0, max(1, K // 3) + 1)))
    z = rng.integers(0, used, N).astype(np.int32)
    if N > 40:
        z[rng.choice(N, N // 10, replace=False)] = -1
    mask = rng.random(N) < rng.choice([0.0, 0.05, 0.5])
    rec = np.ma.masked_array(np.zeros(N, dtype=[("f0", np.float32)]), mask=[(bool(m),) for m in mask])
    rec.data["f0"] = f["values"]
    view = common_amd.DataView.from_recarray(gpu_ctx, rec)
    st = common_amd.State(gpu_ctx, [(orc.NICH, 0)], K)
    st.set_hp(0, orc.Family(orc.NICH, f["hp"], 0, "f64").hp)
    z0 = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.accumulate(view, z0)
    st.set_alpha(float(rng.choice([0.1, 1.0, 7.5])))
    row0 = int(rng.integers(0, max(1, N // 3)))
    nrows = N - row0
    picks = {}
    for name, pin in (("rowwise", "1"), ("transposed", "2")):
        monkeypatch.setenv("MSC_SWEEP_NICH1", pin)
        zt = z0[row0:].clone()
        st.sweep_assign(view, zt, seed=seed, sweep=3, row0=row0, nrows=nrows, row_id0=10 * row0)
        picks[name] = zt.cpu().numpy()
    a, b = picks["rowwise"], picks["transposed"]
    assert a.min() >= 0 and a.max() < K and b.min() >= 0 and b.max() < K
    differ = int((a != b).sum())
    assert differ <= max(1, nrows // 2000), (differ, nrows, K)      # a dart within rounding of a CDF step, if any


@pytest.mark.parametrize("K,N", [(1025, 3000), (2048, 5000), (3000, 260), (4096, 1), (5000, 129)])
def test_sweep_single_nich_beyond_1024_groups_matches_oracle(gpu_ctx, K, N):
    """k_sweep_nich1_rows (lane <-> row, the groups as scalar operands): against the oracle sweep, with singletons,
    unassigned rows, empty groups and a few far outliers (rows that take the exact-maximum pass)"""
    import common_amd
    seed, sweep_idx, alpha = 100 + K, 2, 1.9
    rng = np.random.default_rng(seed)
    f = make_feature(orc.NICH, N, K, rng)
    used = K - K // 8
    z = rng.integers(0, used - 30, N).astype(np.int32)
    if N > 100:
        z[rng.choice(N, 20, replace=False)] = np.arange(used - 30, used - 10)      # twenty singletons
        z[rng.choice(np.nonzero(z < used - 30)[0], N // 20, replace=False)] = -1
        far = rng.choice(N, 10, replace=False)
        f["values"][far] = (rng.choice([-1.0, 1.0], 10) * 10.0 ** rng.uniform(4, 7, 10)).astype(np.float32)
    F = orc.Family(orc.NICH, f["hp"], 0, "f64")
    ss64 = F.accumulate(K, f["values"], z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of([f]))
    st = common_amd.State(gpu_ctx, [(orc.NICH, 0)], K)
    st.set_hp(0, F.hp)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.accumulate(view, zt)
    st.set_alpha(alpha)
    st.sweep_assign(view, zt, seed=seed, sweep=sweep_idx)
    got = zt.cpu().numpy()
    assert got.min() >= 0 and got.max() < K
    want, scores = orc.sweep([(F, ss64, f["values"])], K, alpha, z, seed, sweep_idx, "f64", want_scores=True)
    _check_agreement(got, want, scores, seed, sweep_idx, 0.99 if N > 200 else 0.97)


def test_sweep_beyond_1024_groups_masked_rows_follow_the_prior_and_shards_agree(gpu_ctx):
    import common_amd
    K, N = 1500, 40000
    rng = np.random.default_rng(9)
    f = make_feature(orc.NICH, N, K, rng)
    z = rng.integers(0, 60, N).astype(np.int32)                    # 60 groups in use, 1440 empty
    mask = np.zeros(N, dtype=bool)
    mask[:20000] = True                                             # the first half: prior only
    rec = np.ma.masked_array(np.zeros(N, dtype=[("f0", np.float32)]), mask=[(bool(m),) for m in mask])
    rec.data["f0"] = f["values"]
    view = common_amd.DataView.from_recarray(gpu_ctx, rec)
    st = common_amd.State(gpu_ctx, [(orc.NICH, 0)], K)
    st.set_hp(0, orc.Family(orc.NICH, f["hp"], 0, "f64").hp)
    z0 = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.accumulate(view, z0)
    st.set_alpha(3.0)
    whole = torch.full((N,), -1, dtype=torch.int32, device=gpu_ctx.torch_device)   # unassigned: no leave-one-out
    st.sweep_assign(view, whole, seed=6, sweep=0)
    got = whole.cpu().numpy()
    counts = st.get_group_counts().astype(np.float64)
    pc = np.where(counts > 0, counts, 3.0 / (counts == 0).sum())
    emp_used = np.bincount(got[:20000], minlength=K)[:60] / 20000.0
    assert np.abs(emp_used - (pc / pc.sum())[:60]).max() < 0.012           # masked rows: the CRP prior
    assert abs((got[:20000] >= 60).mean() - 3.0 / pc.sum()) < 0.01        # ... alpha's share spread over the empty groups
    parts = torch.full((N,), -1, dtype=torch.int32, device=gpu_ctx.torch_device)
    for lo, n in (common_amd.dist.shard_rows(N, 3, r) for r in range(3)):
        zs = parts[lo:lo + n].contiguous()
        st.sweep_assign(view, zs, seed=6, sweep=0, row0=lo, nrows=n, row_id0=lo)
        parts[lo:lo + n] = zs
    assert torch.equal(whole, parts)


def test_outliers_in_tiny_groups_keep_a_finite_leave_one_out_score(gpu_ctx, nich1_kernel):
    """a row 1e6 away in a group of three: count_times_variance is ~1e12 as a float (ulp 1e5), so the variance rebuilt
    without the row can come out below zero -- it is clamped at zero instead of turning the score into NaN (which sent
    such rows to group 0)"""
    import common_amd
    K, N, seed, sweep_idx, alpha = 1000, 3000, 1125, 2, 1.9
    rng = np.random.default_rng(seed)
    f = make_feature(orc.NICH, N, K, rng)
    z = rng.integers(0, 850, N).astype(np.int32)
    far = rng.choice(N, 30, replace=False)
    f["values"][far] = (rng.choice([-1.0, 1.0], 30) * 10.0 ** rng.uniform(4, 7, 30)).astype(np.float32)
    F = orc.Family(orc.NICH, f["hp"], 0, "f64")
    ss64 = F.accumulate(K, f["values"], z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of([f]))
    st = common_amd.State(gpu_ctx, [(orc.NICH, 0)], K)
    st.set_hp(0, F.hp)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.accumulate(view, zt)
    st.set_alpha(alpha)
    loo = st.score_value(view, z=zt).cpu().numpy()
    assert np.isfinite(loo[:, :850]).all()
    st.sweep_assign(view, zt, seed=seed, sweep=sweep_idx)
    want, scores = orc.sweep([(F, ss64, f["values"])], K, alpha, z, seed, sweep_idx, "f64", want_scores=True)
    _check_agreement(zt.cpu().numpy(), want, scores, seed, sweep_idx, 0.995)


def test_large_and_small_mixed_sweeps_draw_the_same_assignments(gpu_ctx):
    """40k rows take the kernel whose waves split the lookup and the nich phase (k_sweep_tile_roles), a third of them
    the ones that run the phases in turn: the same scores bit for bit, hence the same draws (what lets a shard of any
    size reproduce the unsharded sweep)"""
    import common_amd
    rng = np.random.default_rng(77)
    N, K = 40_000, 120
    specs = [(orc.BB, 0), (orc.NICH, 0), (orc.GP, 0), (orc.DD, 6), (orc.NICH, 0), (orc.BB, 0)]
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z = rng.integers(0, K - 5, N).astype(np.int32)
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, specs, K)
    load_state(st, fs)
    st.set_group_counts(np.bincount(z, minlength=K).astype(np.uint32))
    st.set_alpha(1.1)
    whole = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.sweep_assign(view, whole, seed=12, sweep=4)
    parts = torch.from_numpy(z).to(gpu_ctx.torch_device)
    for lo, n in (common_amd.dist.shard_rows(N, 3, r) for r in range(3)):
        zs = parts[lo:lo + n].contiguous()
        st.sweep_assign(view, zs, seed=12, sweep=4, row0=lo, nrows=n, row_id0=lo)
        parts[lo:lo + n] = zs
    assert torch.equal(whole, parts)
    assert (whole.cpu().numpy() != z).mean() > 0.05


TAIL_PLANS = {
    "mixed": [(orc.BB, 0), (orc.GP, 0), (orc.NICH, 0), (orc.DD, 9), (orc.NICH, 0), (orc.BB, 0), (orc.BNB, 0)],
    "lookups_only": [(orc.BB, 0), (orc.GP, 0), (orc.DD, 9), (orc.BB, 0), (orc.BNB, 0)],      # (k_sweep_lookups<false, 1 | 2>)
    "nich_only": [(orc.NICH, 0)] * 3,                                                        # (k_sweep_nich_pack<false, false, 1 | 2>)
    "mostly_nich": [(orc.DD, 9)] + [(orc.NICH, 0)] * 4,                                      # (k_sweep_nich_pack<false, true, 1 | 2>)
}


@pytest.mark.parametrize("plan", sorted(TAIL_PLANS))
@pytest.mark.parametrize("K,empty", [(257, 0), (300, 3), (320, 40), (321, 0), (350, 5), (384, 70)])
def test_fused_sweep_with_a_narrow_tail_matches_oracle_and_its_shards(gpu_ctx, K, empty, plan, monkeypatch):
    """256 < K <= 384 on a state of lookup + nich features: the groups beyond the first tile are scored by the narrow kernel
    (k_score_tail_rows: 64 or 128 floats per row, leave-one-out value and prior included) and the role-split sweep kernel
    draws over tile + tail (k_sweep_tile_roles<1 | 2>, sample_tile_and_tail) -- for every row range of a view.  Against the oracle's sweep
    (every disagreeing draw on a CDF step), rows whose own group lies in the tail and empty tail groups included; and the
    same sweep in three shards draws the same assignments."""
    import common_amd
    monkeypatch.setenv("MSC_TAIL_MIN_ROWS", "1")            # (3000 and 20k rows here; the library's own mark: ~770 rows a tail group)
    specs = TAIL_PLANS[plan]
    got, want, scores, z = _run(gpu_ctx, specs, 3000, K, seed=40 + K, sweep_idx=2, alpha=0.8, empty=empty)
    assert (z >= 256).any() or K - empty <= 256
    _check_agreement(got, want, scores, 40 + K, 2, 0.995)
    assert (got >= 256).any()                                   # draws do land in the tail
    # shards: the same state, 20k rows, whole against three parts
    rng = np.random.default_rng(K)
    N = 20_000
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    zz = rng.integers(0, K - empty, N).astype(np.int32)
    fs = state_from_assignment(feats, K, zz)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    load_state(st, fs)
    st.set_group_counts(np.bincount(zz, minlength=K).astype(np.uint32))
    st.set_alpha(0.8)
    whole = torch.from_numpy(zz).to(gpu_ctx.torch_device)
    st.sweep_assign(view, whole, seed=9, sweep=1)
    parts = torch.from_numpy(zz).to(gpu_ctx.torch_device)
    for lo, n in (common_amd.dist.shard_rows(N, 3, r) for r in range(3)):
        zs = parts[lo:lo + n].contiguous()
        st.sweep_assign(view, zs, seed=9, sweep=1, row0=lo, nrows=n, row_id0=lo)
        parts[lo:lo + n] = zs
    assert torch.equal(whole, parts)
    assert (whole.cpu().numpy() != zz).mean() > 0.05


ROWS_PLANS = {
    # (tables too large for the K <= 64 narrow tiling's LDS: these states are the lane <-> row kernel's)
    "mixed": [(orc.DD, 60), (orc.BB, 0), (orc.GP, 0), (orc.NICH, 0), (orc.DD, 64), (orc.NICH, 0), (orc.DD, 50), (orc.DD, 61),
              (orc.BBNC, 0), (orc.DD, 33)],
    "lookups_only": [(orc.DD, 64)] * 5 + [(orc.GP, 0)],
    "many_stages": [(orc.DD, 60)] * 9 + [(orc.NICH, 0)] * 9,
}


@pytest.mark.parametrize("plan", sorted(ROWS_PLANS))
@pytest.mark.parametrize("K,empty", [(2, 0), (5, 1), (17, 3), (33, 0), (48, 9), (64, 2), (65, 0), (100, 20), (128, 1)])
def test_sweep_on_the_lane_row_kernel_matches_oracle_and_its_shards(gpu_ctx, plan, K, empty):
    """A state of at most 128 groups on a plan of lookup + plain nich features sweeps on the lane <-> row kernel whatever
    the row count: up to 64 groups a lane scores its row against every group and draws by itself (k_score_tail_rows<.,
    false, true>: 16 / 32 / 48 / 64 sums a lane), beyond that the scores go through 128 floats per row to the row sampler.
    Against the oracle's sweep (every disagreeing draw on a CDF step), empty groups on offer; unassigned rows draw too; the
    same sweep in three shards draws the same assignments; and the sweep step's tables equal an accumulate over the result."""
    _rows_sweep_case(gpu_ctx, ROWS_PLANS[plan], K, empty)


@pytest.mark.parametrize("K,empty", [(100, 7), (128, 0), (200, 30), (256, 1)])
def test_sweep_of_a_nich_only_state_matches_oracle_and_its_shards(gpu_ctx, K, empty):
    """a state of plain nich features alone (a mixture of independent Gaussians per dimension) with rows enough: the fused
    step on k_sweep_nich_pack -- every wave a nich wave, nothing through LDS; up to 128 groups in PAIR mode -- against the
    oracle's sweep, and three shards of the view draw what the whole draws"""
    _pair_sweep_case(gpu_ctx, [(orc.NICH, 0)] * 6, K, empty)


@pytest.mark.parametrize("K,empty", [(100, 7), (128, 0), (200, 30), (256, 1)])
def test_sweep_of_a_lookups_only_state_matches_oracle_and_its_shards(gpu_ctx, K, empty):
    """a state of staged lookup features alone (a mixture of categoricals and counts) with rows enough: the fused step on
    k_sweep_lookups -- every wave a lookup wave of 16 sums; up to 128 groups in PAIR mode -- against the oracle's sweep, and
    three shards of the view draw what the whole draws"""
    _pair_sweep_case(gpu_ctx, [(orc.BB, 0)] * 5 + [(orc.DD, 9), (orc.GP, 0), (orc.BB, 0), (orc.DD, 70)], K, empty)


@pytest.mark.parametrize("K,empty", [(65, 0), (100, 20), (127, 3), (128, 1)])
def test_sweep_in_pair_mode_matches_oracle_and_its_shards(gpu_ctx, K, empty):
    """65 .. 128 groups and rows enough for the role-split kernels (40k; no kernel forced): k_sweep_tile_roles<0, PAIR> -- a
    lane carries two groups, a float4 of sums two rows, the draw runs over 64 x 2 entries.  Against the oracle's sweep (every
    disagreeing draw on a CDF step), empty groups on offer, an unassigned row; three shards of the view draw what the whole
    draws (the mode follows the view's rows, not the call's)."""
    _pair_sweep_case(gpu_ctx, ROWS_PLANS["mixed"], K, empty)


def _pair_sweep_case(gpu_ctx, specs, K, empty):
    import common_amd
    N = 40_000
    got, want, scores, z = _run(gpu_ctx, specs, N, K, seed=170 + K, sweep_idx=1, alpha=1.1, empty=empty)
    _check_agreement(got, want, scores, 170 + K, 1, 0.995)
    rng = np.random.default_rng(K)
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    zz = rng.integers(0, K - empty, N).astype(np.int32)
    zz[5] = -1
    fs = state_from_assignment(feats, K, zz)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    load_state(st, fs)
    st.set_group_counts(np.bincount(zz[zz >= 0], minlength=K).astype(np.uint32))
    st.set_alpha(1.1)
    whole = torch.from_numpy(zz).to(gpu_ctx.torch_device)
    st.sweep_assign(view, whole, seed=9, sweep=1)
    parts = torch.from_numpy(zz).to(gpu_ctx.torch_device)
    for lo, n in (common_amd.dist.shard_rows(N, 3, r) for r in range(3)):
        zs = parts[lo:lo + n].contiguous()
        st.sweep_assign(view, zs, seed=9, sweep=1, row0=lo, nrows=n, row_id0=lo)
        parts[lo:lo + n] = zs
    assert torch.equal(whole, parts)
    w = whole.cpu().numpy()
    assert w.min() >= 0 and w.max() < K and (w != zz).mean() > 0.05


def test_sweep_on_the_lane_row_kernel_sixteen_sums(gpu_ctx):
    """(12 groups and more table rows than the narrow tiling's LDS holds even then: the kernel's narrowest instantiation)"""
    _rows_sweep_case(gpu_ctx, [(orc.DD, 64)] * 17 + [(orc.NICH, 0)], 12, 2)


def _rows_sweep_case(gpu_ctx, specs, K, empty):
    import common_amd
    os.environ["MSC_TAIL_MIN_ROWS"] = "1"                   # (a few thousand rows here; the library's own mark is ~1500 rows a group)
    try:
        _rows_sweep_body(gpu_ctx, specs, K, empty)
    finally:
        os.environ.pop("MSC_TAIL_MIN_ROWS", None)


def _rows_sweep_body(gpu_ctx, specs, K, empty):
    import common_amd
    got, want, scores, z = _run(gpu_ctx, specs, 2000, K, seed=70 + K, sweep_idx=1, alpha=1.1, empty=empty)
    _check_agreement(got, want, scores, 70 + K, 1, 0.995)
    rng = np.random.default_rng(K)
    N = 9_000
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    zz = rng.integers(0, max(1, K - empty), N).astype(np.int32)
    zz[5] = -1
    fs = state_from_assignment(feats, K, zz)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    load_state(st, fs)
    st.set_group_counts(np.bincount(zz[zz >= 0], minlength=K).astype(np.uint32))
    st.set_alpha(1.1)
    whole = torch.from_numpy(zz).to(gpu_ctx.torch_device)
    st.sweep_assign(view, whole, seed=9, sweep=1)
    parts = torch.from_numpy(zz).to(gpu_ctx.torch_device)
    for lo, n in (common_amd.dist.shard_rows(N, 3, r) for r in range(3)):
        zs = parts[lo:lo + n].contiguous()
        st.sweep_assign(view, zs, seed=9, sweep=1, row0=lo, nrows=n, row_id0=lo)
        parts[lo:lo + n] = zs
    assert torch.equal(whole, parts)
    w = whole.cpu().numpy()
    assert w.min() >= 0 and w.max() < K and (K == 1 or (w != zz).mean() > 0.05)
    # a whole step (draw, then the tables rebuilt from the new assignment) against accumulate over the same assignment
    zs = torch.from_numpy(np.where(zz < 0, 0, zz)).to(gpu_ctx.torch_device)
    st.accumulate(view, zs)
    st.sweep_step(view, zs, seed=4, sweep=0)
    ref = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    ref.accumulate(view, zs)
    assert np.array_equal(st.get_group_counts(), ref.get_group_counts())
    assert np.array_equal(st.get_group_counts(), np.bincount(zs.cpu().numpy(), minlength=K))
    for f in range(len(specs)):
        a, b = st.get_ss(f), ref.get_ss(f)
        for name in a.dtype.names:
            if name != "p":                                     # (bbnc's group parameter is the state's own, not a sum over rows)
                assert np.array_equal(a[name], b[name]), (f, name)
