"""GPU parity for the two remaining component models of microscopes/models.pyx:185-290:
BetaNegativeBinomial (distributions.hpp:29-36; a gp-like exact count table) and the in-tree
Dirichlet-Multinomial (src/models/dm.cpp:10-97; dim + 1 count lookups per row)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.conftest import load_golden
from tests.gpu_helpers import (TOL, audit, crp_prior_matrix, load_state, make_feature, oracle_scores, recarray_of, rel_err,
                               state_from_assignment)

pytestmark = pytest.mark.gpu


def _setup(gpu_ctx, specs, N, K, seed, hp=None):
    import common_amd
    rng = np.random.default_rng(seed)
    feats = [make_feature(f, N, K, rng, d, hp=(hp or {}).get(i)) for i, (f, d) in enumerate(specs)]
    z = rng.integers(0, K, N).astype(np.int32)
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    return feats, z, fs, view, st


@pytest.mark.parametrize("specs,K", [
    ([(orc.BNB, 0)], 20),
    ([(orc.DM, 5)], 33),
    ([(orc.DM, 2)], 300),                       # two k-tiles
    ([(orc.DM, 40)], 17),                       # more stages than fit the register budget of a naive unroll
    ([(orc.BB, 0), (orc.DM, 7), (orc.NICH, 0), (orc.BNB, 0), (orc.DM, 3), (orc.GP, 0)], 64),
])
def test_score_value_plain_and_leave_one_out(gpu_ctx, specs, K):
    N = 2000
    feats, z, fs, view, st = _setup(gpu_ctx, specs, N, K, seed=len(specs) * 100 + K)
    load_state(st, fs)
    st.set_group_counts(np.bincount(z, minlength=K).astype(np.uint32))
    got = st.score_value(view).cpu().numpy()
    assert rel_err(got, oracle_scores(feats, fs)).max() <= TOL
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    loo = st.score_value(view, z=zt).cpu().numpy()
    assert rel_err(loo, oracle_scores(feats, fs, z=z)).max() <= TOL
    st.set_alpha(0.7)
    both = st.score_value(view, z=zt, crp_prior=True).cpu().numpy()
    want = oracle_scores(feats, fs, z=z) + crp_prior_matrix(np.bincount(z, minlength=K), 0.7, z)
    assert rel_err(both, want).max() <= TOL


@pytest.mark.parametrize("family,dim", [(orc.BNB, 0), (orc.DM, 6), (orc.DM, 128)])
def test_accumulate_commit_and_score_data(gpu_ctx, family, dim):
    N, K = 3000, 19
    feats, z, fs, view, st = _setup(gpu_ctx, [(family, dim)], N, K, seed=dim + 3)
    st.set_hp(0, fs[0][0].hp)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.accumulate(view, zt)
    rec = st.get_ss(0)
    want = fs[0][2]
    for name in rec.dtype.names:
        if np.issubdtype(rec.dtype[name].base, np.integer):
            assert np.array_equal(rec[name], want[name]), name        # counts: bit-exact
        else:
            assert rel_err(rec[name], fs[0][1][name]).max() <= TOL, name
    assert np.array_equal(st.get_group_counts(), np.bincount(z, minlength=K))
    sd = st.score_data().cpu().numpy()[0]
    assert rel_err(sd, fs[0][0].score_data_all(fs[0][1])).max() <= TOL
    # subtract half of the rows again: the tables must equal those of the other half alone
    half = N // 2
    st.accumulate(view, zt[:half], row0=0, nrows=half, reset=False, subtract=True)
    rest = orc.Family(family, feats[0]["hp"], dim, "f64").accumulate(K, feats[0]["values"][half:], z[half:])
    rec = st.get_ss(0)
    for name in rec.dtype.names:
        if np.issubdtype(rec.dtype[name].base, np.integer):
            assert np.array_equal(rec[name], rest[name]), name
        else:
            # the additive tables are doubles and commit rounds once to float: what is left after the subtraction is
            # within the plain gate of the other half's own statistics, relative to the column's largest entry (a
            # group's ratio / log_prod of all rows is what the double sums were rounded at; round 2: 1e-4)
            audit("bnb_dm.float_field_after_subtract." + name,
                  np.abs(rec[name] - rest[name]).max() / max(1.0, np.abs(rest[name]).max()), TOL)


def test_counts_beyond_the_table_take_the_double_path(gpu_ctx):
    """values >= 1024 (the table cap) are scored by the large-count kernel, plain and leave-one-out"""
    import common_amd
    rng = np.random.default_rng(8)
    N, K = 600, 9
    fb = make_feature(orc.BNB, N, K, rng)
    fb["values"][::7] += rng.integers(1024, 200000, len(fb["values"][::7])).astype(np.uint32)
    fd = make_feature(orc.DM, N, K, rng, 4)
    fd["values"][::5, 2] += rng.integers(1024, 50000, len(fd["values"][::5])).astype(np.int32)   # category and total
    fd["values"][1::11, 0] += 600                                                               # total only
    fd["values"][1::11, 1] += 600
    feats = [fb, fd]
    z = rng.integers(0, K, N).astype(np.int32)
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(orc.BNB, 0), (orc.DM, 4)], K)
    load_state(st, fs)
    got = st.score_value(view).cpu().numpy()
    assert rel_err(got, oracle_scores(feats, fs)).max() <= TOL
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    loo = st.score_value(view, z=zt).cpu().numpy()
    assert rel_err(loo, oracle_scores(feats, fs, z=z)).max() <= TOL


@pytest.mark.parametrize("name,family", [("bnb", orc.BNB), ("dm", orc.DM)])
def test_per_value_api_against_the_scipy_golden_vectors(gpu_ctx, name, family):
    """group::score_value / score_data / add_value / remove_value through msc_value_op_single"""
    import common_amd
    for case in load_golden(name):
        dim = case.get("dim", 0)
        rec = np.zeros(1, dtype=common_amd.ss_dtype(family, dim))
        rows = np.asarray(case["rows"]).reshape((-1,) + ((dim,) if dim else ()))
        for v in rows:
            gpu_ctx.value_op(family, dim, "add", case["hp"], rec, v)
        for k, want in case["ss"].items():
            if np.issubdtype(rec.dtype[k].base, np.integer):
                assert np.array_equal(rec[k][0], np.asarray(want)), (name, k)
            else:
                # a float field after len(rows) per-value updates, each rounding the field once (the reference keeps
                # it in float too, distributions.hpp:38-45 / dm.hpp:86-88): len(rows) half-ulps of the largest value
                # the running sum takes -- the budget of the float STATE, whoever does the arithmetic
                budget = max(1, len(rows)) * 2.0 ** -24
                audit("bnb_dm.per_value.float_field." + name, abs(float(rec[k][0]) - want) / max(1.0, abs(want)) / budget, 1.0)
        got = [gpu_ctx.value_op(family, dim, "score_value", case["hp"], rec, np.asarray(v)) for v in case["probe"]]
        # what the device is answerable for: the double twin evaluated on the record the device itself holds (integers
        # exact; dm's float `ratio` carries the running-sum rounding above and enters its score) -- the plain gate
        F = orc.Family(family, case["hp"], dim, "f64")
        held = orc.widen_ss(family, rec.astype(orc.ss_dtype(family, dim, "f32")), dim)
        tw = [F.score_value(held, 0, np.asarray(v)) for v in case["probe"]]
        audit("bnb_dm.per_value.score_value_vs_twin." + name, rel_err(got, tw).max(), TOL)
        # ... and the scipy closed form on exact data (measured 5e-8, profiles/r03_tolerance_audit.json; round 2: +1e-5)
        audit("bnb_dm.per_value.score_value_vs_scipy." + name, rel_err(got, case["score_value"]).max(), TOL)
        sd = gpu_ctx.value_op(family, dim, "score_data", case["hp"], rec)
        tw_sd = F.score_data(held, 0)
        audit("bnb_dm.per_value.score_data_vs_twin." + name, abs(sd - tw_sd) / max(1.0, abs(tw_sd)), TOL)
        audit("bnb_dm.per_value.score_data_vs_scipy." + name, abs(sd - case["score_data"]) / max(1.0, abs(case["score_data"])), TOL)   # (round 2: 2e-5)
        if len(rows):
            gpu_ctx.value_op(family, dim, "remove", case["hp"], rec, rows[-1])
            gpu_ctx.value_op(family, dim, "add", case["hp"], rec, rows[-1])
            for k, want in case["ss"].items():
                if np.issubdtype(rec.dtype[k].base, np.integer):
                    assert np.array_equal(rec[k][0], np.asarray(want))


def test_sweep_with_bnb_and_dm_features_matches_the_oracle(gpu_ctx):
    from tests.test_gpu_sweep import _check_agreement, _run
    specs = [(orc.BNB, 0), (orc.DM, 6), (orc.NICH, 0)]
    got, want, scores, _ = _run(gpu_ctx, specs, 2500, 90, seed=21, sweep_idx=2)
    _check_agreement(got, want, scores, 21, 2, 0.998)


def test_masked_dm_rows_contribute_nothing(gpu_ctx):
    import common_amd
    rng = np.random.default_rng(4)
    N, K, C = 1500, 11, 5
    feats = [make_feature(orc.DM, N, K, rng, C), make_feature(orc.NICH, N, K, rng)]
    z = rng.integers(0, K, N).astype(np.int32)
    rec = recarray_of(feats)
    mask = np.zeros(N, dtype=[("f0", np.bool_, (C,)), ("f1", np.bool_)])
    hide = rng.random(N) < 0.2
    mask["f0"][hide, rng.integers(0, C, hide.sum())] = True        # one masked element hides the whole vector
    view = common_amd.DataView.from_recarray(gpu_ctx, np.ma.masked_array(rec, mask=mask))
    keep = ~hide
    fs_keep = state_from_assignment([dict(feats[0], values=feats[0]["values"][keep])], K, z[keep])
    fs_all = state_from_assignment([feats[1]], K, z)
    st = common_amd.State(gpu_ctx, [(orc.DM, C), (orc.NICH, 0)], K)
    st.set_hp(0, fs_keep[0][0].hp)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.accumulate(view, zt)
    assert np.array_equal(st.get_ss(0)["counts"], fs_keep[0][2]["counts"])
    load_state(st, [fs_keep[0], fs_all[0]])
    got = st.score_value(view).cpu().numpy()
    dm_part = fs_keep[0][0].score_matrix(fs_keep[0][1], feats[0]["values"])
    dm_part[hide] = 0.0
    want = dm_part + fs_all[0][0].score_matrix(fs_all[0][1], feats[1]["values"])
    assert rel_err(got, want).max() <= TOL


def _small_counts(feat):
    """the same feature with counts small enough that all dim + 1 tables fit the LDS slot (row totals of at most ~10)"""
    return dict(feat, values=(feat["values"] // 5).astype(np.int32))


DM_ROWS_PLANS = {
    "four_dm4": [(orc.DM, 4)] * 4,
    "mixed": [(orc.BB, 0), (orc.DM, 4), (orc.NICH, 0), (orc.DM, 3), (orc.GP, 0), (orc.NICH, 0)],
}


@pytest.mark.parametrize("plan", sorted(DM_ROWS_PLANS))
@pytest.mark.parametrize("K", [5, 32, 50, 64])
def test_dm_states_of_few_groups_score_on_the_lane_row_kernel(gpu_ctx, plan, K, monkeypatch):
    """Round 5: a state of at most 64 groups whose dm features have small counts (tables staged whole) takes the lane <-> row
    kernel (k_score_tail_rows<..., DMF = true>): dim + 1 (hi, lo) lookups per row and feature, the cost following the groups.
    Plain, leave-one-out and leave-one-out + prior against the oracle (the per-feature tolerances add: a sum of D features is
    gated at 1e-6 sum_f max(1, |score_f|), tests/test_gpu_score.py _gate_on_sum); src/models/dm.cpp:39-76."""
    import common_amd
    from tests.test_gpu_score import _gate_on_sum
    monkeypatch.setenv("MSC_TAIL_MIN_ROWS", "16384")       # (the library's own mark grows with the groups: ~1000 rows a group)
    N = 20_000
    rng = np.random.default_rng(K * 7 + len(plan))
    feats = [make_feature(f, N, K, rng, d) for f, d in DM_ROWS_PLANS[plan]]
    feats = [_small_counts(f) if f["family"] == orc.DM else f for f in feats]
    z = rng.integers(0, K, N).astype(np.int32)
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    load_state(st, fs)
    st.set_group_counts(np.bincount(z, minlength=K).astype(np.uint32))
    got = st.score_value(view).cpu().numpy()
    assert gpu_ctx.last_kernel("score").startswith("k_score_tail_rows<") and gpu_ctx.last_kernel("score").endswith("true>")
    rows = np.arange(N)
    audit("dm_lane_row.plain", _gate_on_sum(got, feats, fs, rows), TOL)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    loo = st.score_value(view, z=zt).cpu().numpy()
    audit("dm_lane_row.loo", _gate_on_sum(loo, feats, fs, rows, z=z), TOL)
    st.set_alpha(0.7)
    both = st.score_value(view, z=zt, crp_prior=True).cpu().numpy()
    audit("dm_lane_row.loo_prior", _gate_on_sum(both, feats, fs, rows, z=z, prior=crp_prior_matrix(np.bincount(z, minlength=K), 0.7, z)), TOL)
    # the same rows in three shards (few enough rows each that the library picks the tile kernel for them): the same bits,
    # as the choice between the two kernels by the call's row count requires
    parts = torch.empty((N, K), dtype=torch.float32, device=gpu_ctx.torch_device)
    for lo, n in (common_amd.dist.shard_rows(N, 3, r) for r in range(3)):
        parts[lo:lo + n] = st.score_value(view, z=zt[lo:lo + n].contiguous(), crp_prior=True, row0=lo, nrows=n)
    assert gpu_ctx.last_kernel("score").startswith("k_score_tile<")
    assert np.array_equal(parts.cpu().numpy(), both)


def test_masked_dm_rows_contribute_nothing_on_the_lane_row_kernel(gpu_ctx, monkeypatch):
    """(a masked view hands every column a mask: the bool column beside the dm feature becomes a lookup with a zero row; a
    masked nich column AND a dm feature in one plan stay with the tile kernel -- the lane <-> row kernel has an instantiation
    for either, none for both)"""
    import common_amd
    monkeypatch.setenv("MSC_TAIL_MIN_ROWS", "16384")
    rng = np.random.default_rng(14)
    N, K, C = 20_000, 21, 4
    feats = [_small_counts(make_feature(orc.DM, N, K, rng, C)), make_feature(orc.BB, N, K, rng)]
    z = rng.integers(0, K, N).astype(np.int32)
    rec = recarray_of(feats)
    mask = np.zeros(N, dtype=[("f0", np.bool_, (C,)), ("f1", np.bool_)])
    hide = rng.random(N) < 0.2
    mask["f0"][hide, rng.integers(0, C, hide.sum())] = True        # one masked element hides the whole vector
    view = common_amd.DataView.from_recarray(gpu_ctx, np.ma.masked_array(rec, mask=mask))
    keep = ~hide
    fs_keep = state_from_assignment([dict(feats[0], values=feats[0]["values"][keep])], K, z[keep])
    fs_all = state_from_assignment(feats[1:], K, z)
    st = common_amd.State(gpu_ctx, [(orc.DM, C), (orc.BB, 0)], K)
    load_state(st, [fs_keep[0]] + fs_all)
    got = st.score_value(view).cpu().numpy()
    assert gpu_ctx.last_kernel("score").startswith("k_score_tail_rows<")
    dm_part = fs_keep[0][0].score_matrix(fs_keep[0][1], feats[0]["values"])
    dm_part[hide] = 0.0
    want = dm_part + sum(f[0].score_matrix(f[1], feats[1 + i]["values"]) for i, f in enumerate(fs_all))
    assert rel_err(got, want).max() <= TOL


@pytest.mark.parametrize("K,empty", [(7, 1), (32, 0), (64, 5)])
def test_sweep_of_a_dm_state_on_the_lane_row_kernel_matches_the_oracle(gpu_ctx, K, empty, monkeypatch):
    """... and the fused sweep of such a state (scores + draw in one launch, a lane draws its own row) against the oracle's
    sweep, every disagreeing draw on a CDF step"""
    from tests.test_gpu_sweep import _check_agreement, _run
    monkeypatch.setenv("MSC_TAIL_MIN_ROWS", "1")
    specs = [(orc.DM, 4), (orc.BB, 0), (orc.DM, 3), (orc.NICH, 0)]
    got, want, scores, _ = _run(gpu_ctx, specs, 3000, K, seed=60 + K, sweep_idx=2, alpha=0.8, empty=empty, small_dm=True)
    assert gpu_ctx.last_kernel("sweep").startswith("k_score_tail_rows<")
    _check_agreement(got, want, scores, 60 + K, 2, 0.995)
