"""GPU: masked (missing) values.  The reference never hands a masked value to a group
(distributions.hpp:269,276,283 assert !anymasked; its callers skip masked features), so a masked
(row, feature) must add nothing to a score and nothing to the suff-stats."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.gpu_helpers import TOL, load_state, make_feature, rel_err

pytestmark = pytest.mark.gpu

SPECS = [(orc.BB, 0), (orc.GP, 0), (orc.DD, 6), (orc.NICH, 0), (orc.NIW, 3)]


def _masked_setup(gpu_ctx, specs, N, K, seed, frac=0.3):
    import common_amd
    rng = np.random.default_rng(seed)
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z = rng.integers(0, K, N).astype(np.int32)
    masks = [rng.random(N) < frac for _ in feats]
    dt = np.dtype([("f%d" % i, f["np_dtype"]) for i, f in enumerate(feats)])
    data = np.zeros(N, dtype=dt)
    mask = np.zeros(N, dtype=[("f%d" % i, np.bool_, np.dtype(f["np_dtype"]).shape) for i, f in enumerate(feats)])
    for i, f in enumerate(feats):
        data["f%d" % i] = f["values"]
        m = masks[i]
        mask["f%d" % i] = m if np.dtype(f["np_dtype"]).shape == () else np.repeat(m[:, None], f["dim"], 1)
    arr = np.ma.masked_array(data, mask=mask)
    view = common_amd.DataView.from_recarray(gpu_ctx, arr)
    st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    # suff-stats the oracle's way: a masked row is simply not added to that feature's groups
    fs = []
    for f, m in zip(feats, masks):
        F = orc.Family(f["family"], f["hp"], f["dim"], "f64")
        ss64 = F.accumulate(K, f["values"], np.where(m, -1, z).astype(np.int32))
        ss32 = orc.narrow_ss(f["family"], ss64, f["dim"])
        fs.append((F, orc.widen_ss(f["family"], ss32, f["dim"]), ss32))
    return feats, masks, z, fs, view, st


def _oracle(feats, masks, fs, z=None):
    total = None
    for f, m, (F, ss64, _) in zip(feats, masks, fs):
        sc = F.score_matrix(ss64, f["values"], None if z is None else np.where(m, -1, z).astype(np.int32))
        sc[m] = 0.0
        total = sc if total is None else total + sc
    return total


def test_masked_values_are_skipped_by_accumulate(gpu_ctx):
    feats, masks, z, fs, view, st = _masked_setup(gpu_ctx, SPECS, 4000, 23, seed=1)
    st.accumulate(view, torch.from_numpy(z).to(gpu_ctx.torch_device))
    for i, (F, ss64, _) in enumerate(fs):
        rec = st.get_ss(i)
        for name in rec.dtype.names:
            a, b = rec[name].astype(np.float64), np.asarray(ss64[name], np.float64)
            if np.issubdtype(rec.dtype[name].base, np.integer):
                assert np.array_equal(a, b), (F.family, name)
            else:
                assert rel_err(a, b).max() <= TOL, (F.family, name)
    assert np.array_equal(st.get_group_counts(), np.bincount(z, minlength=23))   # rows still belong to groups


@pytest.mark.parametrize("specs", [SPECS, [(orc.NICH, 0)], [(orc.BB, 0), (orc.DD, 6)]])
def test_masked_values_add_nothing_to_scores(gpu_ctx, specs):
    feats, masks, z, fs, view, st = _masked_setup(gpu_ctx, specs, 1500, 40, seed=2)
    load_state(st, fs)
    got = st.score_value(view).cpu().numpy()
    assert rel_err(got, _oracle(feats, masks, fs)).max() <= TOL
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    got = st.score_value(view, z=zt).cpu().numpy()
    assert rel_err(got, _oracle(feats, masks, fs, z)).max() <= TOL


@pytest.mark.parametrize("pin", ["1", "2"])
def test_fully_masked_row_scores_zero_and_samples_from_the_prior(gpu_ctx, monkeypatch, pin):
    monkeypatch.setenv("MSC_SWEEP_NICH1", pin)          # the row-at-a-time and the transposed single-nich kernel
    import common_amd
    N, K = 20000, 8
    x = np.ma.masked_array(np.zeros(N, dtype=[("f0", np.float32)]), mask=[(True,)] * N)
    view = common_amd.DataView.from_recarray(gpu_ctx, x)
    st = common_amd.State(gpu_ctx, [(orc.NICH, 0)], K)
    counts = np.array([5, 1, 0, 9, 20, 0, 3, 2], dtype=np.uint32)
    st.set_group_counts(counts)
    st.set_alpha(2.0)
    assert np.all(st.score_value(view).cpu().numpy() == 0.0)
    zt = torch.full((N,), -1, dtype=torch.int32, device=gpu_ctx.torch_device)
    st.sweep_assign(view, zt, seed=4, sweep=0)
    emp = np.bincount(zt.cpu().numpy(), minlength=K) / N
    pc = np.where(counts > 0, counts, 2.0 / 2)            # pseudocounts: alpha / 2 empty groups
    assert np.abs(emp - pc / pc.sum()).max() < 0.015


@pytest.mark.parametrize("K,masked_cols", [(90, {0, 1, 3}), (300, {0, 3}), (120, {0, 2, 7})])
def test_masked_lookup_columns_keep_the_tile_kernels_fast_path(gpu_ctx, K, masked_cols, monkeypatch):
    """Masked bb / gp / dd columns reach the tile kernels with the mask folded in (a masked row holds the index of the
    family's zero table row: abi.cpp bind_view / plan_groups), so such a state keeps the lookup runs and the role-split
    kernels; a masked nich column (third case, column 2) still takes the generic path.  40k rows against the oracle,
    against short slices (other launch shapes: same bits), plain and leave-one-out + prior, and -- K <= 256 -- as a fused
    sweep against the same sweep in three shards; accumulate sees the original column and mask."""
    import common_amd
    monkeypatch.setenv("MSC_TAIL_MIN_ROWS", "16384")      # (40k rows: the lane <-> row kernel takes the partly filled tile)
    specs = [(orc.BB, 0), (orc.GP, 0), (orc.NICH, 0), (orc.DD, 7), (orc.BB, 0), (orc.DD, 40), (orc.NICH, 0), (orc.GP, 0),
             (orc.NICH, 0), (orc.BBNC, 0)]
    N = 40_000
    rng = np.random.default_rng(K)
    feats = [make_feature(f, N, K, rng, d) for f, d in specs]
    z = rng.integers(0, K, N).astype(np.int32)
    masks = [(rng.random(N) < 0.2) if i in masked_cols else np.zeros(N, dtype=bool) for i in range(len(feats))]
    dev = gpu_ctx.torch_device
    cols = [torch.from_numpy(np.ascontiguousarray(f["values"])).to(dev) for f in feats]
    mts = [torch.from_numpy(masks[i].astype(np.uint8)).to(dev) if i in masked_cols else None for i in range(len(feats))]
    view = common_amd.DataView.from_tensors(gpu_ctx, cols, mts)
    st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    fs = []
    for f, m in zip(feats, masks):
        F = orc.Family(f["family"], f["hp"], f["dim"], "f64")
        init = None
        if f["family"] == orc.BBNC:
            init = np.zeros(K, dtype=orc.ss_dtype(orc.BBNC, 0, "f64"))
            init["p"] = np.random.default_rng(K).uniform(0.05, 0.95, K).astype(np.float32)
        ss32 = orc.narrow_ss(f["family"], F.accumulate(K, f["values"], np.where(m, -1, z).astype(np.int32), ss_init=init), f["dim"])
        fs.append((F, orc.widen_ss(f["family"], ss32, f["dim"]), ss32))
    load_state(st, fs)
    st.set_group_counts(np.bincount(z, minlength=K).astype(np.uint32))
    st.set_alpha(0.9)
    zt = torch.from_numpy(z).to(dev)
    rows = np.random.default_rng(1).choice(N, 300, replace=False)

    def twin(zz):
        total, mag = None, None
        for f, m, (F, ss64, _) in zip(feats, masks, fs):
            sc = F.score_matrix(ss64, f["values"][rows], None if zz is None else np.where(m, -1, zz).astype(np.int32)[rows])
            sc[m[rows]] = 0.0
            total = sc if total is None else total + sc
            mag = np.maximum(1.0, np.abs(sc)) if mag is None else mag + np.maximum(1.0, np.abs(sc))
        return total, mag
    plain = st.score_value(view)
    want, mag = twin(None)
    assert (np.abs(plain.cpu().numpy()[rows] - want) / mag).max() <= TOL
    loo = st.score_value(view, z=zt)
    want, mag = twin(z)
    assert (np.abs(loo.cpu().numpy()[rows] - want) / mag).max() <= TOL
    full = st.score_value(view, z=zt, crp_prior=True)
    for row0, n in ((0, 129), (N - 77, 77), (N // 2 + 3, 500)):
        assert torch.equal(st.score_value(view, row0=row0, nrows=n), plain[row0:row0 + n])
        assert torch.equal(st.score_value(view, row0=row0, nrows=n, z=zt[row0:row0 + n].contiguous(), crp_prior=True),
                           full[row0:row0 + n])
    if K <= 256:                                           # the fused mixed sweep: whole against three shards
        whole = zt.clone()
        st.sweep_assign(view, whole, seed=5, sweep=1)
        parts = zt.clone()
        for lo, n in (common_amd.dist.shard_rows(N, 3, r) for r in range(3)):
            zs = parts[lo:lo + n].contiguous()
            st.sweep_assign(view, zs, seed=5, sweep=1, row0=lo, nrows=n, row_id0=lo)
            parts[lo:lo + n] = zs
        assert torch.equal(whole, parts)
        assert (whole.cpu().numpy() != z).mean() > 0.05
    # the suff-stats a device-side accumulate builds (original column + mask) are the masked rows left out
    st2 = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    for i, (F, _, ss32) in enumerate(fs):
        st2.set_hp(i, F.hp)
    st2.accumulate(view, zt)
    for i, (F, _, ss32) in enumerate(fs):
        rec = st2.get_ss(i)
        for name in rec.dtype.names:
            if np.issubdtype(rec.dtype[name].base, np.integer):
                assert np.array_equal(rec[name], ss32[name]), (i, name)
