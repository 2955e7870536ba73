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
