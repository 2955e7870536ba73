"""GPU: a relation (irm's data: an entity x entity array with missing cells) served by the same kernels.
The relation's cells become the rows of a one-feature dataview, msc_relation_blocks maps the two
domains' cluster assignments to the cell's block, and accumulate / score_value work per block
(relation/dataview.hpp:25-578)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.gpu_helpers import TOL, rel_err

pytestmark = pytest.mark.gpu


def _relation(rng, shape, frac_missing=0.3):
    data = rng.random(shape) < 0.4
    mask = rng.random(shape) < frac_missing
    return np.ma.masked_array(data, mask=mask)


def test_cells_blocks_and_block_suffstats_of_a_2d_relation(gpu_ctx):
    import common_amd
    rng = np.random.default_rng(3)
    n0, n1, K0, K1 = 37, 53, 4, 5
    rel = _relation(rng, (n0, n1))
    view = common_amd.RelationView(gpu_ctx, rel)
    assert view.cells.nrows == n0 * n1
    z0 = rng.integers(0, K0, n0).astype(np.int32)
    z1 = rng.integers(0, K1, n1).astype(np.int32)
    z0[5] = -1                                                  # an entity that is not assigned yet
    dev = gpu_ctx.torch_device
    zc = view.blocks([torch.from_numpy(z0).to(dev), torch.from_numpy(z1).to(dev)], [K0, K1])
    want = np.where((z0[:, None] >= 0) & (z1[None, :] >= 0), z0[:, None] * K1 + z1[None, :], -1).reshape(-1)
    assert np.array_equal(zc.cpu().numpy(), want)               # bit-exact index arithmetic
    # block suff-stats: heads / tails of the present cells of every block
    st = common_amd.State(gpu_ctx, [(orc.BB, 0)], K0 * K1)
    st.accumulate(view.cells, zc)
    rec = st.get_ss(0)
    present = ~np.ma.getmaskarray(rel).reshape(-1) & (want >= 0)
    vals = np.ma.getdata(rel).reshape(-1)
    heads = np.bincount(want[present & vals], minlength=K0 * K1)
    tails = np.bincount(want[present & ~vals], minlength=K0 * K1)
    assert np.array_equal(rec["heads"], heads) and np.array_equal(rec["tails"], tails)
    assert np.array_equal(st.get_group_counts(), np.bincount(want[want >= 0], minlength=K0 * K1))
    # per-cell scores against every block, leave-one-out for the cell's own block; a missing cell scores 0
    got = st.score_value(view.cells, z=zc).cpu().numpy()
    F = orc.Family(orc.BB, dict(alpha=1.0, beta=1.0), 0, "f64")
    ss = np.zeros(K0 * K1, dtype=orc.ss_dtype(orc.BB, 0, "f64"))
    ss["heads"], ss["tails"] = heads, tails
    zz = np.where(present, want, -1).astype(np.int32)           # a missing cell was never added to its block
    ref = F.score_matrix(ss, vals.astype(np.uint8), zz)
    ref[np.ma.getmaskarray(rel).reshape(-1)] = 0.0
    assert rel_err(got, ref).max() <= TOL


def test_three_dimensional_relation_and_bad_arguments(gpu_ctx):
    import common_amd
    from common_amd._lib import MicroscopesHipError
    rng = np.random.default_rng(4)
    shape, Ks = (6, 5, 7), (2, 3, 2)
    rel = rng.normal(size=shape).astype(np.float32)
    view = common_amd.RelationView(gpu_ctx, rel)
    dev = gpu_ctx.torch_device
    zs = [rng.integers(0, k, n).astype(np.int32) for n, k in zip(shape, Ks)]
    zc = view.blocks([torch.from_numpy(z).to(dev) for z in zs], Ks).cpu().numpy()
    want = (zs[0][:, None, None] * Ks[1] + zs[1][None, :, None]) * Ks[2] + zs[2][None, None, :]
    assert np.array_equal(zc, want.reshape(-1))
    st = common_amd.State(gpu_ctx, [(orc.NICH, 0)], int(np.prod(Ks)))
    st.accumulate(view.cells, torch.from_numpy(zc).to(dev))
    assert np.array_equal(st.get_ss(0)["count"], np.bincount(zc, minlength=int(np.prod(Ks))))
    with pytest.raises(ValueError):
        view.blocks([torch.from_numpy(zs[0]).to(dev)], [2])      # one vector per dimension
    with pytest.raises(MicroscopesHipError):
        view.blocks([torch.from_numpy(z).to(dev) for z in zs], [2, 3, 0])   # a dimension without groups
