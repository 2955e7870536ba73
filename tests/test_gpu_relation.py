"""GPU: a relation (irm's data: an entity x entity array with missing cells) served by the same kernels.
The relation's cells become the rows of a one-feature dataview, msc_relation_blocks maps the two
domains' cluster assignments to the cell's block, and accumulate / score_value work per block
(relation/dataview.hpp:25-578)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.gpu_helpers import TOL, audit, rel_err

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


def _oracle_entity_scores(F, hp_ss_of_block, rel, zs, Ks, dim, e):
    """what irm computes for entity e of dimension dim: remove its slice's cells from their blocks, then for every
    candidate cluster g sum score_value of the slice's cells against the blocks they would lie in (host loop over the
    slice, as relation/dataview.hpp's slice iterators feed it)"""
    data, mask = np.ma.getdata(rel), np.ma.getmaskarray(rel)
    nd = rel.ndim
    strides = [int(np.prod(Ks[d + 1:])) for d in range(nd)]
    ss = hp_ss_of_block.copy()
    cells = [idx for idx in np.ndindex(*rel.shape) if idx[dim] == e and not mask[idx] and all(zs[d][idx[d]] >= 0 for d in range(nd) if d != dim)]
    if zs[dim][e] >= 0:
        for idx in cells:                                       # the entity leaves: its cells leave their blocks
            b = sum(int(zs[d][idx[d]]) * strides[d] for d in range(nd))
            F.remove_value(ss, b, data[idx])
    out, mag = np.zeros(Ks[dim]), np.zeros(Ks[dim])
    for g in range(Ks[dim]):
        for idx in cells:
            b = g * strides[dim] + sum(int(zs[d][idx[d]]) * strides[d] for d in range(nd) if d != dim)
            s = F.score_value(ss, b, data[idx])
            out[g] += s
            mag[g] += max(1.0, abs(s))
    return out, np.maximum(mag, 1.0)


@pytest.mark.parametrize("family,shape,Ks", [(orc.BB, (23, 31), (4, 5)), (orc.NICH, (9, 7, 6), (3, 2, 4)), (orc.GP, (20, 17), (3, 70))])
def test_slice_scores_are_irms_per_entity_candidate_scores(gpu_ctx, family, shape, Ks):
    """msc_relation_slice_scores against the per-entity loop irm runs over relation slices: for a handful of entities of
    every dimension, remove the entity's cells (msc_accumulate SUBTRACT), score the cells against every block, reduce the
    slice per candidate cluster, put the cells back"""
    import common_amd
    rng = np.random.default_rng(sum(shape))
    if family == orc.BB:
        data = rng.random(shape) < 0.4
        hp = dict(alpha=1.0, beta=1.0)
    elif family == orc.GP:
        data = rng.poisson(3.0, shape).astype(np.uint32)
        hp = dict(alpha=1.0, inv_beta=1.0)
    else:
        data = rng.normal(0, 2, shape).astype(np.float32)
        hp = dict(mu=0., kappa=1., sigmasq=1., nu=1.)
    rel = np.ma.masked_array(data, mask=rng.random(shape) < 0.25)
    view = common_amd.RelationView(gpu_ctx, rel)
    dev = gpu_ctx.torch_device
    nd, nblocks = len(shape), int(np.prod(Ks))
    zs = [rng.integers(0, k, n).astype(np.int32) for n, k in zip(shape, Ks)]
    zs[-1][0] = -1                                              # an unassigned entity on the last dimension
    zt = [torch.from_numpy(z).to(dev) for z in zs]
    zc = view.blocks(zt, Ks)
    st = common_amd.State(gpu_ctx, [(family, 0)], nblocks)
    st.accumulate(view.cells, zc)
    F = orc.Family(family, hp, 0, "f64")
    base = orc.widen_ss(family, st.get_ss(0).astype(orc.ss_dtype(family, 0, "f32")))
    zch = zc.cpu().numpy()
    ncells = view.cells.nrows
    cell_index = np.arange(ncells).reshape(shape)
    for dim in range(nd):
        off = view.slice_offsets(zt, Ks, dim)
        ents = [0, shape[dim] - 1, int(rng.integers(1, shape[dim] - 1))]
        for e in ents:
            in_slice = np.zeros(ncells, dtype=bool)
            in_slice[np.take(cell_index, e, axis=dim).reshape(-1)] = True
            z_rm = torch.from_numpy(np.where(in_slice, zch, -1).astype(np.int32)).to(dev)
            st.accumulate(view.cells, z_rm, reset=False, subtract=True)          # the entity's cells leave their blocks
            scores = st.score_value(view.cells)                                   # [ncells, nblocks]
            got = view.slice_scores(scores, off, dim, Ks)[e].cpu().numpy()
            st.accumulate(view.cells, z_rm, reset=False)                          # ... and come back
            want, mag = _oracle_entity_scores(F, base, rel, zs, Ks, dim, e)
            # a sum of up to ~60 float cell scores, each within the plain gate of its own magnitude: the errors add
            # (round 2: 4x the gate on the sum's magnitude)
            audit("relation.slice_sum_vs_oracle", (np.abs(got - want) / mag).max(), TOL)
    # all entities of a dimension in one call: every row equals the reduction done on the host from the same score matrix
    scores = st.score_value(view.cells)
    sh = scores.cpu().numpy().astype(np.float64)
    for dim in range(nd):
        off = view.slice_offsets(zt, Ks, dim)
        got = view.slice_scores(scores, off, dim, Ks).cpu().numpy()
        offh = off.cpu().numpy()
        stride = int(np.prod(Ks[dim + 1:]))
        want = np.zeros((shape[dim], Ks[dim]))
        for e in range(shape[dim]):
            for c in np.take(cell_index, e, axis=dim).reshape(-1):
                if offh[c] >= 0:
                    want[e] += sh[c, offh[c] + stride * np.arange(Ks[dim])]
        assert rel_err(got, want).max() <= 1e-6
    off0 = view.slice_offsets(zt, Ks, 0)
    with pytest.raises(ValueError):
        view.slice_scores(scores[:, :2].contiguous(), off0, 0, Ks)   # score rows shorter than the blocks: refused on the host
    # ... and below the wrapper: the C ABI refuses candidates that run past the row, and a row that holds the candidates
    # but not every offset (off is caller data) is caught by the kernel -- skipped, reported, MSC_EDEVICE at the next call
    import ctypes as C
    lib, nd = gpu_ctx.lib, len(shape)
    stride0 = int(np.prod(Ks[1:]))
    out = torch.zeros((shape[0], Ks[0]), dtype=torch.float32, device=dev)
    shp = (C.c_uint64 * nd)(*shape)

    def raw(mat):
        return lib.msc_relation_slice_scores(gpu_ctx._h, C.c_void_p(mat.data_ptr()), mat.stride(0), nd, shp, 0, None, None,
                                             C.c_void_p(off0.data_ptr()), int(Ks[0]), stride0, shape[0],
                                             C.c_void_p(out.data_ptr()), out.stride(0))
    assert raw(scores[:, :2].contiguous()) == -1                     # MSC_EINVAL: (ncand - 1) * stride >= ld
    narrow = scores[:, :(Ks[0] - 1) * stride0 + 1].contiguous()      # holds candidate ncand - 1 at offset 0 only
    assert int(off0.max()) > 0
    assert raw(narrow) == 0                                          # launched: the arguments alone are consistent
    with pytest.raises(common_amd.MicroscopesHipError) as ei:
        gpu_ctx.synchronize()
    assert ei.value.code == -6 and "offset beyond the score row" in str(ei.value)
    gpu_ctx.synchronize()                                            # reported once
    assert bool(torch.isfinite(view.slice_scores(scores, off0, 0, Ks)).all())     # and the context works on


def test_sparse_2d_relation_equals_the_dense_masked_one(gpu_ctx):
    """sparse_2d_dataview (relation/dataview.pyx; compressed_2darray): the stored entries of a scipy.sparse matrix as
    cells -- block suff-stats, per-cell scores and both dimensions' slice reductions equal those of the dense relation
    with the absent entries masked"""
    import scipy.sparse as sp
    import common_amd
    rng = np.random.default_rng(21)
    n0, n1, K0, K1 = 41, 29, 5, 3
    present = rng.random((n0, n1)) < 0.35
    present[7, :] = False                                       # an empty row and an empty column
    present[:, 4] = False
    vals = rng.poisson(2.5, (n0, n1)).astype(np.uint32) + 1     # (stored zeros would be dropped by scipy)
    m = sp.csr_matrix(np.where(present, vals, 0).astype(np.uint32))
    sview = common_amd.SparseRelationView(gpu_ctx, m)
    dview = common_amd.RelationView(gpu_ctx, np.ma.masked_array(vals, mask=~present))
    assert sview.nnz() == int(present.sum()) and sview.shape == (n0, n1)
    dev = gpu_ctx.torch_device
    zs = [rng.integers(0, K0, n0).astype(np.int32), rng.integers(0, K1, n1).astype(np.int32)]
    zs[1][3] = -1
    zt = [torch.from_numpy(z).to(dev) for z in zs]
    Ks = [K0, K1]
    zc_s, zc_d = sview.blocks(zt, Ks), dview.blocks(zt, Ks)
    cell_of = np.flatnonzero(present.reshape(-1))               # CSR order = row-major order of the present cells
    assert np.array_equal(zc_s.cpu().numpy(), zc_d.cpu().numpy()[cell_of])
    st_s = common_amd.State(gpu_ctx, [(orc.GP, 0)], K0 * K1)
    st_d = common_amd.State(gpu_ctx, [(orc.GP, 0)], K0 * K1)
    st_s.accumulate(sview.cells, zc_s)
    st_d.accumulate(dview.cells, zc_d)
    a, b = st_s.get_ss(0), st_d.get_ss(0)
    assert np.array_equal(a["count"], b["count"]) and np.array_equal(a["sum"], b["sum"])
    sc_s, sc_d = st_s.score_value(sview.cells), st_d.score_value(dview.cells)
    assert rel_err(sc_s.cpu().numpy(), sc_d.cpu().numpy()[cell_of]).max() <= TOL
    for dim in (0, 1):
        got = sview.slice_scores(sc_s, sview.slice_offsets(zt, Ks, dim), dim, Ks).cpu().numpy()
        want = dview.slice_scores(sc_d, dview.slice_offsets(zt, Ks, dim), dim, Ks).cpu().numpy()
        assert got.shape == (m.shape[dim], Ks[dim])
        audit("relation.sparse_vs_dense_slice_sums", rel_err(got, want).max(), TOL)   # (the same terms in the same order: 0 measured; round 2: 2e-6)
    assert np.all(sview.slice_scores(sc_s, sview.slice_offsets(zt, Ks, 0), 0, Ks).cpu().numpy()[7] == 0.0)   # the empty row
    with pytest.raises(ValueError):
        common_amd.RelationView(gpu_ctx, np.zeros((3, 0)))      # empty dims not allowed (relation/_dataview.pyx:33-34)
