"""GPU: seeded random feature lists against the oracle.  The tile kernels walk a host-made plan (feature groups
sharing the LDS slot, runs of lookup features, a nich tail phase, masked features on the generic path, tables
larger than a feature may stage); random mixtures of families, dimensions, masks, group counts and row counts
exercise the planner's corner cases, which hand-picked cases tend to miss."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.gpu_helpers import TOL, audit, crp_prior_matrix, load_state, make_feature, rel_err

pytestmark = pytest.mark.gpu

FAMS = [orc.BB, orc.BBNC, orc.GP, orc.BNB, orc.DD, orc.NICH, orc.NIW, orc.DM]


def _random_spec(rng):
    nf = int(rng.integers(1, 24))
    spec = []
    for _ in range(nf):
        fam = FAMS[int(rng.integers(0, len(FAMS)))]
        if fam == orc.NIW and rng.random() < 0.7:
            fam = orc.NICH                                   # keep niw rare (it owns a separate kernel)
        dim = 0
        if fam == orc.DD:
            dim = int(rng.choice([2, 3, 17, 64, 65, 100, 128]))   # 65+: more categories than a feature may stage
        elif fam == orc.NIW:
            # (1 .. 5: the register kernel; now and then the f64 matrix kernels of dim <= 16, <= 32 and beyond)
            dim = int(rng.integers(1, 6)) if rng.random() < 0.7 else int(rng.choice([9, 16, 17, 32, 33, 48]))
        elif fam == orc.DM:
            dim = int(rng.integers(2, 9))
        spec.append((fam, dim))
    return spec


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MSC_FUZZ_SEEDS", "32"))))
def test_random_feature_lists_match_the_oracle(gpu_ctx, seed):
    import common_amd
    rng = np.random.default_rng(1000 + seed)
    spec = _random_spec(rng)
    N = int(rng.choice([1, 5, 64, 127, 129, 300, 777]))
    K = int(rng.choice([1, 3, 16, 20, 33, 64, 255, 256, 257, 300]))
    feats = [make_feature(f, N, K, rng, d) for f, d in spec]
    if rng.random() < 0.3:                                       # gp counts beyond what a feature stages / tables hold
        for f in feats:
            if f["family"] == orc.GP and N > 2:
                f["values"][rng.integers(0, N, 2)] += np.uint32(rng.choice([70, 2000]))
    z = rng.integers(0, K, N).astype(np.int32)
    # masks on a random subset of the features
    masked = [rng.random() < 0.25 for _ in feats]
    dt = np.dtype([("f%d" % i, f["np_dtype"]) for i, f in enumerate(feats)])
    data = np.zeros(N, dtype=dt)
    mask = np.zeros(N, dtype=[("f%d" % i, np.bool_, np.dtype(f["np_dtype"]).shape) for i, f in enumerate(feats)])
    rowmask = []
    for i, f in enumerate(feats):
        data["f%d" % i] = f["values"]
        m = (rng.random(N) < 0.3) if masked[i] else np.zeros(N, dtype=bool)
        rowmask.append(m)
        mask["f%d" % i] = m if np.dtype(f["np_dtype"]).shape == () else np.repeat(m[:, None], f["dim"], 1)
    arr = np.ma.masked_array(data, mask=mask) if any(masked) else data
    view = common_amd.DataView.from_recarray(gpu_ctx, arr)
    st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    # suff-stats the oracle's way (a masked row is not part of the feature's groups)
    fs, total, total_loo, mag, mag_loo = [], None, None, None, None
    for f, m in zip(feats, rowmask):
        F = orc.Family(f["family"], f["hp"], f["dim"], "f64")
        init = None
        if f["family"] == orc.BBNC:
            init = np.zeros(K, dtype=orc.ss_dtype(orc.BBNC, 0, "f64"))
            init["p"] = np.random.default_rng(K).uniform(0.05, 0.95, K).astype(np.float32)
        zz = np.where(m, -1, z).astype(np.int32)
        ss64 = F.accumulate(K, f["values"], zz, ss_init=init)
        ss32 = orc.narrow_ss(f["family"], ss64, f["dim"])
        ss64 = orc.widen_ss(f["family"], ss32, f["dim"])
        fs.append((F, ss64, ss32))
        a = F.score_matrix(ss64, f["values"])
        b = F.score_matrix(ss64, f["values"], zz)
        a[m] = 0.0
        b[m] = 0.0
        total = a if total is None else total + a
        total_loo = b if total_loo is None else total_loo + b
        mag = np.maximum(1.0, np.abs(a)) if mag is None else mag + np.maximum(1.0, np.abs(a))
        mag_loo = np.maximum(1.0, np.abs(b)) if mag_loo is None else mag_loo + np.maximum(1.0, np.abs(b))
    load_state(st, fs)
    cnt = np.bincount(z, minlength=K).astype(np.uint32)
    st.set_group_counts(cnt)
    # the gate of a SUM of D float feature scores: every feature within the north star's 1e-6 * max(1, |score_f|), and
    # the errors add -- 1e-6 * sum_f max(1, |score_f|) (scores of mixed sign cancel in the sum but their errors do not).
    # No factor per family and none per feature count beyond that (round 2 had 4x for niw and len // 8).
    got = st.score_value(view).cpu().numpy()
    audit("fuzz.sum_of_features", (np.abs(got - total) / mag).max(), TOL)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.set_alpha(1.7)
    both = st.score_value(view, z=zt, crp_prior=True).cpu().numpy()
    want = total_loo + crp_prior_matrix(cnt, 1.7, z)
    audit("fuzz.sum_of_features_loo_prior", (np.abs(both - want) / np.maximum(mag_loo, np.abs(want))).max(), TOL)
    # and the integer suff-stats of a device-side accumulate
    st2 = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    for i, (F, _, ss32) in enumerate(fs):
        st2.set_hp(i, F.hp)
        if F.family == orc.BBNC:
            e = np.zeros(K, dtype=common_amd.ss_dtype(orc.BBNC))
            e["p"] = ss32["p"]
            st2.set_ss(i, e)
    st2.accumulate(view, zt)
    for i, (F, _, ss32) in enumerate(fs):
        rec = st2.get_ss(i)
        for name in rec.dtype.names:
            if np.issubdtype(rec.dtype[name].base, np.integer):
                assert np.array_equal(rec[name], ss32[name]), (seed, i, name)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MSC_FUZZ_SWEEP_SEEDS", "12"))))
def test_random_sweeps_draw_the_oracles_assignments(gpu_ctx, seed):
    """fused tile sweep (K <= 256), fused nich sweep, and the materialising path (K > 256, niw), unmasked"""
    from tests.test_gpu_sweep import _check_agreement, _run
    rng = np.random.default_rng(5000 + seed)
    spec = _random_spec(rng)
    N = int(rng.choice([64, 300, 1000]))
    K = int(rng.choice([2, 17, 40, 64, 256, 300]))
    got, want, scores, _ = _run(gpu_ctx, spec, N, K, seed=300 + seed, sweep_idx=seed % 5, alpha=0.9, empty=min(2, K - 1))
    _check_agreement(got, want, scores, 300 + seed, seed % 5, 0.98)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MSC_FUZZ_ROWS_SEEDS", "16"))))
def test_random_scalar_feature_lists_on_many_rows(gpu_ctx, seed):
    """The kernels a plan takes when the rows fill the chip -- role-split, nich-only (with or without a few lookups from L2),
    lookups-only, each in PAIR mode up to 128 groups, the lane <-> row kernel, the fused tail -- chosen by the plan's prices:
    random lists of scalar families (some columns masked), 33k-70k rows, 40-384 groups.  A sample of rows against the oracle
    (plain and leave-one-out + prior), slices of a few hundred rows -- the kernels that run the phases one after the other
    -- bit for bit equal to the whole, and a sweep in three shards equal to the whole's."""
    import common_amd
    rng = np.random.default_rng(7000 + seed)
    fams = [orc.BB, orc.BB, orc.GP, orc.BNB, orc.DD, orc.NICH, orc.NICH, orc.BBNC]
    kind = seed % 4                                              # 0 mixed, 1 lookups only, 2 nich only, 3 mostly nich
    nf = int(rng.integers(2, 14))
    spec = []
    for i in range(nf):
        fam = fams[int(rng.integers(0, len(fams)))]
        if kind == 1 and fam == orc.NICH:
            fam = orc.GP
        if kind == 2 or (kind == 3 and i >= 2):
            fam = orc.NICH
        spec.append((fam, int(rng.choice([2, 9, 33, 100, 128])) if fam == orc.DD else 0))
    # (round 5: now and then dm features with small counts -- tables staged whole: the lane <-> row kernel's dm instantiation
    # up to 128 groups, the 8-wave tile kernel beyond and for the slices of a few hundred rows, which must agree bit for bit)
    n_dm = int(rng.integers(1, 3)) if kind in (0, 1) and seed % 3 == 0 else 0
    for _ in range(n_dm):
        spec.insert(int(rng.integers(0, len(spec) + 1)), (orc.DM, int(rng.integers(2, 6))))
    N = int(rng.choice([33_000, 50_000, 70_000])) + int(rng.integers(0, 700))
    K = int(rng.choice([40, 64, 100, 128, 200, 256, 300, 384])) - int(rng.integers(0, 3))
    feats = [make_feature(f, N, K, rng, d) for f, d in spec]
    feats = [dict(f, values=(f["values"] // 5).astype(np.int32)) if f["family"] == orc.DM else f for f in feats]
    z = rng.integers(0, K, N).astype(np.int32)
    z[7] = -1
    # (masked lookup columns keep the fast kernels; now and then a masked nich column: the one-phase-after-the-other kernels)
    masked = [kind == 0 and f["family"] != orc.DM and (f["family"] != orc.NICH or seed % 16 == 0) and rng.random() < 0.2 for f in feats]
    dev = gpu_ctx.torch_device
    cols = [torch.from_numpy(np.ascontiguousarray(f["values"])).to(dev) for f in feats]
    rowmask = [(rng.random(N) < 0.2) if m else np.zeros(N, dtype=bool) for m in masked]
    mts = [torch.from_numpy(m.astype(np.uint8)).to(dev) if mk else None for m, mk in zip(rowmask, masked)]
    view = common_amd.DataView.from_tensors(gpu_ctx, cols, mts)
    st = common_amd.State(gpu_ctx, [(f["family"], f["dim"]) for f in feats], K)
    fs = []
    for f, m in zip(feats, rowmask):
        F = orc.Family(f["family"], f["hp"], f["dim"], "f64")
        init = None
        if f["family"] == orc.BBNC:
            init = np.zeros(K, dtype=orc.ss_dtype(orc.BBNC, 0, "f64"))
            init["p"] = np.random.default_rng(K).uniform(0.05, 0.95, K).astype(np.float32)
        ss32 = orc.narrow_ss(f["family"], F.accumulate(K, f["values"], np.where(m, -1, z).astype(np.int32), ss_init=init), f["dim"])
        fs.append((F, orc.widen_ss(f["family"], ss32, f["dim"]), ss32))
    load_state(st, fs)
    cnt = np.bincount(z[z >= 0], minlength=K).astype(np.uint32)
    st.set_group_counts(cnt)
    st.set_alpha(1.3)
    zt = torch.from_numpy(z).to(dev)
    rows = np.unique(np.concatenate([[0, 7, N - 1], rng.choice(N, 200, replace=False)]))

    def twin(zz):
        total, mag = None, None
        for f, m, (F, ss64, _) in zip(feats, rowmask, fs):
            sc = F.score_matrix(ss64, f["values"][rows], None if zz is None else np.where(m, -1, zz).astype(np.int32)[rows])
            sc[m[rows]] = 0.0
            total = sc if total is None else total + sc
            mag = np.maximum(1.0, np.abs(sc)) if mag is None else mag + np.maximum(1.0, np.abs(sc))
        return total, mag
    plain = st.score_value(view)
    want, mag = twin(None)
    audit("fuzz_rows.sum_of_features", (np.abs(plain.cpu().numpy()[rows] - want) / mag).max(), TOL)
    both = st.score_value(view, z=zt, crp_prior=True)
    want, mag = twin(z)
    want = want + crp_prior_matrix(cnt, 1.3, z[rows])
    audit("fuzz_rows.sum_of_features_loo_prior", (np.abs(both.cpu().numpy()[rows] - want) / np.maximum(mag, np.abs(want))).max(), TOL)
    big0 = int(rng.integers(1, N // 3))                          # (a large range off the view's start: the same kernels, another row0)
    for row0, n in ((0, 300), (N // 2 + 3, 129), (N - 70, 70), (big0, N - big0 - int(rng.integers(0, 200)))):
        assert torch.equal(st.score_value(view, row0=row0, nrows=n), plain[row0:row0 + n]), (seed, spec, K, row0)
        assert torch.equal(st.score_value(view, row0=row0, nrows=n, z=zt[row0:row0 + n].contiguous(), crp_prior=True), both[row0:row0 + n]), (seed, spec, K, row0)
    whole = zt.clone()
    st.sweep_assign(view, whole, seed=11, sweep=2)
    parts = zt.clone()
    for lo, n in (common_amd.dist.shard_rows(N, 3, r) for r in range(3)):
        zs = parts[lo:lo + n].contiguous()
        st.sweep_assign(view, zs, seed=11, sweep=2, row0=lo, nrows=n, row_id0=lo)
        parts[lo:lo + n] = zs
    assert torch.equal(whole, parts), (seed, spec, K)
    w = whole.cpu().numpy()
    assert w.min() >= 0 and w.max() < K


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MSC_FUZZ_ROWS_SWEEP_SEEDS", "8"))))
def test_random_sweeps_on_many_rows_draw_the_oracles_assignments(gpu_ctx, seed):
    """the same kinds of feature lists, 33k-40k rows: one fused assignment step against the oracle's sweep -- every
    disagreeing draw on a CDF step -- on whichever kernel the plan's prices pick"""
    from tests.test_gpu_sweep import _check_agreement, _run
    rng = np.random.default_rng(9000 + seed)
    fams = [orc.BB, orc.BB, orc.GP, orc.BNB, orc.DD, orc.NICH, orc.NICH, orc.BBNC]
    kind = seed % 4                                              # 0 mixed, 1 lookups only, 2 nich only, 3 mostly nich
    spec = []
    for i in range(int(rng.integers(2, 10))):
        fam = fams[int(rng.integers(0, len(fams)))]
        if kind == 1 and fam == orc.NICH:
            fam = orc.GP
        if kind == 2 or (kind == 3 and i >= 2):
            fam = orc.NICH
        spec.append((fam, int(rng.choice([2, 9, 33])) if fam == orc.DD else 0))
    N = 33_000 + int(rng.integers(0, 7000))
    K = int(rng.choice([40, 64, 100, 128, 200, 256, 300, 384])) - int(rng.integers(0, 3))
    empty = int(rng.integers(0, max(1, K // 8)))
    got, want, scores, z = _run(gpu_ctx, spec, N, K, seed=900 + seed, sweep_idx=3, alpha=0.9, empty=empty)
    _check_agreement(got, want, scores, 900 + seed, 3, 0.995)
