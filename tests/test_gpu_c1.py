"""BASELINE config C1 on the device at its exact workload (Beta-Bernoulli, N=10k, K=16, D=8, alpha = beta = 2, seed
73): the same inputs the CPU baseline driver generates, the full [N, K] matrix and the suff-stats against the oracle."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.gpu_helpers import TOL, rel_err
from tests.test_c1_cpu import D, HP, K, N, run_c1

pytestmark = pytest.mark.gpu


def test_c1_full_matrix_and_suffstats(gpu_ctx, tmp_path):
    import common_amd
    rec, z, cols = run_c1(tmp_path)
    arr = np.zeros(N, dtype=[("f%d" % f, np.bool_) for f in range(D)])       # 8-byte records, as perf_group's 8 x TYPE_B
    for f in range(D):
        arr["f%d" % f] = cols[f] != 0
    assert arr.dtype.itemsize == 8
    view = common_amd.DataView.from_recarray(gpu_ctx, arr)
    st = common_amd.State(gpu_ctx, [(common_amd.BB, 0)] * D, K)
    for f in range(D):
        st.set_hp(f, HP)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.accumulate(view, zt)
    F64 = orc.Family(orc.BB, HP, 0, "f64")
    want = np.zeros((N, K))
    for f in range(D):
        ss = F64.accumulate(K, cols[f], z)
        got = st.get_ss(f)
        assert np.array_equal(got["heads"], ss["heads"]) and np.array_equal(got["tails"], ss["tails"])     # bit-exact
        want += F64.score_matrix(ss, cols[f])
    assert np.array_equal(st.get_group_counts(), np.bincount(z, minlength=K))
    out = st.score_value(view).cpu().numpy()
    assert out.shape == (N, K)
    assert rel_err(out, want).max() <= TOL                     # all 160 000 entries = 1.28 M evaluations
    # the checksum the CPU driver prints ("ignore:" in bin/perf_group.cpp:90,107,124) is reproduced by the device
    assert abs(float(out.astype(np.float64).sum()) - rec["score_sum"]) <= 1e-6 * abs(rec["score_sum"])
    # leave-one-out (remove_value before score_value, the perf_group iteration's add / remove / score shape)
    loo = st.score_value(view, z=zt).cpu().numpy()
    want_loo = np.zeros((N, K))
    for f in range(D):
        ss = F64.accumulate(K, cols[f], z)
        want_loo += F64.score_matrix(ss, cols[f], z)
    assert rel_err(loo, want_loo).max() <= TOL
    # score_data for all 8 x 16 groups
    sd = st.score_data().cpu().numpy()
    for f in range(D):
        ss = F64.accumulate(K, cols[f], z)
        assert rel_err(sd[f], F64.score_data_all(ss)).max() <= TOL
