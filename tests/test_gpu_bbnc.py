"""GPU: the non-conjugate Beta-Bernoulli model (src/models/bbnc.cpp:22-73): an explicit per-group p
that is state (not a statistic), heads/tails that only enter score_data."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.gpu_helpers import TOL, load_state, make_feature, oracle_scores, recarray_of, rel_err, state_from_assignment

pytestmark = pytest.mark.gpu


def test_bbnc_accumulate_keeps_p_and_counts_heads_tails(gpu_ctx):
    import common_amd
    rng = np.random.default_rng(2)
    N, K = 3000, 12
    feats = [make_feature(orc.BBNC, N, K, rng), make_feature(orc.NICH, N, K, rng)]
    z = rng.integers(0, K, N).astype(np.int32)
    fs = state_from_assignment(feats, K, z)
    view = common_amd.DataView.from_recarray(gpu_ctx, recarray_of(feats))
    st = common_amd.State(gpu_ctx, [(orc.BBNC, 0), (orc.NICH, 0)], K)
    empty = np.zeros(K, dtype=common_amd.ss_dtype(orc.BBNC))
    empty["p"] = fs[0][2]["p"]
    st.set_ss(0, empty)
    zt = torch.from_numpy(z).to(gpu_ctx.torch_device)
    st.accumulate(view, zt)
    rec = st.get_ss(0)
    assert np.array_equal(rec["p"], fs[0][2]["p"])
    assert np.array_equal(rec["heads"], fs[0][2]["heads"]) and np.array_equal(rec["tails"], fs[0][2]["tails"])
    got = st.score_value(view, z=zt).cpu().numpy()
    assert rel_err(got, oracle_scores(feats, fs, z=z)).max() <= TOL
    # golden: score_value = log p / log(1-p); score_data closed form with scipy's betaln
    from scipy.special import betaln
    sd = st.score_data().cpu().numpy()[0]
    p, h, t = rec["p"].astype(np.float64), rec["heads"], rec["tails"]
    want = -betaln(1.0, 1.0) + h * np.log(p) + t * np.log1p(-p)      # alpha = beta = 1: flat prior
    assert rel_err(sd, want).max() <= TOL
