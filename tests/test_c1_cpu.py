"""BASELINE config C1 (bin/perf_group.cpp's shape: Beta-Bernoulli, N=10k rows, K=16 groups, D=8 boolean features,
alpha = beta = 2, seed 73) on the CPU reference path: the baseline driver that bench.py times (oracle/perf_group_cpu,
virtual group API, one value at a time) must score exactly what the batch entry of the oracle scores."""
import json
import os
import subprocess

import numpy as np

from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, K, D = 10_000, 16, 8
HP = dict(alpha=2.0, beta=2.0)          # bin/perf_group.cpp:43-44


def run_c1(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    dump = str(tmp_path / "c1.bin")
    out = subprocess.check_output([os.path.join(ROOT, "oracle", "perf_group_cpu"), "c1", str(N), str(K), "1", "73", dump])
    rec = json.loads(out.decode().strip().splitlines()[-1])
    raw = np.fromfile(dump, dtype=np.uint8)
    z = raw[:4 * N].view(np.int32).copy()
    cols = raw[4 * N:].reshape(D, N).copy()
    return rec, z, cols


def test_perf_group_cpu_c1_scores_what_the_batch_oracle_scores(tmp_path):
    rec, z, cols = run_c1(tmp_path)
    assert (rec["N"], rec["K"], rec["D"], rec["evals"]) == (N, K, D, N * K * D)
    assert z.min() >= 0 and z.max() < K and set(np.unique(cols)) <= {0, 1}
    F32 = orc.Family(orc.BB, HP, 0, "f32")
    F64 = orc.Family(orc.BB, HP, 0, "f64")
    total32, total64 = 0.0, 0.0
    for f in range(D):
        ss = F32.accumulate(K, cols[f], z)
        # suff-stats: integer counts, exactly the bincounts
        heads = np.bincount(z, weights=cols[f].astype(np.float64), minlength=K).astype(np.uint32)
        assert np.array_equal(ss["heads"], heads) and np.array_equal(ss["heads"] + ss["tails"], np.bincount(z, minlength=K))
        m32 = F32.score_matrix(ss, cols[f])
        total32 += float(m32.astype(np.float64).sum())
        total64 += float(F64.score_matrix(orc.widen_ss(orc.BB, ss), cols[f]).sum())
        # closed form: log((v ? alpha + h : beta + t) / (alpha + beta + h + t)) in float64
        h, t = ss["heads"].astype(np.float64), ss["tails"].astype(np.float64)
        want = np.where(cols[f][:, None] != 0, np.log((2.0 + h) / (4.0 + h + t)), np.log((2.0 + t) / (4.0 + h + t)))
        assert np.abs(m32 - want).max() <= 1e-6 * np.maximum(1.0, np.abs(want)).max()
    # the driver's per-value pass and the batch entry add up the same 1.28 M float scores
    assert abs(rec["score_sum"] - total32) <= 1e-9 * abs(total32)
    assert abs(rec["score_sum"] - total64) <= 1e-6 * abs(total64)
    assert rec["score_evals_per_s_1core"] > 0 and rec["perf_group_evals_per_s"] > 0
