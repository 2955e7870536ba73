#!/usr/bin/env python
"""The single-nich fused sweep across table sizes: rows/s and evaluations/s per K (which kernel takes it is the
launcher's choice: k_narrow up to 64 groups, k_sweep_nich1_t up to 1024, k_sweep_nich1_rows beyond).
usage: tools/scans/sweep_vs_k.py > profiles/<tag>_sweep_vs_k.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import common_amd  # noqa: E402
from tools.bench_configs import make_columns, timed  # noqa: E402

ctx = common_amd.Context(0)
spec = [(common_amd.NICH, 0)]
print("single nich feature, fused Gibbs assignment pass (msc_sweep_assign), HIP events, MI355X")
print("%8s %10s %10s %12s %14s" % ("K", "rows", "ms", "rows/s", "evals/s"))
for K in (16, 64, 65, 128, 256, 512, 1024, 1025, 2048, 4096, 8192, 16384):
    N = 1_000_000 if K <= 4096 else 250_000
    cols, z = make_columns(ctx, spec, N, K, 73)
    view = common_amd.DataView.from_tensors(ctx, cols)
    st = common_amd.State(ctx, spec, K)
    st.accumulate(view, z)
    it = [0]

    def one():
        it[0] += 1
        st.sweep_assign(view, z, seed=1, sweep=it[0])
    w, a, m = timed(one, 20, 5)
    print("%8d %10d %10.4f %12.3e %14.3e" % (K, N, a, N / (a * 1e-3), N * K / (a * 1e-3)), flush=True)
    del cols, view, st
