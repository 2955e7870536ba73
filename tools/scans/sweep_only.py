#!/usr/bin/env python
"""The sweep's assignment pass alone, for profiling one kernel: tools/scans/sweep_only.py N K [steps]
(a fresh assignment each step: the counts are re-accumulated between the steps, as a sweep step does)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import common_amd  # noqa: E402
from tools.bench_configs import make_columns, timed  # noqa: E402

N, K = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctx = common_amd.Context(0)
spec = [(common_amd.NICH, 0)]
cols, z = make_columns(ctx, spec, N, K, 73)
view = common_amd.DataView.from_tensors(ctx, cols)
st = common_amd.State(ctx, spec, K)
st.accumulate(view, z)
it = [0]


def one():
    it[0] += 1
    st.sweep_assign(view, z, seed=1, sweep=it[0])


w, a, m = timed(one, steps, warmup=2)
print("N %d K %d sweep_assign avg %.4f ms min %.4f ms" % (N, K, a, m))
