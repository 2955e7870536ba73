#!/usr/bin/env python
"""the leave-one-out pre-pass on C3's columns (1M rows, K = 256): score pass with leave-one-out + prior against the plain one
-- the difference is k_loo_own(_lds) + the finish -- and, under rocprofv3, the kernel's own average.  usage: loo_case.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import common_amd  # noqa: E402
from tools.bench_configs import make_columns, timed  # noqa: E402

ctx = common_amd.Context(0)
N, K = 1_000_000, 256
spec = [(common_amd.BB, 0), (common_amd.GP, 0), (common_amd.DD, 32), (common_amd.NICH, 0)] * 16
cols, z = make_columns(ctx, spec, N, K, 73)
view = common_amd.DataView.from_tensors(ctx, cols)
st = common_amd.State(ctx, spec, K)
st.set_alpha(1.0)
st.accumulate(view, z)
zs = z.clone()
it = [0]


def step():
    it[0] += 1
    st.sweep_step(view, zs, seed=73, sweep=it[0])


print(json.dumps({"sweep_step_ms": timed(step, 20, warmup=3)[1],
                  "sweep_assign_ms": timed(lambda: st.sweep_assign(view, zs, seed=1, sweep=it[0]), 20, warmup=2)[1]}), flush=True)
