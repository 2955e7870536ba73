#!/usr/bin/env python
"""C3's columns on SMALL row counts: sweep step and score pass, default against the A/B switches (each state planned under its
own environment).  usage: tools/scans/small_n.py [N ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import common_amd  # noqa: E402
from tools.bench_configs import make_columns, timed  # noqa: E402

Ns = [int(a) for a in sys.argv[1:]] or [4096, 20000, 65536]
ctx = common_amd.Context(0)
spec = [(common_amd.BB, 0), (common_amd.GP, 0), (common_amd.DD, 32), (common_amd.NICH, 0)] * 16
for N in Ns:
    for K in (32, 100, 300):
        cols, z = make_columns(ctx, spec, N, K, 73)
        view = common_amd.DataView.from_tensors(ctx, cols)
        row = {}
        for name, env in (("default", {}), ("no_bb_fuse", {"MSC_NO_BB_FUSE": "1"}), ("no_fused_tail", {"MSC_NO_FUSED_TAIL": "1"}),
                          ("rows_forced", {"MSC_TAIL_MIN_ROWS": "1"})):
            os.environ.update(env)
            st = common_amd.State(ctx, spec, K)
            st.set_alpha(1.0)
            st.accumulate(view, z)
            zz, it = z.clone(), [0]

            def step():
                st.sweep_step(view, zz, seed=1, sweep=it[0])
                it[0] += 1
            out = torch.empty((N, K), dtype=torch.float32, device=ctx.torch_device)
            row[name] = (round(timed(step, 20, warmup=3)[1], 4), round(timed(lambda: st.score_value(view, out=out), 20)[1], 4))
            for k in env:
                os.environ.pop(k)
            del st
        print(N, K, row, flush=True)
