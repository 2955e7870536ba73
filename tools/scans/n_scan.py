#!/usr/bin/env python
"""score pass / sweep step against the ROW COUNT (launch shapes change with it): C3's columns at a few table sizes.
usage: tools/scans/n_scan.py [K ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import common_amd  # noqa: E402
from tools.bench_configs import make_columns, timed  # noqa: E402

Ks = [int(a) for a in sys.argv[1:]] or [32, 256, 300]
ctx = common_amd.Context(0)
spec = [(common_amd.BB, 0), (common_amd.GP, 0), (common_amd.DD, 32), (common_amd.NICH, 0)] * 16
for K in Ks:
    for N in (1000, 4096, 8192, 12000, 16384, 20000, 24576, 32768, 40000, 49152, 65536, 100000, 131072):
        cols, z = make_columns(ctx, spec, N, K, 73)
        view = common_amd.DataView.from_tensors(ctx, cols)
        st = common_amd.State(ctx, spec, K)
        st.set_alpha(1.0)
        st.accumulate(view, z)
        zz, it = z.clone(), [0]

        def step():
            st.sweep_step(view, zz, seed=1, sweep=it[0])
            it[0] += 1
        out = torch.empty((N, K), dtype=torch.float32, device=ctx.torch_device)
        sw = timed(step, 20, warmup=3)[1]
        sc = timed(lambda: st.score_value(view, out=out), 20)[1]
        print("K %4d N %7d  sweep_step %.4f ms  score %.4f ms  (%.2f / %.2f ns per row)" % (K, N, sw, sc, sw * 1e6 / N, sc * 1e6 / N), flush=True)
        del st, view
