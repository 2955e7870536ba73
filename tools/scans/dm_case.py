#!/usr/bin/env python
"""4 x dm(4) on 1M rows ("1.2 ms whatever K is"): the scoring pass at several K, for the counter
evidence of profiles/r04_dm.txt.  usage: tools/scans/dm_case.py [K ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import common_amd  # noqa: E402
from tools.bench_configs import make_columns, timed  # noqa: E402

Ks = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [32, 256]
small = "--small" in sys.argv          # counts small enough that all dim + 1 tables of a feature fit the LDS slot (staged)
ctx = common_amd.Context(0)
N = 1_000_000
spec = [(common_amd.DM, 4)] * 4
for K in Ks:
    cols, z = make_columns(ctx, spec, N, K, 73)
    if small:
        cols = [(c // 2).contiguous() for c in cols]
    view = common_amd.DataView.from_tensors(ctx, cols)
    st = common_amd.State(ctx, spec, K)
    st.accumulate(view, z)
    out = torch.empty((N, K), dtype=torch.float32, device=ctx.torch_device)
    ms = timed(lambda: st.score_value(view, out=out), 10)[1]
    mx = [int(c.max().item()) for c in cols]
    print(json.dumps({"K": K, "score_ms": ms, "column_maxima": mx, "rows_total_max": [int(c.sum(1).max().item()) for c in cols]}), flush=True)
    del out, st, view
