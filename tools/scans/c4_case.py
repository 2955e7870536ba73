#!/usr/bin/env python
"""C4 (NIW dim 32, N = 256k, K = 128): the scoring pass on the f64 matrix pipe, HIP-event average.  usage: tools/scans/c4_case.py [dim ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import common_amd  # noqa: E402
from tools.bench_configs import make_columns, timed  # noqa: E402

ctx = common_amd.Context(0)
for d in [int(a) for a in sys.argv[1:]] or [32]:
    N, K = 262_144, 128
    spec = [(common_amd.NIW, d)]
    cols, z = make_columns(ctx, spec, N, K, 73)
    view = common_amd.DataView.from_tensors(ctx, cols)
    st = common_amd.State(ctx, spec, K)
    st.accumulate(view, z)
    out = torch.empty((N, K), dtype=torch.float32, device=ctx.torch_device)
    r = {"dim": d}
    r["score_f64_ms"] = timed(lambda: st.score_value(view, out=out), 20, warmup=5)[1]
    r["kernel"] = ctx.last_kernel("score")
    r["score_f64_loo_ms"] = timed(lambda: st.score_value(view, z=z, out=out), 20, warmup=5)[1]     # (+ k_niw_loo_patch)
    r["kernel_loo"] = ctx.last_kernel("score")
    if d <= 32:
        r["score_f32_ms"] = timed(lambda: st.score_value(view, out=out, niw_f32=True), 20, warmup=5)[1]
    print(json.dumps(r), flush=True)
