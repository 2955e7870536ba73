#!/usr/bin/env python
"""NIW scoring pass (f64 MFMA kernel) at N = 256k, K = 128 for several dimensions: ms, TFLOP/s on the 2 d^2 N K count,
fraction of the 78.6 TFLOP/s f64 matrix peak.  usage: tools/scans/niw_dims.py [d ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import common_amd  # noqa: E402
from tools.bench_configs import make_columns, timed  # noqa: E402

dims = [int(a) for a in sys.argv[1:]] or [32, 48, 64, 128]
ctx = common_amd.Context(0)
N, K = 262_144, 128
for d in dims:
    spec = [(common_amd.NIW, d)]
    cols, z = make_columns(ctx, spec, N, K, 73)
    view = common_amd.DataView.from_tensors(ctx, cols)
    st = common_amd.State(ctx, spec, K)
    st.accumulate(view, z)
    out = torch.empty((N, K), dtype=torch.float32, device=ctx.torch_device)
    w, avg, mn = timed(lambda: st.score_value(view, out=out), 10, warmup=3)
    flops = 2.0 * d * d * N * K
    nb = (d + 15) // 16
    print(json.dumps({"d": d, "ms": avg, "ms_min": mn, "tflops_survey_count": flops / (avg * 1e-3) / 1e12,
                      "frac_of_78.6": flops / (avg * 1e-3) / 1e12 / 78.6,
                      "executed_fraction_of_count": (nb + 1) / (2.0 * nb) * (16 * nb) ** 2 / float(d * d)}), flush=True)
    del out, view, st, cols
    torch.cuda.empty_cache()
