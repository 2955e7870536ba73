#!/usr/bin/env python
"""The narrow tail of a mixed state (K just past a 256-group tile): leave-one-out + prior scores of the whole matrix, from the
lane<->row kernel and from the lanes-are-groups one (MSC_TAIL_OLD, read per call), compared bit for bit.
usage: tools/scans/tail_ab.py [K ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import common_amd  # noqa: E402
from tools.bench_configs import make_columns  # noqa: E402

Ks = [int(a) for a in sys.argv[1:]] or [257, 273, 300, 320]
ctx = common_amd.Context(0)
N = 30_011
spec = [(common_amd.BB, 0), (common_amd.GP, 0), (common_amd.DD, 32), (common_amd.NICH, 0)] * 5
for K in Ks:
    cols, z = make_columns(ctx, spec, N, K, 73)
    z[:7] = K - 1                                # a small group in the tail; one row alone in its group
    z[7] = K - 2
    view = common_amd.DataView.from_tensors(ctx, cols)
    st = common_amd.State(ctx, spec, K)
    st.set_alpha(1.3)
    st.accumulate(view, z)
    for kw in (dict(), dict(z=z, crp_prior=True)):
        got = []
        for old in (False, True):
            if old:
                os.environ["MSC_TAIL_OLD"] = "1"
            else:
                os.environ.pop("MSC_TAIL_OLD", None)
            out = torch.full((N, K), -7.0, dtype=torch.float32, device=ctx.torch_device)
            st.score_value(view, out=out, **kw)
            torch.cuda.synchronize()
            got.append(out.cpu().numpy())
        a, b = got
        print(K, sorted(kw), "differing words:", int((a.view(np.uint32) != b.view(np.uint32)).sum()), "of", a.size,
              "max |d|", float(np.nanmax(np.abs(a - b))), flush=True)
