#!/usr/bin/env python
"""Does a SPREAD store order make the C2 pass's rate independent of where the driver put the score matrix?
  Eight 1 GB matrices from torch.empty held side by side (the placement lottery: some take the
dense write stream at ~5.5 TB/s, some at ~7), the same pass into each with the resident waves writing one dense window
(MSC_NICH1_SPREAD=0) and S windows spread over the whole buffer (S = 4, 16, 64, 256, 1024); results are checked equal.
(Round 5: the knobs this scan drives live in tools/microbench/r05_experiment_switches.patch, not in the product.)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import common_amd  # noqa: E402
from bench import c2_data  # noqa: E402

ctx = common_amd.Context(0)
N, K = 1_000_000, 256
x, z = c2_data(torch, ctx.torch_device, N, K, 73)
view = common_amd.DataView.from_tensors(ctx, [x])
st = common_amd.State(ctx, [(common_amd.NICH, 0)], K)
st.accumulate(view, z)
bufs = [torch.empty((N, K), dtype=torch.float32, device=ctx.torch_device) for _ in range(8)]
if "--lib-alloc" in sys.argv:                             # ... and two from the library's placing allocator
    for _ in range(2):
        bufs.append(ctx.alloc((N, K), torch.float32))
        print(json.dumps({"msc_device_alloc": ctx.alloc_stats()}), flush=True)
alg = 4.0 * N + 4.0 * N * K


def rate(buf, steps=60):
    for _ in range(30):
        st.score_value(view, out=buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        st.score_value(view, out=buf)
    e1.record()
    torch.cuda.synchronize()
    return alg / (e0.elapsed_time(e1) / steps * 1e-3) / 1e9


for _ in range(300):
    st.score_value(view, out=bufs[0])                     # clocks
os.environ["MSC_NICH1_SPREAD"] = "0"
st.score_value(view, out=bufs[0])
ref = bufs[0].clone()
sweep = [("spread", S) for S in (0, 4, 16, 64, 256, 1024, 0)]
if "--stores" in sys.argv:                                # the store's cache policy instead (MSC_NICH1_STORE)
    sweep = [("store", f) for f in (0, 1, 6, 2, 0, 1)]
names = {0: "nt (product)", 1: "plain", 2: "sc1", 3: "sc0 sc1", 4: "nt sc1", 5: "nt sc0 sc1", 6: "sc0"}
for what, v in sweep:
    os.environ["MSC_NICH1_SPREAD" if what == "spread" else "MSC_NICH1_STORE"] = str(v)
    rates = [rate(b) for b in bufs]
    st.score_value(view, out=bufs[1])
    torch.cuda.synchronize()
    same = bool(torch.equal(bufs[1], ref))
    print(json.dumps({what: v if what == "spread" else names[v], "GBps": [round(r) for r in rates],
                      "frac_of_8TBps": [round(r / 8000, 3) for r in rates], "same_result": same}), flush=True)
