#!/usr/bin/env python
"""C3's columns (bb + gp + dd32 + nich) x16, N = 1M: the pieces of a sweep step one by one, at several K.
usage: tools/scans/c3_pieces.py [K ...]   (default 256)
  score pass, leave-one-out + prior score pass (what the sweep's first half costs), accumulate, whole sweep step
  --family=bb|gp|dd|nich : sixteen columns of one family;  --spec=bb:8,nich:8 : any feature list"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import common_amd  # noqa: E402
from tools.bench_configs import make_columns, timed  # noqa: E402

Ks = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [256]
masked = "--masked" in sys.argv or "--masked-nich" in sys.argv
ctx = common_amd.Context(0)
N = 1_000_000
spec = [(common_amd.BB, 0), (common_amd.GP, 0), (common_amd.DD, 32), (common_amd.NICH, 0)] * 16
for a in sys.argv[1:]:                              # --family=bb|gp|dd|nich: sixteen columns of one family
    if a.startswith("--family="):
        spec = [{"bb": (common_amd.BB, 0), "gp": (common_amd.GP, 0), "dd": (common_amd.DD, 32),
                 "nich": (common_amd.NICH, 0)}[a.split("=")[1]]] * 16
    if a.startswith("--spec="):                     # --spec=bb:8,nich:8,dd32:2 : any list, in that order
        fam = {"bb": (common_amd.BB, 0), "gp": (common_amd.GP, 0), "dd32": (common_amd.DD, 32), "dd8": (common_amd.DD, 8), "nich": (common_amd.NICH, 0)}
        spec = []
        for part in a.split("=")[1].split(","):
            name, n = part.split(":")
            spec += [fam[name]] * int(n)
for K in Ks:
    cols, z = make_columns(ctx, spec, N, K, 73)
    masks = None
    if masked:                                  # one masked bb column: the first phase is no longer lookup runs only
        masks = [None] * len(cols)
        masks[0] = (torch.rand(N, device=ctx.torch_device) < 0.05).to(torch.uint8).contiguous()
        if "--masked-nich" in sys.argv:         # ... and a masked nich column
            masks[3] = (torch.rand(N, device=ctx.torch_device) < 0.05).to(torch.uint8).contiguous()
    view = common_amd.DataView.from_tensors(ctx, cols, masks)
    st = common_amd.State(ctx, spec, K)
    st.set_alpha(1.0)
    st.accumulate(view, z)
    out = torch.empty((N, K), dtype=torch.float32, device=ctx.torch_device)
    rec = {"K": K, "masked": masked}
    rec["score_ms"] = timed(lambda: st.score_value(view, out=out), 10)[1]
    rec["score_loo_crp_ms"] = timed(lambda: st.score_value(view, out=out, z=z, crp_prior=True), 10)[1]
    zs = z.clone()
    it = [0]

    def step():
        it[0] += 1
        st.sweep_step(view, zs, seed=73, sweep=it[0])
    rec["sweep_step_ms"] = timed(step, 10, warmup=2)[1]
    rec["sweep_assign_ms"] = timed(lambda: st.sweep_assign(view, zs, seed=1, sweep=it[0]), 10, warmup=1)[1]
    rec["accumulate_ms"] = timed(lambda: st.accumulate(view, zs), 10, warmup=1)[1]
    print(json.dumps(rec), flush=True)
    del out, view, st
    torch.cuda.empty_cache()
