#!/usr/bin/env python
"""a grid of (feature list, rows, groups): scoring pass and sweep assign, ms -- to find table sizes / row counts where the
kernel choice goes wrong (a pass should not get dearer with fewer groups or fewer rows).
usage: tools/scans/grid_scan.py [--spec=a:n,b:m ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import common_amd  # noqa: E402
from tools.bench_configs import make_columns, timed  # noqa: E402

FAM = {"bb": (common_amd.BB, 0), "gp": (common_amd.GP, 0), "dd32": (common_amd.DD, 32), "dd8": (common_amd.DD, 8),
       "dd100": (common_amd.DD, 100), "nich": (common_amd.NICH, 0), "bnb": (common_amd.BNB, 0)}
specs = [a.split("=")[1] for a in sys.argv[1:] if a.startswith("--spec=")] or ["bb:8", "bb:4,gp:4,dd8:4,nich:4", "dd100:4,nich:2", "nich:3"]
Ns = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--n=")] or [50_000, 200_000, 1_000_000]
Ks = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--k=")] or [16, 48, 64, 100, 128, 200, 256, 300, 512]
ctx = common_amd.Context(0)
for sp in specs:
    spec = []
    for part in sp.split(","):
        name, n = part.split(":")
        spec += [FAM[name]] * int(n)
    for N in Ns:
        row = []
        for K in Ks:
            cols, z = make_columns(ctx, spec, N, K, 73)
            view = common_amd.DataView.from_tensors(ctx, cols)
            st = common_amd.State(ctx, spec, K)
            st.set_alpha(1.0)
            st.accumulate(view, z)
            out = torch.empty((N, K), dtype=torch.float32, device=ctx.torch_device)
            sc = timed(lambda: st.score_value(view, out=out), 6, warmup=2)[1]
            zs = z.clone()
            sw = timed(lambda: st.sweep_assign(view, zs, seed=1, sweep=3), 6, warmup=2)[1]
            row.append((K, round(sc, 4), round(sw, 4)))
            del out, view, st
        torch.cuda.empty_cache()
        print(json.dumps({"spec": sp, "N": N, "K_score_sweep": row}), flush=True)
